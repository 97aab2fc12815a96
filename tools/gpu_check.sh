#!/bin/bash
# what the driver runs at round end, in one gpurun call: the GPU test suite, smoke(), one bench line
mkdir -p gpurun_out/check
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/check/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/check/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/check/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/check/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/check/bench.json
