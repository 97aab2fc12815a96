mkdir -p gpurun_out/r5g
./tools/probe/probe_eigh_pad0 > gpurun_out/r5g/eigh_pad0.txt 2>&1; ./tools/probe/probe_eigh_pad1 > gpurun_out/r5g/eigh_pad1.txt 2>&1
grep -A1 "n=240" gpurun_out/r5g/eigh_pad0.txt; echo ---; grep -A1 "n=240" gpurun_out/r5g/eigh_pad1.txt
timeout 1200 python -m pytest tests/test_multirank_one_gpu.py -x -q -m gpu > gpurun_out/r5g/pytest_multirank.txt 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r5g/pytest_multirank.txt
