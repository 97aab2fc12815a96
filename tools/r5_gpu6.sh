set -x
mkdir -p gpurun_out/r5f
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5f/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r5f/pytest_gpu.txt
