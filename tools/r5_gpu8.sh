mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests/test_multirank_one_gpu.py -x -q -m gpu > gpurun_out/r5h/pytest_multirank.txt 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r5h/pytest_multirank.txt
bash tools/prof_bench.sh r5h --steps 20 --warmup 5 > gpurun_out/r5h/prof_driver.txt 2>&1
cat gpurun_out/r5h/prof_driver.txt | cut -c1-140
