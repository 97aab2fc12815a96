mkdir -p gpurun_out/r6a
(timeout 240 python tools/oneshot_processes_m120.py 2 1536 8 1) > gpurun_out/r6a/os2.txt 2>&1
(timeout 300 python tools/oneshot_processes_m120.py 3 1536 8 2) > gpurun_out/r6a/os3.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6a/driver.json 2> gpurun_out/r6a/driver.err
python bench.py --no-cpu-baseline --plain --images 7500 --steps 60 > gpurun_out/r6a/shard7500.json 2>/dev/null
tail -5 gpurun_out/r6a/os2.txt; tail -8 gpurun_out/r6a/os3.txt
