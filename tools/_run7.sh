bash tools/prof_bench.sh r06_c3 --steps 20 --warmup 5 > gpurun_out/prof_r06_c3.txt 2>&1
bash tools/prof_bench.sh r06_shard --images 7500 --steps 60 --warmup 5 > gpurun_out/prof_r06_shard.txt 2>&1
head -30 gpurun_out/prof_r06_c3.txt | cut -c1-150; head -30 gpurun_out/prof_r06_shard.txt | cut -c1-150
