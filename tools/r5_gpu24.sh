mkdir -p gpurun_out/r5q
timeout 300 python tools/sweep_bgemm_per.py 7500 0 35 40 41 36 34 33 32 31 30 28 27 26 25 24 22 21 20 44 48 49 52 56 60 64 70 80 0 > gpurun_out/r5q/sweep_per_7500.txt 2>&1
cat gpurun_out/r5q/sweep_per_7500.txt | tail -30
timeout 300 python tools/sweep_bgemm_per.py 60000 0 269 268 270 272 264 256 280 288 300 314 320 235 236 240 248 209 200 192 188 171 160 157 150 128 376 384 400 0 > gpurun_out/r5q/sweep_per_60000.txt 2>&1
cat gpurun_out/r5q/sweep_per_60000.txt | tail -31
