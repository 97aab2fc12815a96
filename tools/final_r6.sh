#!/bin/bash
# the closing measurements of round 6 (one MI355X): driver form with the CPU oracle beside it, whole sweep, per-rank shares of a 2 / 4 / 8-GPU
# run, the SURVEY 8(d) workload, config 5 shard in three arithmetics, and the rocprofv3 kernel trace of the driver form and of the 7 500-image shard
mkdir -p gpurun_out/final6
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final6/driver_form.json 2> gpurun_out/final6/driver_form.err
python bench.py > gpurun_out/final6/default.json 2> gpurun_out/final6/default.err
for n in 30000 15000 7500; do python bench.py --no-cpu-baseline --plain --images $n --steps 60 > gpurun_out/final6/shard$n.json 2>/dev/null; done
for dt in f64 f32 bf16; do python bench.py --no-cpu-baseline --no-extras --maxm 300 --images 7500 --steps 20 --literal-steps 0 --dtype $dt > gpurun_out/final6/m300_$dt.json 2>/dev/null; done
python bench.py --no-cpu-baseline --workload 8d > gpurun_out/final6/w8d.json 2>/dev/null
for f in driver_form default shard30000 shard15000 shard7500 w8d m300_f64 m300_f32 m300_bf16; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/final6/%s.json"%f).read().strip().splitlines()[-1])
    k=d["kernel_ms_per_step"]
    print(f, "%.1f/s %.3f ms lit %s | roofline %s %.3f (traffic %s) | step_exec %.3f | grad %.3f svd %.3f |" % (d["value"], d["ms_per_step"], d.get("value_literal_order") and round(d["value_literal_order"],1), d["roofline"]["kernel"][:12], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline_step"]["frac_executed"], d["gradient_phase_ms"], d["svd_ms"]), {a:round(b,3) for a,b in k.items()}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print(f, "failed", e)
PY
done
bash tools/prof_bench.sh final6 --steps 20 --warmup 5 > gpurun_out/final6/prof_driver.txt 2>&1
head -45 gpurun_out/final6/prof_driver.txt | cut -c1-130
bash tools/prof_bench.sh final6_shard --images 7500 --steps 60 > gpurun_out/final6/prof_shard.txt 2>&1
head -30 gpurun_out/final6/prof_shard.txt | cut -c1-130
