#!/bin/bash
# round-3 closing measurements: driver-form line, the 7 500-image shard at m = 120 and m = 300 (f64 / f64_e32 / bf16x3 / bf16)
o=gpurun_out/final_r3; mkdir -p $o
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/driver_form.json 2> $o/driver_form.err
python bench.py --no-cpu-baseline --plain --images 7500 --steps 60 --warmup 10 > $o/shard7500.json 2>/dev/null
for dt in f64 f64_e32 f32 bf16x3 bf16; do
  python bench.py --no-cpu-baseline --plain --maxm 300 --images 7500 --steps 20 --warmup 10 --dtype $dt > $o/m300_$dt.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final_r3/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); k=d["kernel_ms_per_step"]
        print(f.split("/")[-1], d["dtype"], "%.1f/s %.3f ms | rf %s %.3f | svd %.3f |"%(d["value"],d["ms_per_step"],d["roofline"]["kernel"],d["roofline"]["frac"],d["svd_ms"]), {a:round(b,3) for a,b in k.items()}, "cost", d["last_cost_per_image"])
    except Exception as e: print(f,"failed",e)
PY
