mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/default.json 2> gpurun_out/final/default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/driver_form.json 2>/dev/null
python bench.py --steps 40 --no-cpu-baseline > gpurun_out/final/window40.json 2>/dev/null
python bench.py --steps 40 --no-cpu-baseline --dtype f64_e32 > gpurun_out/final/window40_e32.json 2>/dev/null
python bench.py --steps 40 --no-cpu-baseline --single-label 3 > gpurun_out/final/single3.json 2>/dev/null
python bench.py --no-cpu-baseline --workload 8d > gpurun_out/final/w8d.json 2>/dev/null
python bench.py --no-cpu-baseline --images 7500 --steps 60 > gpurun_out/final/shard7500.json 2>/dev/null
python bench.py --no-cpu-baseline --maxm 300 --images 7500 --steps 20 --literal-steps 0 > gpurun_out/final/m300_shard.json 2>/dev/null
for f in default driver_form window40 window40_e32 single3 w8d shard7500 m300_shard; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.load(open("gpurun_out/final/%s.json"%f))
    k=d["kernel_ms_per_step"]
    print(f, "%.1f/s %.3f ms lit %s | rf %.3f hbm %.3f step_exec %.3f alg %.3f | grad %.3f svd %.3f |" % (d["value"], d["ms_per_step"], d["value_literal_order"] and round(d["value_literal_order"],1), d["roofline"]["frac"], (d["roofline_hbm"] or {}).get("frac",0), d["roofline_step"]["frac_executed"], d["roofline_step"]["algorithmic_gflop_per_step"] / d["ms_per_step"] / 78.6, d["gradient_phase_ms"], d["svd_ms"]), {a:round(b,3) for a,b in k.items()}, d.get("cpu_baseline",{}).get("value"), d["device_gb"])
except Exception as e: print(f, "failed", e)
PY
done
