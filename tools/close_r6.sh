mkdir -p gpurun_out/close6
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/close6/driver_form.json 2> gpurun_out/close6/driver_form.err
python bench.py --no-cpu-baseline --workload 8d > gpurun_out/close6/w8d.json 2>/dev/null
python bench.py --no-cpu-baseline --plain --images 7500 --steps 60 > gpurun_out/close6/shard7500.json 2>/dev/null
TNML_RES_PACE=1 python bench.py --no-cpu-baseline --plain --images 7500 --steps 60 > gpurun_out/close6/shard7500_pace1.json 2>/dev/null
for f in driver_form w8d shard7500 shard7500_pace1; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
d=json.loads(open("gpurun_out/close6/%s.json"%f).read().strip().splitlines()[-1])
print(f, round(d["value"],1), round(d["ms_per_step"],3), "grad", round(d["gradient_phase_ms"],3), "svd", round(d["svd_ms"],3), {a:round(b,3) for a,b in d["kernel_ms_per_step"].items()})
for k,v in (d.get("roofline_kernels") or {}).items(): print("   ", k, v["binding_roof"], round(v["frac_of_binding_roof"],3), "mfma", round(v["frac"],3), "hbm", round(v["hbm_frac"],3), "us", round(v["avg_launch_ms"]*1e3,1))
PY
done
