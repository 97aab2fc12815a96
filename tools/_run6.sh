mkdir -p gpurun_out/r6m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6m/pytest.txt 2>&1; tail -5 gpurun_out/r6m/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6m/driver.json 2> gpurun_out/r6m/driver.err
python bench.py --no-cpu-baseline --plain --images 7500 --steps 60 > gpurun_out/r6m/shard7500.json 2>/dev/null
python - <<'PY'
import json
for f in ["driver","shard7500"]:
    try:
        d=json.loads(open("gpurun_out/r6m/%s.json"%f).read().strip().splitlines()[-1])
        k=d["kernel_ms_per_step"]
        print(f, "%.1f/s %.3f ms"%(d["value"],d["ms_per_step"]), "grad_phase %.3f svd %.3f"%(d["gradient_phase_ms"],d["svd_ms"]), {a:round(b,3) for a,b in k.items()})
        for kk,v in d.get("roofline_kernels",{}).items(): print("   ",kk, round(v["frac"],3), round(v["avg_launch_ms"]*1e3,1),"us", v["kernel"][:30])
        print("   spec", d.get("speculative_split"))
    except Exception as e: print(f, "failed", e)
PY
