mkdir -p gpurun_out/r5q gpurun_out/final5
export TNML_COMMIT=985533e
./tools/probe/probe_sgemm > gpurun_out/r5q/probe_sgemm2.txt 2>&1; cut -c1-175 gpurun_out/r5q/probe_sgemm2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "svd or split or spectr or speculative or cli_driver" > gpurun_out/r5q/pytest_split.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5q/pytest_split.txt
bash tools/pmc_bench.sh > gpurun_out/r5q/pmc_run.txt 2>&1; tail -25 gpurun_out/r5q/pmc_run.txt | cut -c1-200
cp gpurun_out/pmc/pmc_traffic.json profiles/r05_pmc_traffic.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final5/driver_form.json 2> gpurun_out/final5/driver_form.err
python bench.py > gpurun_out/final5/default.json 2> gpurun_out/final5/default.err
python bench.py --no-cpu-baseline --plain --images 7500 --steps 60 > gpurun_out/final5/shard7500.json 2>/dev/null
python - <<'PY'
import json
for f in ("driver_form","default","shard7500"):
    d=json.loads(open("gpurun_out/final5/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["svd_ms"], d["roofline"]["kernel"][:20], d["roofline"]["frac"], d["roofline"].get("traffic"), (d.get("cpu_baseline") or {}).get("value"))
    for r in d.get("roofline_kernels") or []: print("   ", r)
PY
