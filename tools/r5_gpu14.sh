mkdir -p gpurun_out/r5n
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5n/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5n/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5n/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5n/smoke.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5n/driver_form.json 2> gpurun_out/r5n/driver_form.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5n/driver_form.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['svd_ms'], d['roofline']['kernel'][:20], d['roofline']['frac'])
for k,v in d['roofline_kernels'].items(): print(k, round(v['frac'],3), round(v['avg_launch_ms']*1e3,1), v['traffic_over_algorithmic'])
PY
