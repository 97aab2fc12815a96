mkdir -p gpurun_out/r5k
./tools/probe/probe_teig > gpurun_out/r5k/probe_teig.txt 2>&1; head -3 gpurun_out/r5k/probe_teig.txt | cut -c1-200
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5k/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r5k/pytest_gpu.txt
for dt in f64 f32 bf16; do
  timeout 300 python bench.py --maxm 300 --images 7500 --steps 20 --literal-steps 0 --no-cpu-baseline --no-extras --dtype $dt > gpurun_out/r5k/m300_$dt.json 2> gpurun_out/r5k/m300_$dt.err; echo "m300 $dt rc=$?"
done
python - <<'PY'
import json
for f in ('m300_f64','m300_f32','m300_bf16'):
    try:
        d=json.loads(open('gpurun_out/r5k/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), round(d['ms_per_step'],3), 'svd', round(d['svd_ms'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, d['config']['workload'][:60])
    except Exception as e:
        print(f, 'failed', e); print(open('gpurun_out/r5k/%s.err'%f).read()[-800:])
PY
