mkdir -p gpurun_out/r5i
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5i/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r5i/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --plain > gpurun_out/r5i/bench_plain.json 2> gpurun_out/r5i/bench_plain.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 60 --warmup 10 --images 7500 --no-cpu-baseline --plain > gpurun_out/r5i/bench_shard.json 2> gpurun_out/r5i/bench_shard.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ('bench_plain','bench_shard'):
    try:
        d=json.loads(open('gpurun_out/r5i/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['svd_ms'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, d['roofline']['kernel'][:20], d['roofline']['frac'])
    except Exception as e:
        print(f, 'failed', e); print(open('gpurun_out/r5i/%s.err'%f).read()[-1500:])
PY
