mkdir -p gpurun_out/r5q
timeout 120 ./tools/probe/probe_eigh > gpurun_out/r5q/probe_eigh.txt 2>&1; cat gpurun_out/r5q/probe_eigh.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "svd or split or spectr or speculative" > gpurun_out/r5q/pytest_split.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5q/pytest_split.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --plain > gpurun_out/r5q/bench_plain.json 2> gpurun_out/r5q/bench_plain.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 60 --warmup 10 --images 7500 --no-cpu-baseline --plain > gpurun_out/r5q/bench_shard.json 2> gpurun_out/r5q/bench_shard.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ('bench_plain','bench_shard'):
    try:
        d=json.loads(open('gpurun_out/r5q/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['svd_ms'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'failed', e); print(open('gpurun_out/r5q/%s.err'%f).read()[-1500:])
PY
