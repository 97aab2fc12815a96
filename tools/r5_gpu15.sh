mkdir -p gpurun_out/r5p
./tools/probe/probe_eigh > gpurun_out/r5p/probe_eigh.txt 2>&1; grep -A2 "n=" gpurun_out/r5p/probe_eigh.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "svd or split or spectr or speculative" > gpurun_out/r5p/pytest_split.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5p/pytest_split.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --plain > gpurun_out/r5p/bench_plain.json 2> gpurun_out/r5p/bench_plain.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 60 --warmup 10 --images 7500 --no-cpu-baseline --plain > gpurun_out/r5p/bench_shard.json 2> gpurun_out/r5p/bench_shard.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ('bench_plain','bench_shard'):
    try:
        d=json.loads(open('gpurun_out/r5p/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['svd_ms'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'failed', e); print(open('gpurun_out/r5p/%s.err'%f).read()[-1500:])
PY
