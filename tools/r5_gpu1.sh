set -x
mkdir -p gpurun_out/r5a
./tools/probe/probe_teig > gpurun_out/r5a/probe_teig.txt 2>&1; echo "teig rc=$?"
./tools/probe/probe_eigh > gpurun_out/r5a/probe_eigh.txt 2>&1; echo "eigh rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "svd or split or spectr" > gpurun_out/r5a/pytest_split.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r5a/pytest_split.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --plain > gpurun_out/r5a/bench_plain.json 2> gpurun_out/r5a/bench_plain.err; echo "bench rc=$?"
cat gpurun_out/r5a/probe_teig.txt gpurun_out/r5a/probe_eigh.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5a/bench_plain.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['svd_ms'], d['kernel_ms_per_step'])
PY
