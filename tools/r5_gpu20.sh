mkdir -p gpurun_out/r5q
cp tnml_amd/libtnml.so /tmp/libtnml_cur.so
{
python tools/cli_deviation.py "current"
TNML_SMALL_GEMM=0 python tools/cli_deviation.py "current, split products on rocBLAS"
cp tools/variants/libtnml_old_teig.so tnml_amd/libtnml.so; python tools/cli_deviation.py "inverse iteration with two full sweeps (before 56d8f83)"
cp tools/variants/libtnml_old_sytrd.so tnml_amd/libtnml.so; python tools/cli_deviation.py "tridiagonalisation of 14286ab"
cp /tmp/libtnml_cur.so tnml_amd/libtnml.so
} > gpurun_out/r5q/cli_deviation.txt 2>&1
cat gpurun_out/r5q/cli_deviation.txt | cut -c1-250
for w in 0 240 480 960 1280; do
  TNML_BGEMM_WGS=$w timeout 300 python bench.py --steps 60 --warmup 10 --images 7500 --no-cpu-baseline --plain > gpurun_out/r5q/bw_shard_$w.json 2> gpurun_out/r5q/bw_shard_$w.err
done
for w in 0 640 1280 5120; do
  TNML_BGEMM_WGS=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --plain > gpurun_out/r5q/bw_full_$w.json 2> gpurun_out/r5q/bw_full_$w.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5q/bw_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernel_ms_per_step']
        print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],4), 'bgemm', round(k['bgemm'],4), 'cg_vec', round(k['cg_vec'],4), 'svd', round(d['svd_ms'],3))
    except Exception as e:
        print(f, 'failed', e)
PY
