mkdir -p gpurun_out/r5q
for p in 41 31 27 25 49; do
  TNML_BGEMM_PER=$p timeout 300 python bench.py --steps 60 --warmup 10 --images 7500 --no-cpu-baseline --plain > gpurun_out/r5q/bp_shard_$p.json 2> gpurun_out/r5q/bp_shard_$p.err
done
for p in 236 209 189 171 377; do
  TNML_BGEMM_PER=$p timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --plain > gpurun_out/r5q/bp_full_$p.json 2> gpurun_out/r5q/bp_full_$p.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5q/bp_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernel_ms_per_step']
        print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],4), 'bgemm', round(k['bgemm'],4), 'cg_vec', round(k['cg_vec'],4), 'svd', round(d['svd_ms'],3))
    except Exception as e:
        print(f, 'failed', e)
PY
