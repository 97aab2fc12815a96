mkdir -p gpurun_out/r5q
python tools/cli_deviation.py "inverse iteration start reverted" > gpurun_out/r5q/cli_deviation2.txt 2>&1; head -2 gpurun_out/r5q/cli_deviation2.txt | cut -c1-250
for w in 152 190 228 304; do
  TNML_BGEMM_WGS=$w timeout 300 python bench.py --steps 60 --warmup 10 --images 7500 --no-cpu-baseline --plain > gpurun_out/r5q/bw2_shard_$w.json 2> gpurun_out/r5q/bw2_shard_$w.err
done
for w in 152 228 304 380; do
  TNML_BGEMM_WGS=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --plain > gpurun_out/r5q/bw2_full_$w.json 2> gpurun_out/r5q/bw2_full_$w.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5q/bw2_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernel_ms_per_step']
        print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],4), 'bgemm', round(k['bgemm'],4), 'cg_vec', round(k['cg_vec'],4), 'svd', round(d['svd_ms'],3))
    except Exception as e:
        print(f, 'failed', e)
PY
