mkdir -p gpurun_out/devin
TNML_SPEC_SPLIT=0 TNML_SVD_PRINT=-1 python bench.py --no-cpu-baseline --plain --steps 60 --warmup 5 > gpurun_out/devin/full.json 2> gpurun_out/devin/full.err
TNML_SPEC_SPLIT=0 TNML_SVD_PRINT=-1 python bench.py --no-cpu-baseline --plain --images 7500 --steps 100 > gpurun_out/devin/shard.json 2> gpurun_out/devin/shard.err
TNML_SPEC_SPLIT=0 TNML_SVD_PRINT=-1 python bench.py --no-cpu-baseline --plain --workload 8d --steps 200 > gpurun_out/devin/w8d.json 2> gpurun_out/devin/w8d.err
for f in full shard w8d; do echo $f; grep svd_check gpurun_out/devin/$f.err | wc -l; grep svd_check gpurun_out/devin/$f.err | tail -40 | cut -c1-150; done
