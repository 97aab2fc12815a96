#!/bin/bash
# round 3: wider tiles of the fused gradient GEMM at the C3 shape (240 x 240, 60 000 images)
out=${1:-gpurun_out/tune_bgemm_r3.txt}
: > $out
for c in 0 3 4 6 7 8; do
  TNML_BGF_CFG=$c python bench.py --steps 12 --warmup 4 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('TNML_BGF_CFG=$c: %7.2f bond updates/s | bgemm %.4f ms (4 launches = %.1f us each)  slab_reduce %.4f  fwd_fused %.4f' % (d['value'], k.get('bgemm',0), 250*k.get('bgemm',0), k.get('slab_reduce',0), k.get('fwd_fused',0)))" >> $out
done
cat $out
