#!/bin/bash
# tile configurations of the feature GEMM (forward) and the gradient GEMM at maxm = 300 (BASELINE config 5) on one rank's
# share of an 8-GPU run (7 500 images) and on 30 000 images; prints the per-class kernel times of bench.py
out=${1:-gpurun_out/tune_m300.txt}
: > $out
run() {
  python bench.py --maxm 300 --images $IM --steps 8 --warmup 3 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('%-34s images %6d: %7.2f bond updates/s | fgemm_fwd %.3f ms (5 launches)  bgemm %.3f ms (4 launches)  shift %.3f  svd %.3f' % ('$1', $IM, d['value'], k.get('fgemm_fwd',0), k.get('bgemm',0), k.get('fgemm_shift',0), k.get('svd',0)))" >> $out
}
for IM in ${IMS:-7500}; do
  for c in 1 2 3 4 5 6 7 8 9; do TNML_FG64_BIG_CFG=$c TNML_BGF_BIG_CFG=1 run "TNML_FG64_BIG_CFG=$c"; done
  for c in 2 3 4 5 6 7 8 9; do TNML_BGF_BIG_CFG=$c TNML_FG64_BIG_CFG=1 run "TNML_BGF_BIG_CFG=$c"; done
done
cat $out
