#!/bin/bash
# second pass of tools/tune_m300.sh: more gradient-GEMM tiles around the winner, the shift form, and the winners at 30 000 images
out=${1:-gpurun_out/tune_m300b.txt}
: > $out
run() {
  python bench.py --maxm 300 --images $IM --steps 8 --warmup 3 --no-cpu-baseline --plain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('%-34s images %6d: %7.2f bond updates/s | fgemm_fwd %.3f ms (5 launches)  bgemm %.3f ms (4 launches)  shift %.3f  svd %.3f' % ('$1', $IM, d['value'], k.get('fgemm_fwd',0), k.get('bgemm',0), k.get('fgemm_shift',0), k.get('svd',0)))" >> $out
}
IM=7500
for c in 0 10 11 12 13 14 15 16; do TNML_BGF_BIG_CFG=$c run "TNML_BGF_BIG_CFG=$c"; done
for c in 1 3 4 7 8; do TNML_FG64_SHIFT_CFG=$c run "TNML_FG64_SHIFT_CFG=$c"; done
IM=30000
for c in 0 3 6; do TNML_FG64_BIG_CFG=$c run "TNML_FG64_BIG_CFG=$c"; done
for c in 0 3 8; do TNML_BGF_BIG_CFG=$c run "TNML_BGF_BIG_CFG=$c"; done
cat $out
