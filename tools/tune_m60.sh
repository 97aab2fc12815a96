#!/bin/bash
# tile configurations of the feature / gradient GEMM at m = 60 (bonds that have shrunk to minm = maxm/2): bench window with maxm = 60
for v in 0 1 2 3 4; do
  r=$(TNML_FG64_GEN_CFG=$v python bench.py --no-cpu-baseline --plain --maxm 60 --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%.1f bond updates/s fgemm_fwd %.3f bgemm %.3f labeldot %.3f shift %.3f svd %.3f' % (d['value'], k['fgemm_fwd'], k['bgemm'], k['labeldot'], k['fgemm_shift'], k['svd']))")
  echo "FG64_GEN_CFG=$v: $r"
done
for v in 1 2 3 4 5 6; do
  r=$(TNML_BGF_GEN_CFG=$v python bench.py --no-cpu-baseline --plain --maxm 60 --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%.1f bond updates/s fgemm_fwd %.3f bgemm %.3f' % (d['value'], k['fgemm_fwd'], k['bgemm']))")
  echo "BGF_GEN_CFG=$v: $r"
done
