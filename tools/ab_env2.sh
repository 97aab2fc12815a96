#!/bin/bash
# A/B of environment knobs on a 60-step window: tools/ab_env2.sh "A=1 B=0" "A=0 B=1" ...  (each setting run twice, interleaved)
for rep in 1 2; do for v in "$@"; do
  r=$(env $v python bench.py --no-cpu-baseline --steps 60 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%.2f/s %.3f ms/step  fwd %.3f (%.1f TF) shift %.3f bgemm %.3f ldot %.3f (%.0f GB/s) pupd %.3f' % (d['value'], d['ms_per_step'], k['fgemm_fwd'], d['roofline']['achieved'], k['fgemm_shift'], k['bgemm'], k['labeldot'], d['roofline_hbm']['achieved'], k.get('p_update', 0)))")
  echo "$v: $r"
done; done
