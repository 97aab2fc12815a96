#!/bin/bash
# A/B of one environment knob on the default bench: tools/ab_env.sh NAME v1 v2 ... (each value run twice, interleaved)
name=$1; shift
for rep in 1 2; do for v in "$@"; do
  r=$(env $name=$v python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%.2f/s %.3f ms/step  svd %.3f fwd %.3f bgemm %.3f ldot %.3f shift %.3f' % (d['value'], d['ms_per_step'], k['svd'], k['fgemm_fwd'], k['bgemm'], k['labeldot'], k['fgemm_shift']))")
  echo "$name=$v: $r"
done; done
