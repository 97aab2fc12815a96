#!/bin/bash
# A/B of an environment knob on one box, alternating runs: tools/ab_env3.sh VAR "a b" [bench args]
var=$1; vals=$2; shift 2
for rep in 1 2 3; do for v in $vals; do
  r=$(env $var=$v python bench.py --no-cpu-baseline --plain "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f bond updates/s  %.4f ms/step  svd %.3f' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('svd',0)))")
  echo "$var=$v rep $rep: $r"
done; done
