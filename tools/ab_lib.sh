#!/bin/bash
# A/B of library builds on one box: tools/ab_lib.sh "tools/ab/a.so tools/ab/b.so" [bench args]; alternates the builds three times
libs=$1; shift
cp tnml_amd/libtnml.so /tmp/libtnml_orig.so
for rep in 1 2 3; do for l in $libs; do
  cp $l tnml_amd/libtnml.so
  r=$(python bench.py --no-cpu-baseline --plain "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%.2f bond updates/s  %.4f ms/step  bgemm %.4f fgemm_fwd %.4f shift %.4f svd %.3f' % (d['value'], d['ms_per_step'], k.get('bgemm',0), k.get('fgemm_fwd',0), k.get('fgemm_shift',0), k.get('svd',0)))")
  echo "$l rep $rep: $r"
done; done
cp /tmp/libtnml_orig.so tnml_amd/libtnml.so
