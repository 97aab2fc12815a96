#!/bin/bash
# feature-GEMM tile configurations on one rank's share of an 8-GPU run (7500 images): tools/tune_shard.sh cfg...
for cfg in "$@"; do
  r=$(TNML_FG64_CFG=$cfg python bench.py --images 7500 --steps 40 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%.1f/s %.3f ms/step  fwd %.3f (%.1f TF) cost %.9f' % (d['value'], d['ms_per_step'], k['fgemm_fwd'], d['roofline']['achieved'], d['last_cost_per_image']))")
  echo "TNML_FG64_CFG=$cfg: $r"
done
