#!/bin/bash
# label-carrying shift GEMM (k_fgemm64 shift form) tile configurations: kernel_ms_per_step.fgemm_shift of a short bench
for cfg in 0 1 2 3 4 5 6 7 8; do
  r=$(TNML_FG64_SHIFT_CFG=$cfg python bench.py --no-cpu-baseline --steps 16 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step']['fgemm_shift'], d['ms_per_step'])" 2>&1 | tail -1)
  echo "shift_cfg $cfg: fgemm_shift_ms ms_per_step = $r"
done
