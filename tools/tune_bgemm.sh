#!/bin/bash
# A/B of the k_bgemm64 tile configurations (TNML_BG64_CFG) at the m=120 interior-bond shape
for cfg in ${CFGS:-0 1 2 3 4}; do
  export TNML_BG64_CFG=$cfg
  ok=$(python -m pytest tests/test_gpu_parity.py -q -k "m120" 2>&1 | tail -1)
  python bench.py --no-cpu-baseline --sites 48 --warmup 10 --steps 12 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('cfg $cfg | $ok | bgemm64 %.1f us/launch  slab_reduce %.1f us | bond updates/s %.1f' % (1e3 * k['bgemm'] / 4, 1e3 * k['slab_reduce'] / 4, d['value']))"
done
