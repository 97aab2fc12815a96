#!/bin/bash
# A/B of the k_fgemm64 tile configurations (TNML_FG64_CFG) at the m=120 interior-bond shape:
# parity of the m=120 test, then mean launch time from bench.py's HIP-event profile.
for cfg in ${CFGS:-0 1 2 3 4 5 6 7}; do
  export TNML_FG64_CFG=$cfg
  ok=$(python -m pytest tests/test_gpu_parity.py -q -k "m120" 2>&1 | tail -1)
  python bench.py --no-cpu-baseline --sites 48 --warmup 10 --steps 12 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('cfg $cfg | $ok | fgemm64 %.1f us/launch  %.1f TF (%.0f%%) | bond updates/s %.1f' % (1e3 * r['avg_launch_ms'], r['achieved'], 100 * r['frac'], d['value']))"
done
