#!/bin/bash
# feature-GEMM launch time against the image count (tile-round quantisation of the 128-image tiles over 256 CUs)
for ni in "$@"; do
  python bench.py --images $ni --no-cpu-baseline --steps 20 --warmup 4 2>&1 | tail -1 | NI=$ni python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']; ni=int(os.environ['NI']); k=d['kernel_ms_per_step']
print('images %6d  128-image tiles %4d  fgemm %.4f ms  %.1f TF  | step %.3f ms  bgemm %.3f labeldot %.3f' % (ni, (ni+255)//256*2, r['avg_launch_ms'], r['achieved'], d['ms_per_step'], k['bgemm'], k['labeldot']))"
done
