#!/bin/bash
# kernel trace of tools/time_split.py at the given bond dimensions: which kernels a split is made of
# (from the repo root on a GPU box:  bash tools/prof_split.sh 320 400)
mkdir -p gpurun_out/prof_split
export TMPDIR=/tmp
root=$PWD
for m in "$@"; do
  out=$root/gpurun_out/prof_split/m$m
  rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $root/tools/time_split.py $m > $out/run.log 2>&1 < /dev/null)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== m $m"; grep "^m " $out/run.log
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("  %-78s %6s %10.3f ms %10.2f us %6.2f %%" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  else tail -5 $out/run.log; fi
  find $out -name '*kernel_trace.csv' -delete
done
