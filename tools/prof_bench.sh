#!/bin/bash
# rocprofv3 kernel trace of a short bench run; summary -> gpurun_out/prof_<tag>/  (copy what matters into profiles/)
tag=${1:-run}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --plain "$@" > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-300
f=$(find $out -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-70s %8s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
for r in rows[:40]:
    print("%-70s %8s %10.3f %10.2f %6.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $out -name '*kernel_trace.csv' -delete
