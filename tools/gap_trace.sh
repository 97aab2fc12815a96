#!/bin/bash
# kernel + memory-copy trace of a short bench window; tools/gap_analysis.py turns it into idle-gap statistics of the stream
tag=${1:-gaps}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --plain --steps 30 --warmup 5 "$@" > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-200
python $GRAFT_REPO_ROOT/tools/gap_analysis.py $out | tee $out/gap_summary.txt
find $out -name '*_trace.csv' -size +30M -delete
