#!/bin/bash
# HBM traffic per kernel launch from the PMC counters (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"):
# FETCH_SIZE and WRITE_SIZE in separate passes, per-kernel mean over the launches of a short bench run.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE MFMA; do
  pmc=$ctr; [ $ctr = MFMA ] && pmc="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out -o $ctr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --plain --steps 6 --warmup 2 "$@" > $out/$ctr.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $out
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +20M -delete
