#!/bin/bash
# HBM traffic per kernel launch from the PMC counters (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"):
# FETCH_SIZE and WRITE_SIZE in separate passes, per-kernel mean over the launches of a short bench run.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE MFMA; do
  pmc=$ctr; [ $ctr = MFMA ] && pmc="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out -o $ctr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > $out/$ctr.log 2>&1
done
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/**/" + ctr + "_counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", ctr); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][ctr] = (sum(v) / len(v), len(v), max(v))
f = glob.glob(out + "/**/MFMA_counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            res[k][c] = (sum(v) / len(v), len(v), max(v))
print("%-72s %7s %14s %14s %14s" % ("kernel", "calls", "FETCH_mean", "FETCH_max", "WRITE_mean"))
rows = sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0, 0))[0] * kv[1].get("FETCH_SIZE", (0, 0, 0))[1])
with open(out + "/pmc_summary.csv", "w") as g:
    g.write("kernel,calls,fetch_size_mean,fetch_size_max,write_size_mean,mfma_busy_cycles_mean,mfma_mops_f64_mean,sq_busy_cycles_mean,grbm_gui_active_mean\n")
    for k, v in rows[:30]:
        fs = v.get("FETCH_SIZE", (0, 0, 0)); ws = v.get("WRITE_SIZE", (0, 0, 0))
        print("%-72s %7d %14.1f %14.1f %14.1f" % (k[:72], fs[1], fs[0], fs[2], ws[0]))
        mf = [v.get(c, (0, 0, 0))[0] for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")]
        g.write('"%s",%d,%.3f,%.3f,%.3f,%.1f,%.1f,%.1f,%.1f\n' % (k, fs[1], fs[0], fs[2], ws[0], mf[0], mf[1], mf[2], mf[3]))
        if mf[3] > 0 and mf[0] > 0:
            print("      MFMA busy %.3e cyc, MOPS_F64 %.3e, SQ busy %.3e, GUI active %.3e -> MfmaUtil %.1f %% (busy / (gui_active * 1024 SIMDs))" % (mf[0], mf[1], mf[2], mf[3], 100 * mf[0] / (mf[3] * 1024)))
PY
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +20M -delete
