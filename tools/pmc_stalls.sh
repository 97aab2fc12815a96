#!/bin/bash
# where the waves of each kernel spend their cycles (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ wait / issue counters
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_stalls
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o st -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > $out/st.log 2>&1
python - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/st_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES"]
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]["SQ_WAVE_CYCLES"]))
print("%-60s %6s " % ("kernel", "calls") + " ".join("%10s" % n[3:13] for n in names))
for k, d in rows[:12]:
    m = {n: (sum(d[n]) / len(d[n]) if d[n] else 0.) for n in names}
    wc = m["SQ_WAVE_CYCLES"] or 1.
    print("%-60s %6d " % (k[:60], len(d["SQ_WAVE_CYCLES"])) + " ".join("%10.3g" % m[n] for n in names))
    print("%-60s        wait_any %.1f%%  wait_inst %.1f%%  active %.1f%%  (of wave cycles)   lds conflict/active %.1f%%" % ("", 100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_ACTIVE_INST_ANY"] / wc, 100 * m["SQ_LDS_BANK_CONFLICT"] / (m["SQ_LDS_IDX_ACTIVE"] or 1.)))
PY
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +20M -delete
