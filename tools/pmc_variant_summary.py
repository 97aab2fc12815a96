"""tools/pmc_grad_variants.sh: per-kernel means of every collected counter, one row per kernel symbol (gradient-GEMM variants and their
ablations, the forward / shift kernels for reference, the Householder chain), plus the ratios that answer 'what saturates':
wave-cycle split (parked / issue-stalled / active), LDS conflict share, instructions by class per MFMA, TA / TCP stall share."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = ("k_bgemm64", "k_grad_", "k_fwd_res", "k_fwd_fused", "k_shift_res", "k_sytrd", "k_slab_reduce")
rows = [(k, d) for k, d in acc.items() if any(w in k for w in want)]
rows.sort(key=lambda kv: kv[0])
def m(d, n):
    v = d.get(n)
    return sum(v) / len(v) if v else float("nan")
for k, d in rows:
    n = max(len(v) for v in d.values())
    print("=" * 150)
    print("%s   (%d dispatches)" % (k[:140], n))
    names = sorted(d)
    for i in range(0, len(names), 4):
        print("   " + "   ".join("%-34s %12.5g" % (x, m(d, x)) for x in names[i:i + 4]))
    wc = m(d, "SQ_WAVE_CYCLES")
    if wc == wc and wc > 0:
        print("   -> of wave cycles: parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %% (LDS part %.1f %%), issuing %.1f %%"
              % (100 * m(d, "SQ_WAIT_ANY") / wc, 100 * m(d, "SQ_WAIT_INST_ANY") / wc, 100 * m(d, "SQ_WAIT_INST_LDS") / wc, 100 * m(d, "SQ_ACTIVE_INST_ANY") / wc))
    ia = m(d, "SQ_LDS_IDX_ACTIVE")
    if ia == ia and ia > 0:
        print("   -> LDS: bank-conflict cycles / active cycles %.1f %%; LDS active / SQ busy cycles %.1f %%" % (100 * m(d, "SQ_LDS_BANK_CONFLICT") / ia, 100 * ia / max(m(d, "SQ_BUSY_CYCLES"), 1)))
    mf = m(d, "SQ_INSTS_MFMA")
    if mf == mf and mf > 0:
        print("   -> per MFMA instruction: VALU %.2f, LDS %.2f, VMEM read %.3f, SALU %.2f" % (m(d, "SQ_INSTS_VALU") / mf, m(d, "SQ_INSTS_LDS") / mf, m(d, "SQ_INSTS_VMEM_RD") / mf, m(d, "SQ_INSTS_SALU") / mf))
    ga = m(d, "GRBM_GUI_ACTIVE")
    if ga == ga and ga > 0:
        tb = m(d, "TA_BUSY")
        if tb == tb:
            print("   -> TA: busy / GUI active (summed over instances) %.3g, address stalled by TC %.3g, data stalled by TC %.3g" % (tb / ga, m(d, "TA_ADDR_STALLED_BY_TC_CYCLES") / ga, m(d, "TA_DATA_STALLED_BY_TC_CYCLES") / ga))
        rq = m(d, "TCP_TCC_READ_REQ")
        if rq == rq and rq > 0:
            print("   -> TCP: mean L2 read latency %.0f cycles, pending-stall cycles / GUI active %.3g" % (m(d, "TCP_TCC_READ_REQ_LATENCY") / rq, m(d, "TCP_PENDING_STALL_CYCLES") / ga))
