"""per-kernel summary of the PMC passes of tools/pmc_bench.sh (FETCH_SIZE, WRITE_SIZE, MFMA counters): python tools/pmc_summary.py <dir>"""
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
            print("      MFMA busy %.3e cyc (64.0 per v_mfma_f64_16x16x4: %.3e instructions = %.3f GFLOP), GUI active %.3e summed over the 8 XCDs = %.0f cycles of kernel time\n      -> matrix pipe busy %.1f %% of the time (busy / (gui_active / 8 * 1024 SIMDs))" % (mf[0], mf[1] * 512 / 2048, mf[1] * 512 / 1e9, mf[3], mf[3] / 8, 100 * mf[0] / (mf[3] / 8 * 1024)))

# ---- stamped traffic record for bench.py (roofline.traffic): bytes per launch of the kernels the bench line names, tagged with
# the kernel symbol, the sha of the kernel sources and the commit they were measured at; bench.py drops the number when the
# sources have changed since.  FETCH_SIZE is doubled (MI355X_MICROARCH.md "HBM": wide streaming reads are reported at half on
# gfx950; the label dot, whose byte count is known exactly, calibrates the factor), both counters are in KB.
import hashlib, json, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in ("kernels_gemm.hip", "kernels_stream.hip", "kernels_fused.hip", "kernels_res.hip", "kernels_grad.hip"):
    h.update(open(os.path.join(root, "tnml_amd", "csrc", f), "rb").read())
# (the first pattern that matches a kernel of the run: the resident-operand kernels of round 4 where they ran, else the tiled ones)
classes = {"fgemm_fwd": ["k_fgemm64<2, 5, 4, 3, 16, double, 2>"], "fgemm_shift": ["k_shift_res", "k_fgemm64<2, 4, 4, 2, 8, double, 1>"],
           "labeldot": ["k_labeldot<4, 2, 10, double, double, double"], "bgemm": ["k_grad_quad<0, 4>", "k_grad_quad<0", "k_bgemm64<5, 1, 3, 4, 1, double>"],
           "fwd_fused": ["k_fwd_res", "k_fwd_fused"]}
rec = {"kernels_src_sha16": h.hexdigest()[:16], "commit": os.environ.get("TNML_COMMIT", "unknown"),
       "workload": "bench.py default (BASELINE config 3, 60000 images, maxm 120, fp64)", "kernels": {}}
for cls, pats in classes.items():
    for pat in pats:
        hit = [(k, v) for k, v in res.items() if pat in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
        if hit:
            k, v = max(hit, key=lambda kv: kv[1]["FETCH_SIZE"][1])
            rec["kernels"][cls] = {"kernel": k, "bytes_per_launch": (2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024.0,
                                   "fetch_size_kb": v["FETCH_SIZE"][0], "write_size_kb": v["WRITE_SIZE"][0], "launches": v["FETCH_SIZE"][1]}
            break
# whole bond update: every dispatch between the k_tridiag_split launches (one per split) of consecutive bond updates, averaged over
# the last three bond updates of the run (the breakdown steps of bench.py: interior bonds with a Label-carrying shiftE)
def step_total(ctr):
    f = glob.glob(out + "/**/" + ctr + "_counter_collection.csv", recursive=True)
    if not f:
        return None
    rows = [(int(r.get("Dispatch_Id", 0) or 0), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == ctr]
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "k_teig_values" in r[1] or "k_tridiag_split" in r[1]]      # one per split
    if len(marks) < 5:
        return None
    a, b, n = marks[-4], marks[-1], 3
    return sum(r[2] for r in rows[a:b]) / n
fs, ws = step_total("FETCH_SIZE"), step_total("WRITE_SIZE")
if fs is not None and ws is not None:
    rec["step"] = {"bytes_per_step": (2.0 * fs + ws) * 1024.0, "fetch_size_kb_per_step": fs, "write_size_kb_per_step": ws, "steps_averaged": 3,
                   "source": "every dispatch of a bond update (between the first eigen-stage launches of consecutive splits), FETCH_SIZE x 2 + WRITE_SIZE, mean of the last three bond updates of the PMC run"}
json.dump(rec, open(out + "/pmc_traffic.json", "w"), indent=1)
if "step" in rec: print("whole bond update: %.1f MB" % (rec["step"]["bytes_per_step"] / 1e6))
print("wrote", out + "/pmc_traffic.json", {k: round(v["bytes_per_launch"] / 1e6, 1) for k, v in rec["kernels"].items()}, "MB per launch")
