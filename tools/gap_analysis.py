"""idle gaps of the GPU timeline from a rocprofv3 --kernel-trace (+ --memory-copy-trace) run: python tools/gap_analysis.py <dir>
Prints, for the steady-state window (the last 20 k_sytrd launches), the busy time per kernel, the idle time per
(previous op -> next op) transition and the totals per bond update."""
import collections, csv, glob, sys
d = sys.argv[1]
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY_" + r.get("Direction", r.get("Name", "?"))))
ops.sort()
sy = [i for i, o in enumerate(ops) if "sytrd" in o[2]]
if len(sy) < 22:
    print("too few bond updates in the trace"); sys.exit(0)
lo, hi = sy[-21], sy[-1]
win = ops[lo:hi]
nb = 20
busy = collections.defaultdict(lambda: [0, 0])
gaps = collections.defaultdict(lambda: [0, 0])
tbusy = tgap = 0
prev_end = win[0][0]
prev_name = "-"
for s, e, nme in win:
    g = s - prev_end
    if g > 0:
        gaps[(prev_name, nme)][0] += g; gaps[(prev_name, nme)][1] += 1; tgap += g
    busy[nme][0] += e - s; busy[nme][1] += 1; tbusy += e - s
    if e > prev_end:
        prev_end = e; prev_name = nme
span = win[-1][1] - win[0][0]
print("window: %d bond updates, %.3f ms each; busy %.3f ms, idle %.3f ms per bond update; %d ops per bond update" % (nb, span / nb / 1e6, tbusy / nb / 1e6, tgap / nb / 1e6, len(win) // nb))
print("\nbusy time per op (us per bond update):")
for k, v in sorted(busy.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %-62s %6.1f x %8.1f us = %8.1f" % (k, v[1] / nb, v[0] / v[1] / 1e3, v[0] / nb / 1e3))
print("\nidle time per transition (us per bond update):")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %-45s -> %-45s %5.1f x %7.1f us = %7.1f" % (k[0][:45], k[1][:45], v[1] / nb, v[0] / v[1] / 1e3, v[0] / nb / 1e3))
