"""Latency of the cross-process one-shot all-reduce (ipc_comm.hip) at the headline payload -- [48 scalars | G] with G = 240 x 240 doubles
= 461 KB -- between PROCESSES that share the one GPU of a box: every rank runs bond updates of a 24-site chain at m = 120 on its image
shard and reports the event-timed 'allreduce' class of its profile (launches, mean us).  No second physical GPU is involved: the
number is the cost of the exchange kernel itself (peer stores, arrival flags, ordered sum), not of an xGMI link.
  python tools/time_oneshot_two_processes.py [nranks=2]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, nranks):
    import numpy as np
    from tnml_amd import lib, synth
    from tnml_amd.fixedl import TrainStates
    N, m, NT = 24, 120, int(os.environ.get("TNML_T1_NT", "2048"))
    labels = synth.synthetic_labels(NT); pixels = synth.synthetic_images(N, labels)
    lo, hi = lib.shard_bounds(NT, nranks, rank)
    ts = TrainStates(labels[lo:hi], N, m, pixels=pixels[lo:hi], device=0, rank=rank, nranks=nranks, NT_total=NT, dtype="f64")
    ts.set_option("comm_timeout_s", 30)
    print("HANDLE " + ts.oneshot_export().hex(), flush=True)
    ts.oneshot_connect([bytes.fromhex(x) for x in sys.stdin.readline().split()])
    ts.set_mps(synth.random_mps(N, m, seed=1)); ts.replica_check(); ts.init()
    import hashlib
    def hsh(x):
        return hashlib.sha1(np.ascontiguousarray(np.asarray(x, dtype=np.float64)).tobytes()).hexdigest()[:12]
    for bb in range(1, 8):
        ts.shiftE(bb, True)
    ts.setBond(8)
    B = ts.bond_tensor(8)
    gh = [hsh(ts.gradient(B)) for _ in range(40)]              # 40 all-reduces of 461 KB: every rank must hold the same bits each time
    print("RESULT " + json.dumps({"rank": rank, "what": "gradient", "hashes": gh}), flush=True)
    if os.environ.get("TNML_T1_REPAIR", "1") == "1":
        ts.set_option("check_replicas", 2)                     # a replica mismatch after a split is repaired (and counted), not an error
    b, ha = 8, 1
    wh = []
    ts.profile(True); ts.profile_reset()
    for _ in range(8):
        ts.bond_update(b, ha, m, m, 1e-10, 4, 1e-3, 1e-10)
        W = ts.get_mps()
        wh.append((b, hsh(W[b - 1]), hsh(W[b]), ts.svd_stats()["cluster_repairs"], ts.replica_repairs()))
        b, ha = lib.sweepnext(b, ha, N)
    ts.synchronize()
    pr = ts.profile_read()
    print("RESULT " + json.dumps({"rank": rank, "what": "sweep", "mode": ts.collective_mode(), "allreduce": pr.get("allreduce"), "W": wh}), flush=True)
    ts.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]))
    nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(nranks)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(nranks)]
    try:
        handles = []
        for p in procs:
            line = p.stdout.readline()
            while line and not line.startswith("HANDLE "):
                line = p.stdout.readline()
            if not line.startswith("HANDLE "):
                print(p.stderr.read()[-3000:]); return 1
            handles.append(line.split()[1])
        for p in procs:
            p.stdin.write(" ".join(handles) + "\n"); p.stdin.flush()
        outs = []
        for p in procs:
            so, se = p.communicate(timeout=300)
            res = [json.loads(ln[7:]) for ln in so.splitlines() if ln.startswith("RESULT ")]
            if p.returncode or len(res) < 2:
                print("rank failed:", se[-1500:])
            outs.append(res)
        if all(len(o) >= 1 for o in outs):
            g = [o[0]["hashes"] for o in outs]
            bad = [i for i in range(len(g[0])) if any(x[i] != g[0][i] for x in g)]
            print("all-reduced gradient, 40 evaluations: %s" % ("every rank holds the same bits each time" if not bad else "ranks DIFFER at evaluations %s" % bad))
            print("   same bits from one evaluation to the next on rank 0: %s" % (len(set(g[0])) == 1))
        if all(len(o) >= 2 for o in outs):
            for i in range(len(outs[0][1]["W"])):
                row = [o[1]["W"][i] for o in outs]
                same = all(r[1:3] == row[0][1:3] for r in row)
                print("bond %d: site tensors after the update %s (cluster repairs so far %s, replica repairs so far %s)" % (row[0][0], "identical on every rank" if same else "DIFFER: %s" % [r[1:3] for r in row], row[0][3], row[0][4]))
            for o in outs:
                n, ms = o[1]["allreduce"] if o[1]["allreduce"] else (0, 0.0)
                print("rank %d of %d (collective mode %d): %d all-reduces of 461 KB, %.1f us each (event-timed on the issuing stream)" % (o[1]["rank"], nranks, o[1]["mode"], n, 1e3 * ms / max(n, 1)))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return 0


if __name__ == "__main__":
    sys.exit(main())
