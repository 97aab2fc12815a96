"""One rank of the cross-process one-shot all-reduce test (tests/test_multirank_one_gpu.py): a process of its own, all ranks on the
one GPU of the box, handles exchanged through the parent over pipes (stdin / stdout lines of hex)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, nranks, NT, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from conftest import make_problem
    from tnml_amd import lib
    from tnml_amd.fixedl import TrainStates, mldmrg
    N, m = 12, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)
    lo, hi = lib.shard_bounds(NT, nranks, rank)
    ts = TrainStates(labels[lo:hi], N, m, phi=phi[lo:hi], rank=rank, nranks=nranks, NT_total=NT)
    ts.set_option("comm_timeout_s", 20)
    h = ts.oneshot_export()
    print("HANDLE " + h.hex(), flush=True)
    handles = [bytes.fromhex(x) for x in sys.stdin.readline().split()]
    ts.oneshot_connect(handles)
    assert ts.collective_mode() == 4
    ts.set_mps(W)
    n = ts.replica_check()
    ts.init()
    B1 = ts.bond_tensor(1)
    G = ts.gradient(B1)
    C0 = ts.quadcost(B1, 1e-3)
    minm = m if mode == "spec" else m // 2
    reps = mldmrg(ts, 1, m, minm, 1e-10, 3, 1e-3, 1e-10, pipelined=True)
    ts.replica_check()
    ts.synchronize()
    a, b = ts.collective_stats()
    out = dict(n=n, G=G.tolist(), C0=C0[0] if isinstance(C0, (tuple, list)) else float(C0), cost=[r["cost"] for r in reps], newm=[r["newm"] for r in reps],
               ncorrect=[r["ncorrect"] for r in reps], W=[np.asarray(A).ravel().tolist() for A in ts.get_mps()], allreduces=a, bcasts=b)
    print("RESULT " + json.dumps(out), flush=True)
    ts.close()


if __name__ == "__main__":
    main()
