"""One rank of the cross-process one-shot all-reduce test AT THE HEADLINE PAYLOAD (tests/test_multirank_one_gpu.py,
test_one_shot_allreduce_across_processes_at_m120): a process of its own, every rank on the one GPU of the box.  The buffer of a collective
is [48 scalars | G] with G = 240 x 240 doubles = 461 KB = 29 chunks of k_os_exchange (ipc_comm.hip) -- the multi-chunk path the m = 6 test
never enters.  The parent carries the IPC handles between the ranks and holds every rank at a start barrier (READY / GO lines) until all
of them have a context, their data on the device and their peers mapped, so that the ranks enter their first collective within
milliseconds of each other.
  argv: rank nranks NT nbonds repair(0|1)"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hsh(x):
    return hashlib.sha1(np.ascontiguousarray(np.asarray(x, dtype=np.float64)).tobytes()).hexdigest()[:16]


def main():
    rank, nranks, NT, nbonds, repair = (int(x) for x in sys.argv[1:6])
    t00 = time.time()

    def note(msg):
        print("[rank %d +%.2fs] %s" % (rank, time.time() - t00, msg), file=sys.stderr, flush=True)
    from tnml_amd import lib, synth
    from tnml_amd.fixedl import TrainStates
    N, m = 24, 120
    labels = synth.synthetic_labels(NT)
    pixels = synth.synthetic_images(N, labels)
    lo, hi = lib.shard_bounds(NT, nranks, rank)
    ts = TrainStates(labels[lo:hi], N, m, pixels=pixels[lo:hi], device=0, rank=rank, nranks=nranks, NT_total=NT, dtype="f64")
    ts.set_option("comm_timeout_s", int(os.environ.get("TNML_T_TIMEOUT", "60")))
    print("HANDLE " + ts.oneshot_export().hex(), flush=True)
    ts.oneshot_connect([bytes.fromhex(x) for x in sys.stdin.readline().split()])
    assert ts.collective_mode() == 4
    ts.set_mps(synth.random_mps(N, m, seed=1))
    note("context, data and peers ready")
    print("READY", flush=True)
    assert sys.stdin.readline().strip() == "GO"                  # the parent's start barrier
    n = ts.replica_check()
    ts.init()
    for bb in range(1, 8):
        ts.shiftE(bb, True)
    ts.setBond(8)
    B = ts.bond_tensor(8)
    note("environments built")
    gh = [hsh(ts.gradient(B)) for _ in range(6)]                 # all-reduces of 461 KB: every rank must hold the same bits each time
    note("gradients done")
    if repair:
        ts.set_option("check_replicas", 2)                       # a replica mismatch after a split is repaired (and counted), not an error
    b, ha = 8, 1
    costs, inflight = [], 0
    depth = int(os.environ.get("TNML_T_DEPTH", "2"))             # 2: pipelined like the sweep drivers (bond k + 1 is enqueued before the report of bond k is read)
    for k in range(nbonds):
        ts.bond_update_begin(b, ha, m, m, 1e-10, 4, 1e-3, 1e-10)
        inflight += 1
        if inflight == depth:
            costs.append(ts.bond_update_end()["cost"])
            inflight -= 1
            note("bond update %d reported" % len(costs))
        b, ha = lib.sweepnext(b, ha, N)
    while inflight:
        costs.append(ts.bond_update_end()["cost"])
        inflight -= 1
    note("%d bond updates done" % nbonds)
    wh = [hsh(A) for A in ts.get_mps()]
    ts.replica_check()
    ts.synchronize()
    a, bc = ts.collective_stats()
    print("RESULT " + json.dumps(dict(rank=rank, n=n, grad=gh, W=wh, cost=costs, allreduces=a, bcasts=bc, mem_kind=ts.oneshot_mem_kind() if hasattr(ts, "oneshot_mem_kind") else None,
                                      repairs=ts.replica_repairs(), svd=ts.svd_stats(), seconds=time.time() - t00)), flush=True)
    ts.close()


if __name__ == "__main__":
    main()
