"""world_size-2 CPU test (gloo) of the data-parallel decomposition used at --gpus N:
images are sharded with tnml_shard_bounds, every rank evaluates the gradient / cost / pAp partials
of its shard, the partials are summed by an all-reduce, and the replicated CG then runs in lock
step.  The per-shard arithmetic here is the CPU oracle (there is no GPU in this test), the
decomposition logic (shard bounds, packed [G | scalars] buffer, identical control flow on all
ranks) is the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import make_problem
    from oracle import pyoracle
    from tnml_amd import lib
    NT = 61                                          # ragged: the last rank takes the remainder
    pixels, labels, phi, W = make_problem(N=10, NT=NT, m=4, seed=5, pixel_boost=200.0)
    lo, hi = lib.shard_bounds(NT, world, rank)
    o = pyoracle.Oracle(phi[lo:hi], labels[lo:hi], W)
    o.init()
    full = pyoracle.Oracle(phi, labels, W)
    full.init()
    lam, npass = 1e-3, 3
    B = full.bond_tensor(1)

    def grad_allreduce(Bx):
        """packed [G | cost_l(10) | ncorrect | pp] buffer, one all-reduce"""
        G = o.gradient(Bx)
        C, lc, cr, nc = o.quadcost(Bx, 0.0)
        buf = torch.from_numpy(np.concatenate([G.ravel(order="F"), lc, [nc, 0.0]]))
        dist.all_reduce(buf)
        buf = buf.numpy()
        return buf[:G.size].reshape(G.shape, order="F"), buf[G.size:G.size + 10].sum()

    # replicated CG (fixedL.cc:349-445) on all-reduced quantities
    G, _ = grad_allreduce(B)
    r = G - lam * B
    p = r.copy()
    costs = []
    for ps in range(1, npass + 1):
        pp = torch.tensor([np.sum(o.forward(p) ** 2)], dtype=torch.float64)
        dist.all_reduce(pp)
        a = np.sum(r ** 2) / (float(pp[0]) + lam * np.sum(p ** 2))
        B = B + a * p
        if ps == npass:
            break
        G, csum = grad_allreduce(B)
        nr = G - lam * B
        beta = np.sum(nr ** 2) / np.sum(r ** 2)
        r = nr
        costs.append(csum + lam * np.sum(B ** 2))
        p = r + beta * p
    Bref, tr = full.cgrad(full.bond_tensor(1), npass, lam, 1e-10)
    # every rank holds bit-identical replicas after the all-reduce
    chk = torch.from_numpy(B.ravel().copy())
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    same = all(torch.equal(lst[0], t) for t in lst)
    q.put((rank, float(np.abs(B - Bref).max() / np.abs(Bref).max()), costs, tr["cost"], same, (lo, hi)))
    dist.destroy_process_group()


def test_sharded_cg_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[5] for r in res] == [(0, 30), (30, 61)]
    for rank, err, costs, ref_costs, same, _ in res:
        assert same, "replicas diverged after all-reduce"
        assert err < 1e-9                                      # partition independence
        np.testing.assert_allclose(costs, ref_costs, rtol=1e-10)
