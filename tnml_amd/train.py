"""Multi-GPU form of the `fixedL <inputfile>` driver: one process per GPU under torch.distributed.run.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m tnml_amd.train input_fixedL

Same input file, data directory, `sites` / `W` files (the formats of tnml_amd/host), `WRITE_WF` / `LAMBDA` hooks and log
lines as the C++ driver (tnml_amd/host/fixedl_main.cpp, which runs one GPU); the training images are sharded over the
ranks with tnml_shard_bounds and the gradient / cost sums go through the library's RCCL all-reduce
(BASELINE config 3: "batch sharded across 8xMI355X").  torch.distributed (gloo) is only the control plane: the RCCL
unique id, the LAMBDA hot reload and barriers.  Rank 0 prints and writes files.  A single process (no launcher)
works too and is what the tests run.
"""
import os
import sys

import numpy as np


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("Usage: python -m tnml_amd.train inputfile")                      # fixedL.cc:579-583
        return 0
    inp = argv[0]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from . import hostlib, lib
    from .fixedl import TrainStates

    def say(msg=""):
        if rank == 0:
            print(msg, flush=True)

    def key(name, default, conv=str):
        v = hostlib.input_get(inp, name)
        return default if v is None else conv(v)

    datadir = key("datadir", "/Users/mstoudenmire/software/tnml/mllib/MNIST")
    Ntrain = key("Ntrain", 60000, int)
    Nbatch = key("Nbatch", 10, int)
    Nsweep = key("Nsweep", 50, int)
    cutoff = key("cutoff", 1e-10, float)
    maxm = key("maxm", 5000, int)
    minm = key("minm", max(10, maxm // 2), int)
    ninitial = key("ninitial", 100, int)
    lam = key("lambda", 0.0, float)
    method = key("method", "conj")
    Npass = key("Npass", 4, int)
    cconv = key("cconv", 1e-10, float)
    seed = key("seed", 1, int)
    precision = key("precision", "f64")
    imglen = key("imglen", 0, int)
    feature_scale = key("feature_scale", 1.0, float)
    dtype = {"f64": "f64", "strict": "f64", "mixed": "f64_e32", "f32": "f32"}.get(precision)
    if dtype is None:
        say("precision must be f64, mixed or f32")
        return 1
    if method != "conj":
        say('method type "%s" not recognized' % method)                        # fixedL.cc:505
        return 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)

    px, lab, _ = hostlib.read_mnist(datadir, True, Ntrain)                      # fixedL.cc:613
    if imglen > 0:
        side = int(round(np.sqrt(px.shape[1])))
        vals = hostlib.reduce(px, side, imglen)
    else:
        vals = None
    N = (vals if vals is not None else px).shape[1]
    c = N // 2
    NT = len(lab)
    say("Training set consists of %d images:" % NT)
    for l in range(10):
        say("  %d of label %d" % (int((lab == l).sum()), l))
    say("%d sites of dimension 2" % N)
    if os.path.exists("sites"):                                                 # fixedL.cc:619-631
        Ns, ds = hostlib.read_sites("sites")
        if ds != 2 or Ns != N:
            say("Error: d=2 but the sites file describes %d sites of dimension %d" % (Ns, ds))
            return 1
    elif rank == 0:
        hostlib.write_sites("sites", N, 2)
    say("Converting training set to MPS")
    say("Total of %d training images" % NT)
    if NT % Nbatch != 0:                                                        # fixedL.cc:84-89
        say("totNtrain=%d, Nbatch=%d, totNtrain%%Nbatch=%d" % (NT, Nbatch, NT % Nbatch))
        say("totNtrain not commensurate with Nbatch")
        return 1
    lo, hi = lib.shard_bounds(NT, world, rank)
    for r in range(world):
        rlo, rhi = lib.shard_bounds(NT, world, r)
        say("Thread %d %d -> %d (%d)" % (r, rlo, rhi, rhi - rlo))               # fixedL.cc:94, one GPU per "thread"

    if rank == 0 and not os.path.exists("W"):                                   # fixedL.cc:702-727
        hostlib.build_initial_w(datadir, Ntrain, ninitial, seed, "W", imglen=imglen, feature_scale=feature_scale)
        say("Done making initial W")
    elif rank == 0:
        say("Reading W from disk")
    if world > 1:
        dist.barrier()
    W = hostlib.read_mps("W")
    if len(W) != N or W[c - 1].ndim != 4:
        say("Expected W to have Label type Index at site %d" % c)
        return 1
    wm = max(max(A.shape[0], A.shape[2]) for A in W)
    # `maxm` is only an upper bound for the reference (default 5000): size the context by what an N-site MPS can reach
    # and what fits this GPU for this shard, and say so
    ctx_maxm = lib.plan_maxm(N, hi - lo, maxm, floor_m=wm, dtype=dtype, device=local_rank)
    if world > 1:
        t = torch.tensor([ctx_maxm], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ctx_maxm = max(int(t[0]), wm)
    if ctx_maxm < maxm:
        say("maxm=%d is beyond what %d sites can reach or this GPU can hold for %d images: bond dimensions are capped at %d" % (maxm, N, hi - lo, ctx_maxm))

    if vals is None and feature_scale == 1.0:
        ts = TrainStates(lab[lo:hi], N, ctx_maxm, pixels=px[lo:hi], device=local_rank, rank=rank, nranks=world, NT_total=NT, dtype=dtype)
    else:
        g = (vals if vals is not None else px.astype(np.float64))[lo:hi] / 255.0
        phi = np.stack([np.ones_like(g), feature_scale * ((g / 255.0) / 4.0)], axis=-1)
        ts = TrainStates(lab[lo:hi], N, ctx_maxm, phi=phi, device=local_rank, rank=rank, nranks=world, NT_total=NT, dtype=dtype)
    if world > 1:
        uid = [TrainStates.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ts.comm_init(uid[0])
    ts.set_mps(W)
    say("Projecting training states...")
    ts.init()                                                                   # fixedL.cc:741
    ts.setBond(1)
    C0, _, _, nc = ts.quadcost(ts.bond_tensor(1), lam)                          # fixedL.cc:745
    say("Percent correct = %.4f%%, # incorrect = %d/%d" % (nc * 100.0 / NT, NT - nc, NT))
    say("Before starting DMRG Cost = %.10f" % (C0 / NT))

    lam_cost = lam                                                              # cargs copy, fixedL.cc:467
    for sw in range(1, Nsweep + 1):
        say("\nSweep %d maxm=%d minm=%d" % (sw, maxm, minm))
        b, ha = 1, 1
        while ha <= 2:
            r = ts.bond_update(b, ha, min(maxm, ctx_maxm), minm, cutoff, Npass, lam, cconv, lam_cost=lam_cost)
            if rank == 0:
                say("Sweep %d Half %d Bond %d" % (sw, ha, r["c"]))
                say("In cgrad, lambda = %.3E" % lam)
                cg = r["cg"]
                for p in range(cg["npass_done"]):
                    say("  Conj grad pass %d" % (p + 1))
                    if p < len(cg["cost"]):
                        say("  Cost = %.10f" % (cg["cost"][p] / NT))
                        say("  |r| = %.1E" % cg["rnorm"][p])
                say("Sweep %d Half %d Bond %d" % (sw, ha, r["c"]))
                say("SVD trunc err = %.2E" % r["truncerr"])
                say("Original m=%d, New m=%d" % (r["origm"], r["newm"]))
                say("norm(newB) = %.12g" % r["norm_newB"])
                say("|B-newB| = %.3E" % r["diff"])
                for l in range(10):
                    say("  Label l=%d C%d = %.10f" % (l, l, r["label_cost"][l] / NT))
                say("  Reg. cost CR = %.10f" % (r["reg_cost"] / NT))
                say("Percent correct = %.4f%%, # incorrect = %d/%d" % (r["ncorrect"] * 100.0 / NT, NT - r["ncorrect"], NT))
                say("--> After SVD, Cost = %.10f" % (r["cost"] / NT))
            # file hooks: rank 0 looks, every rank follows (fixedL.cc:542-559)
            hook = [None, None]
            if rank == 0:
                if os.path.exists("WRITE_WF"):
                    os.remove("WRITE_WF")
                    hook[0] = True
                if os.path.exists("LAMBDA"):
                    try:
                        hook[1] = float(open("LAMBDA").read().split()[0])
                    except (ValueError, IndexError):
                        hook[1] = None
                    os.remove("LAMBDA")
            if world > 1:
                dist.broadcast_object_list(hook, src=0)
            if hook[0]:
                say("File WRITE_WF found")
                say("Writing W to disk")
                Wnow = ts.get_mps()
                if rank == 0:
                    hostlib.write_mps("W", Wnow)
            if hook[1] is not None:
                lam = hook[1]
                say("new lambda = %g" % lam)
            b, ha = lib.sweepnext(b, ha, N)
        say("Writing W to disk")                                                # fixedL.cc:565
        Wnow = ts.get_mps()
        if rank == 0:
            hostlib.write_mps("W", Wnow)
    say("Writing W to disk")                                                    # fixedL.cc:763
    Wnow = ts.get_mps()
    if rank == 0:
        hostlib.write_mps("W", Wnow)
    ts.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
