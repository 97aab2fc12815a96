#!/usr/bin/env python
"""bench.py -- two-site bond updates/sec of the fixedL sweep on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE config 3 -- N=784 sites, maxm=120, 60 000 synthetic
MNIST-shaped images (no MNIST files offline), random-init weight MPS at bond dimension 120,
Npass=4, lambda=1e-3, cutoff=1e-10, minm=maxm (bonds stay at m=120), everything in fp64 like the reference.  One "step" = one
iteration of the mldmrg loop body (fixedL.cc:478-540): setBond + cgrad(Npass) + svd + quadcost + shiftE,
device resident.

Which bonds are timed.  A sweep has four kinds of interior bonds of equal GEMM cost: the Label index sits on
the right (b < N/2) or left (b > N/2) environment, and the environment built by shiftE after the update
carries the Label index (10x the work of a Label-free shift) on exactly half of them.  The default is one
whole sweep, 2(N-1) = 1566 consecutive bond updates (the exact sweep average: chain ends, interior and centre
bonds in their true proportions; 6-7 s).  With --steps < 2(N-1) the timed region is a run of consecutive
interior bonds of the MOST expensive kind (first half-sweep, b > N/2: every step pays the Label-carrying
shift), so a short run never overstates the sweep average.
At --gpus N the 60 000 images are sharded over N ranks (strong scaling) and the gradient / cost
partials are summed by an RCCL all-reduce inside the library; the control plane (unique-id
broadcast, barriers, max-over-ranks timing) uses torch.distributed.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TF = 78.6      # MI355X FP64 matrix peak (AMD spec); 77.4 TF measured (profiles/r01_probe64.txt)
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md "HBM3E peak BW"


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC summary (tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes over this same workload).  FETCH_SIZE is doubled as MI355X_MICROARCH.md "HBM"
    prescribes for wide streaming reads on gfx950; the label-dot kernel, whose byte count is known exactly
    (635.5 MB per launch), calibrates it: 310 398 KB reported = half.  Returns None when no summary is present."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.csv")
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith(kernel_prefix):
            return (2.0 * float(r["fetch_size_mean"]) + float(r["write_size_mean"])) * 1024.0
    return None


def cpu_baseline(maxm, npass, lam, cutoff, nthread, NT_total):
    """The CPU oracle (dense t.v restatement of fixedL.cc) timed on the host cores on a bounded
    sample: N=20 sites (a bond update costs O(NT m^2), independent of N -- SURVEY.md section 5),
    interior bonds 8..9 at the full bond dimension, NT_s images; the rate is scaled to NT_total."""
    from oracle import pyoracle
    from tnml_amd import synth
    N = 20
    NT_s = 240 if maxm >= 100 else 2000
    NT_s = (NT_s // nthread) * nthread
    labels = synth.synthetic_labels(NT_s, seed=7)
    pixels = synth.synthetic_images(N, labels, seed=7)
    phi = pyoracle.features_series(pixels)
    W = synth.random_mps(N, maxm, seed=1)
    o = pyoracle.Oracle(phi, labels, W, nthread=nthread, nbatch=1)
    o.init()
    b0 = 8
    for bb in range(1, b0):
        o.shiftE(bb, True)
    nb = 2
    t0 = time.time()
    for b in range(b0, b0 + nb):                      # the mldmrg loop body, fixedL.cc:482-540
        o.set_bond(b)
        B = o.bond_tensor(b)
        B, _ = o.cgrad(B, npass, lam, 1e-10)
        o.svd_split(B, b, 1, cutoff, maxm, maxm // 2)
        o.quadcost(o.bond_tensor(b), lam)
        o.shiftE(b, True)
    dt = time.time() - t0
    rate_sample = nb / dt
    return {
        "value": rate_sample * NT_s / NT_total,
        "unit": "bond updates/s",
        "cores": nthread,
        "kind": "port",
        "sample": "oracle (dense t.v fp64 restatement of fixedL.cc, %d threads): %d bond updates at m=%d on %d images took "
                  "%.2f s (%.3f bond updates/s); value = that rate x %d/%d (work per bond update is linear in the "
                  "image count)" % (nthread, nb, maxm, NT_s, dt, rate_sample, NT_s, NT_total),
    }


def hbm_roofline(prof_all, NTl, timed, args, world):
    """the HBM-bound kernel of the path: the label dot P_n[l] = sum_q T[q][n] * E[l][q][n] streams the Label-carrying
    environment (10*m values per image) and the GEMM output (m per image) once; algorithmic bytes per launch from the
    bond dimensions of the timed bonds, duration from the library's HIP events (class "labeldot" = the k_labeldot launches
    alone; the P update and the partial-sum reduction are class "p_update") over the untimed breakdown steps"""
    n_ld, ms_ld = prof_all.get("labeldot", (0, 0.0))
    if not n_ld or not timed:
        return None
    esz = 4 if args.dtype == "f32" else 8
    env_sz = 8 if args.dtype == "f64" else 4
    nl = 1 if args.single_label is not None else 10
    by = float(np.mean([NTl * (nl * min(r["mL"], r["mR"]) * (esz if r["label_on_B"] else env_sz) +
                              min(r["mL"], r["mR"]) * (env_sz if r["label_on_B"] else esz) + 4) for r in timed]))
    avg_ms = ms_ld / n_ld
    ach = by / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_labeldot", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic("void k_labeldot<4, 2, 10, double, double, double>") if args.dtype == "f64" and args.maxm == 120 and args.images == 60000 and world == 1 else None,
            "avg_launch_ms": avg_ms, "launches": n_ld, "bytes_per_launch": by}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: one whole sweep, 2(N-1) bond updates (6-7 s at the default workload)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sites", type=int, default=784)
    ap.add_argument("--images", type=int, default=60000)
    ap.add_argument("--maxm", type=int, default=120)
    ap.add_argument("--npass", type=int, default=4)
    ap.add_argument("--minm", type=int, default=None, help="default maxm: every interior bond stays at m = maxm whatever the "
                    "spectrum of the synthetic data (the reference default max(10, maxm/2) lets trained bonds shrink)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f64_e32", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-label", type=int, default=None, help="bench the per-label variant (single.cc, BASELINE config 4: one such "
                    "training per label, replicas only) for this label instead of the fixedL sweep")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 2 * (args.sites - 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    from tnml_amd import lib, synth
    from tnml_amd.fixedl import TrainStates

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    N, NT, maxm = args.sites, args.images, args.maxm
    lam, cutoff, cconv, npass = 1e-3, 1e-10, 1e-10, args.npass
    minm = args.minm if args.minm is not None else maxm

    labels = synth.synthetic_labels(NT)
    pixels = synth.synthetic_images(N, labels)
    W = synth.random_mps(N, maxm, seed=1)
    if args.single_label is not None:
        W[N // 2 - 1] = W[N // 2 - 1][..., 0] * 3.0             # plain weight MPS: no Label index
    lo, hi = lib.shard_bounds(NT, world, rank)
    ts = TrainStates(labels[lo:hi], N, maxm, pixels=pixels[lo:hi], device=local_rank, rank=rank, nranks=world,
                     NT_total=NT, dtype=args.dtype, single_label=args.single_label)
    del pixels
    if world > 1:
        uid = [TrainStates.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ts.comm_init(uid[0])
    ts.set_mps(W)
    t_init = time.time()
    ts.init()
    ts.synchronize()
    t_init = time.time() - t_init

    def sync():
        ts.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    b, ha = 1, 1
    full_sweeps = args.steps >= 2 * (N - 1)
    if not full_sweeps and N >= 64:
        # left environments up to the window start, as a sweep would have left them (setup, untimed)
        b0 = max(N // 2 + 8, min(N // 2 + 8 + (N // 2 - 24 - args.warmup - args.steps) // 2, N - 1))
        for bb in range(1, b0):
            ts.shiftE(bb, True)
        b = b0
    reports = []

    def step():
        nonlocal b, ha
        r = ts.bond_update(b, ha, maxm, minm, cutoff, npass, lam, cconv, report_costs=args.single_label is not None)
        reports.append(r)
        b, ha = lib.sweepnext(b, ha, N)
        if ha > 2:
            b, ha = 1, 1

    for _ in range(args.warmup):
        step()
    # timed region: HIP events only around the roofline kernel (two event records per launch cost host
    # time; timing all ~70 launches of a bond update would lower the very throughput being measured)
    ts.profile(os.environ.get("TNML_BENCH_NOPROF", "0") != "1", only="fgemm_fwd")
    ts.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    ts.profile(False)
    prof = ts.profile_read()
    # untimed extra steps with every kernel class timed: the per-class breakdown
    ts.profile(True)
    ts.profile_reset()
    nbreak = min(args.steps, 10)
    for _ in range(nbreak):
        step()
    ts.profile(False)
    prof_all = ts.profile_read()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    if rank == 0:
        timed = reports[args.warmup:args.warmup + args.steps]
        NTl = hi - lo
        # dominant kernel: the feature GEMM (T = X*B_mat).  Algorithmic flops per launch = SURVEY.md 8(d)
        # GEMM term 2*NT*(2mL)*(2mR) for the images one launch processes (x10 on the two Label-on-B bonds).
        n_fg, ms_fg = prof.get("fgemm_fwd", (0, 0.0))
        # every timed bond calls the feature GEMM 2*npass+1 times with its own (mL, mR); average the flops
        fl = []
        for r in timed:
            fl.append(2.0 * NTl * (2 * r["mL"]) * (2 * r["mR"]) * (10 if (r["label_on_B"] and args.single_label is None) else 1))
        flops_per_launch = float(np.mean(fl)) if fl else 0.0
        avg_ms = ms_fg / max(n_fg, 1)
        achieved_tf = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        peak = F64_MFMA_PEAK_TF if args.dtype != "f32" else F32_MFMA_PEAK_TF
        out = {
            "metric": "two-site bond updates/sec" if args.single_label is None else "two-site bond updates/sec (per-label variant, label %d)" % args.single_label,
            "value": args.steps / elapsed,
            "unit": "bond updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "fixedL N=%d, maxm=%d, %d images (BASELINE config 3), Npass=%d, lambda=%g, minm=%d, "
                                   "%s; timed bonds %d..%d (%s, m=%d)" % (N, maxm, NT, npass, lam, minm,
                                                                               {"f64": "fp64 throughout", "f64_e32": "fp64 MFMA over fp32-stored environments", "f32": "fp32 study mode"}[args.dtype],
                                                                               timed[0]["bond"] if timed else 0,
                                                                               timed[-1]["bond"] if timed else 0,
                                                                               "whole sweeps" if full_sweeps else
                                                                               "consecutive interior bonds, Label-carrying shiftE on each",
                                                                               maxm),
                       "global_images": NT, "sites": N, "maxm": maxm, "parallelism": "dp%d (image sharding + RCCL all-reduce)" % world},
            "roofline": {"bound": "mfma", "kernel": "k_fgemm64" if args.dtype != "f32" else "k_fgemm",
                         "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved_tf / peak,
                         "traffic": pmc_traffic("void k_fgemm64<2, 5, 4, 3, 16, 0, 0, double, 2>") if args.dtype == "f64" and maxm == 120 and NT == 60000 and world == 1 else None,
                         "avg_launch_ms": avg_ms, "launches": n_fg, "flops_per_launch": flops_per_launch},
            "kernel_ms_per_step": {k: v[1] / nbreak for k, v in prof_all.items() if v[0]},
            "roofline_hbm": hbm_roofline(prof_all, NTl, timed, args, world),
            "algebraic_shortcuts": [
                "fast CG: B*t.v is linear in B, so P <- P + a (p*t.v) replaces Npass-1 forward GEMMs per bond (TNML_FAST_CG=0 disables)",
                "the network outputs P_n do not depend on the bond they are evaluated at: the after-SVD quadcost of one bond update "
                "provides the residuals of the next one's first gradient, replacing 1 forward GEMM + label dot per bond (TNML_REUSE_P=0 disables)",
            ] if os.environ.get("TNML_FAST_CG", "1") != "0" or os.environ.get("TNML_REUSE_P", "1") != "0" else [],
            "env_init_s": t_init,
            "device_gb": ts.device_bytes() / 1e9,
            "last_cost_per_image": timed[-1]["cost"] / NT if timed else None,
            "svd_stats": ts.svd_stats(),
        }
        if world == 1 and not args.no_cpu_baseline and args.single_label is None:
            ncore = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(maxm, npass, lam, cutoff, min(16, ncore), NT)   # paralleldo.h:55-56 caps at 16
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ts.close()


if __name__ == "__main__":
    main()
