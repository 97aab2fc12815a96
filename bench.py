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

# The GPU boxes expose 256 hardware threads under a CPU quota of 16 (cgroup cpu.max): numpy / OpenBLAS / OpenMP pools sized
# by the thread count spin through that quota in the set-up phase and the kernel then throttles the whole process in
# 40-100 ms stalls -- which land inside a 20-step timed region (one stall = +50 % on ms_per_step).  Small host pools.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TF = 78.6      # MI355X FP64 matrix peak (AMD spec); 77.4 TF measured (profiles/r01_probe64.txt)
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md "HBM3E peak BW"


def kernel_source_sha16():
    """fingerprint of the kernel sources a PMC measurement belongs to"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels_gemm.hip", "kernels_stream.hip", "kernels_fused.hip", "kernels_res.hip", "kernels_grad.hip"):
        h.update(open(os.path.join(ROOT, "tnml_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_class):
    """HBM bytes per launch of a kernel class from the newest committed PMC record (profiles/r*_pmc_traffic.json, written
    by tools/pmc_summary.py from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over this same workload; FETCH_SIZE
    doubled as MI355X_MICROARCH.md "HBM" prescribes for wide streaming reads on gfx950 -- the label dot, whose byte count
    is known exactly, calibrates the factor).  The record carries the kernel symbol and the sha of the kernel sources it
    was taken at: when the sources have changed since, the number is stale and None is returned."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        rec = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None, None
    if rec.get("kernels_src_sha16") != kernel_source_sha16():
        return None, None
    e = rec.get("kernels", {}).get(kernel_class)
    if not e:
        return None, None
    return e["bytes_per_launch"], {"kernel": e["kernel"], "commit": rec.get("commit"), "kernels_src_sha16": rec.get("kernels_src_sha16"),
                                   "file": "profiles/" + os.path.basename(files[-1])}


def pmc_step_traffic():
    """HBM bytes of one whole bond update from the same PMC record (every dispatch between the splits of consecutive bond updates,
    FETCH_SIZE x 2 + WRITE_SIZE summed); None when the record is stale or has no such entry"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        rec = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    if rec.get("kernels_src_sha16") != kernel_source_sha16():
        return None
    return rec.get("step")


def cpu_quota():
    """CPUs this process may use: the cgroup quota when there is one, else the hardware thread count"""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(maxm, npass, lam, cutoff, nthread, NT_total, full=False):
    """The CPU oracle (dense t.v restatement of fixedL.cc, paralleldo.h chunking, `nthread` <= 16 threads) timed on the host
    cores on a bounded sample of the same workload: N=20 sites (a bond update costs O(NT m^2), independent of N --
    SURVEY.md section 5), consecutive interior bonds 8, 9, ... at the full bond dimension on NT_s images; the rate is scaled
    to NT_total (work per bond update is linear in the image count).  Default: NT_s = 960 (480 for m > 200) and as many
    bond updates as fit ~20 s (>= 2); full=True: the SURVEY.md 8(d) sample, NT_s = 2000 and 20 bonds (minutes)."""
    from oracle import pyoracle
    from tnml_amd import synth
    if full:
        N, NT_s, nb_max, budget = 44, 2000, 20, 1e9
    else:
        N, NT_s, nb_max, budget = 36, (960 if maxm <= 200 else 240) if maxm >= 100 else 4000, 20, 18.0
    NT_s = (NT_s // nthread) * nthread
    labels = synth.synthetic_labels(NT_s, seed=7)
    pixels = synth.synthetic_images(N, labels, seed=7)
    phi = pyoracle.features_series(pixels)
    W = synth.random_mps(N, maxm, seed=1)
    o = pyoracle.Oracle(phi, labels, W, nthread=nthread, nbatch=1)
    o.init()
    b0 = 8                                             # bonds 8.. are m x m on both sides for m <= 128
    for bb in range(1, b0):
        o.shiftE(bb, True)
    nb = 0
    t0 = time.time()
    b = b0
    while nb < nb_max and (nb < 2 or time.time() - t0 < budget) and b < N - 8:
        o.set_bond(b)                                  # the mldmrg loop body, fixedL.cc:482-540
        B = o.bond_tensor(b)
        B, _ = o.cgrad(B, npass, lam, 1e-10)
        o.svd_split(B, b, 1, cutoff, maxm, maxm // 2)
        o.quadcost(o.bond_tensor(b), lam)
        o.shiftE(b, True)
        nb += 1
        b += 1
    dt = time.time() - t0
    rate_sample = nb / dt
    return {
        "value": rate_sample * NT_s / NT_total,
        "unit": "bond updates/s",
        "cores": nthread,
        "cpu_model": cpu_model(),
        "host_cores": os.cpu_count(),
        "host_cpu_quota": cpu_quota(),
        "kind": "port",
        "image_bond_updates_per_s": rate_sample * NT_s,
        "sample": "oracle (dense t.v fp64 restatement of fixedL.cc, %d threads): %d consecutive bond updates (bonds %d..%d of a %d-site chain, "
                  "Label on RE / on B) at m=%d on %d images took %.2f s (%.4f bond updates/s = %.1f image-bond-updates/s); value = that rate "
                  "x %d/%d (work per bond update is linear in the image count)" % (nthread, nb, b0, b0 + nb - 1, N, maxm, NT_s, dt, rate_sample,
                                                                                 rate_sample * NT_s, NT_s, NT_total),
    }


def hbm_roofline(prof_all, NTl, timed, args, world):
    """the HBM-bound kernel of the path: the label dot P_n[l] = sum_q T[q][n] * E[l][q][n] streams the Label-carrying
    environment (10*m values per image) and the GEMM output (m per image) once; algorithmic bytes per launch from the
    bond dimensions of the timed bonds, duration from the library's HIP events (class "labeldot" = the k_labeldot launches
    alone; the P update and the partial-sum reduction are class "p_update") over the untimed breakdown steps"""
    n_ld, ms_ld = prof_all.get("labeldot", (0, 0.0))
    n_ff, ms_ff = prof_all.get("fwd_fused", (0, 0.0))
    if n_ff > n_ld and timed:
        # the label dot runs inside k_fwd_fused: per launch it streams the Label-carrying environment (10 m values per image) and the
        # Label-free one (m per image) once; the GEMM output never goes to HBM.  MFMA-bound kernel, so this is its HBM share.
        by = float(np.mean([NTl * (11 * min(r["mL"], r["mR"]) * 8 + 2 * 2 * 8 + 4) for r in timed]))
        avg_ms = ms_ff / n_ff
        ach = by / (avg_ms * 1e-3) / 1e9
        tr, src = pmc_traffic("fwd_fused") if args.dtype == "f64" and args.maxm == 120 and args.images == 60000 and world == 1 else (None, None)
        return {"bound": "hbm", "kernel": FWD_KERNEL[0] + " (environment streams beside the feature GEMM)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src, "avg_launch_ms": avg_ms, "launches": n_ff, "bytes_per_launch": by}
    if not n_ld or not timed:
        return None
    esz = 8 if args.dtype in ("f64", "f64_e32") else 4
    env_sz = 8 if args.dtype == "f64" else 4
    nl = 1 if args.single_label is not None else 10
    by = float(np.mean([NTl * (nl * min(r["mL"], r["mR"]) * (esz if r["label_on_B"] else env_sz) +
                              min(r["mL"], r["mR"]) * (env_sz if r["label_on_B"] else esz) + 4) for r in timed]))
    avg_ms = ms_ld / n_ld
    ach = by / (avg_ms * 1e-3) / 1e9
    tr, src = pmc_traffic("labeldot") if args.dtype == "f64" and args.maxm == 120 and args.images == 60000 and world == 1 else (None, None)
    return {"bound": "hbm", "kernel": "k_labeldot", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": tr, "traffic_source": src, "avg_launch_ms": avg_ms, "launches": n_ld, "bytes_per_launch": by}


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: re-run this command as N ranks under torch.distributed.run on this
    node (127.0.0.1 rendezvous on a free port) and hand its output through -- rank 0's JSON line is the result."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def baseline_config_name(N, maxm, NT, world, dtype):
    """which of BASELINE.json's configs the run is (or is a share / a variation of)"""
    if N == 784 and maxm == 120 and NT == 60000:
        return "BASELINE config 3" + ("" if world == 8 else ": its 60 000 images on %d GPU%s instead of 8" % (world, "" if world == 1 else "s"))
    if N == 784 and maxm == 300:
        return "BASELINE config 5 (maxm = 300 tolerance study%s)" % ("" if NT == 60000 else ": %d of its 60 000 images" % NT)
    if N == 784 and maxm == 120:
        return "one rank's share of BASELINE config 3: %d of its 60 000 images" % NT if NT * 8 == 60000 else "BASELINE config 3 shape with %d images" % NT
    if N == 784 and maxm == 20 and NT == 10000:
        return "BASELINE config 2"
    if N == 196 and maxm == 10 and NT == 1000:
        return "BASELINE config 1"
    return "not a BASELINE configuration"


def shift_flops(r, NTl, N, single):
    """flops of the shiftE that ends bond update r: 2 NT (2 m_in) m_out, x10 when the new environment carries the Label index"""
    c0 = -1 if single else N // 2
    b, ha = r["bond"], r["half"]
    if ha == 1:
        cs, m_in, m_out = b, r["mL"], r["newm"]
        lab = (not single) and cs >= c0
    else:
        cs, m_in, m_out = b + 1, r["mR"], r["newm"]
        lab = (not single) and cs <= c0
    return 2.0 * NTl * (2 * m_in) * m_out * (10 if lab else 1)


FWD_KERNEL = ["k_fwd_fused"]
GRAD_KERNEL = ["k_bgemm64 (gradient GEMM dP*dag(t.v), Z built while staging)"]


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def second_transport_leg(transport, port, rank, local_rank, world, timeout_s):
    """This rank's share of the SECOND transport's measurement: the same command as a child process (own rendezvous on `port`, the same
    RANK / LOCAL_RANK / WORLD_SIZE), --plain.  Returns (json line of the child's rank 0 or None, error text or None).  A child that dies,
    hangs past `timeout_s` or prints nothing costs an error string, never the parent's line."""
    import subprocess
    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a in ("--allreduce",):
            skip = True
            continue
        if a.startswith("--allreduce=") or a in ("--leg", "--plain", "--no-cpu-baseline"):
            continue
        argv.append(a)
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--allreduce", transport, "--leg", "--plain", "--no-cpu-baseline"]
    env = dict(os.environ)
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)                                         # the child's rank 0 hosts its own store on `port`
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return None, "the %s leg did not finish within %d s (rank %d)" % (transport, timeout_s, rank)
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    if run.returncode != 0:
        return None, "the %s leg ended with code %d on rank %d: %s" % (transport, run.returncode, rank, (run.stderr or run.stdout)[-600:].replace("\n", " | "))
    return (lines[-1] if lines else None), None


def n1_reference(N, maxm, NT, dtype):
    """the newest committed one-GPU line of the same workload and window (profiles/r*_bench_driver_form*.json): what `speedup_vs_n1` divides by"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_driver_form*.json")), reverse=True):
        try:
            d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
        except (OSError, ValueError, IndexError):
            continue
        c = d.get("config", {})
        if d.get("n_gpus") == 1 and d.get("dtype") == dtype and c.get("sites") == N and c.get("maxm") == maxm and c.get("global_images") == NT:
            return d, "profiles/" + os.path.basename(f)
    return None, None


def read_prof(ts):
    """per-class (launches, ms); the one-launch forward pass is reported under 'fwd_fused' whichever kernel ran it (k_fwd_fused, or
    k_fwd_res + its k_pfinish epilogue launch which the library books under 'p_update')"""
    pr = ts.profile_read()
    n, ms = pr.pop("fwd_res", (0, 0.0))
    if n:
        FWD_KERNEL[0] = "k_fwd_res"
        n0, ms0 = pr.get("fwd_fused", (0, 0.0))
        pr["fwd_fused"] = (n0 + n, ms0 + ms)
    n, ms = pr.pop("grad_quad", (0, 0.0))                     # the gradient GEMM is reported under 'bgemm' whichever kernel ran it
    if n:
        GRAD_KERNEL[0] = "k_grad_quad (gradient GEMM dP*dag(t.v): accumulators resident in a quad of workgroups, four MFMA waves per SIMD)"
        n0, ms0 = pr.get("bgemm", (0, 0.0))
        pr["bgemm"] = (n0 + n, ms0 + ms)
    return pr


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
    ap.add_argument("--dtype", default="f64", choices=["f64", "f64_e32", "f32", "bf16x3", "bf16"], help="bf16 / bf16x3: BASELINE config 5's study modes (forward feature GEMM on the bf16 matrix pipe)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="the SURVEY.md 8(d) CPU sample (2000 images, 20 bonds: minutes) instead of the bounded one")
    ap.add_argument("--literal-steps", type=int, default=None, help="bond updates timed in the reference's literal evaluation order "
                    "(fast CG and carried outputs off) after the main window; default min(steps, 100), 0 = skip")
    ap.add_argument("--workload", default="default", choices=["default", "8d"], help="8d: SURVEY.md 8(d) literally -- one untimed warm-up "
                    "sweep with minm = maxm/2 from the random-init W, then the timed bonds with minm = maxm/2")
    ap.add_argument("--single-label", type=int, default=None, help="bench the per-label variant (single.cc, BASELINE config 4: one such "
                    "training per label, replicas only) for this label instead of the fixedL sweep")
    ap.add_argument("--no-extras", action="store_true", help="skip the two secondary measurements of the default line: the two Label-on-B centre bonds "
                    "(centre_bond_ms) and the SURVEY.md 8(d) workload (value_8d)")
    ap.add_argument("--plain", action="store_true", help="main window + breakdown steps only (no literal-order, unfused-forward, centre-bond or 8(d) "
                    "measurements): what the profiler runs of tools/*.sh use, so that the last bond updates of the run are ordinary ones")
    ap.add_argument("--env-budget-gb", type=float, default=0.0, help="cap on the environment slabs held in HBM; what does not fit spills to host memory "
                    "(tnml_set_option env_budget_mb; 0: everything resident, the headline configuration)")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "oneshot"], help="transport of the library's collectives at --gpus N > 1: RCCL (ncclAllReduce / "
                    "ncclBroadcast), or the one-shot all-reduce across processes (IPC-mapped receive regions, device-side arrival flags, no host barrier: ipc_comm.hip)")
    ap.add_argument("--dry-run", action="store_true", help="control plane only (launcher, rendezvous, shard bounds, max-over-ranks clock): no GPU work")
    ap.add_argument("--no-second-transport", action="store_true", help="at --gpus N > 1 the same window is timed over BOTH transports in one invocation (the one "
                    "--allreduce names gives `value`; the other runs afterwards in a child process per rank, so that a failure there cannot "
                    "take the line with it) and reported side by side under `collectives`; this switch skips the second one")
    ap.add_argument("--leg", action="store_true", help=argparse.SUPPRESS)          # internal: this process is the second-transport child of a rank
    ap.add_argument("--share-device", action="store_true", help="test vehicle for one-GPU boxes: every rank uses HIP device 0 (RCCL refuses two ranks on one "
                    "device, so its block carries that error and `value` comes from the one-shot transport)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 2 * (args.sites - 1)
    if args.plain:
        args.no_extras = True
        args.literal_steps = 0

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch
    import torch.distributed as dist
    from tnml_amd import lib, synth
    torch.set_num_threads(min(8, torch.get_num_threads()))

    N, NT, maxm = args.sites, args.images, args.maxm
    lam, cutoff, cconv, npass = 1e-3, 1e-10, 1e-10, args.npass
    minm = args.minm if args.minm is not None else (maxm // 2 if args.workload == "8d" else maxm)
    lo, hi = lib.shard_bounds(NT, world, rank)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.dry_run:
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        cnt = torch.tensor([hi - lo], dtype=torch.int64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt)
        # the handle exchange of --allreduce oneshot runs over the same control plane: exercise it with placeholder handles
        gathered = None
        if args.allreduce == "oneshot" and world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, bytes([rank]) * 64)
            assert [g[0] for g in gathered] == list(range(world))
        # the second transport's leg (a child process per rank with a rendezvous of its own) runs over the same control plane
        second = None
        if world > 1 and not args.leg and not args.no_second_transport:
            other = "oneshot" if args.allreduce == "rccl" else "rccl"
            port = [free_port() if rank == 0 else None]
            dist.broadcast_object_list(port, src=0)
            line, err = second_transport_leg(other, port[0], rank, local_rank, world, 300)
            errs = [None] * world
            dist.all_gather_object(errs, err)
            if rank == 0:
                bad = [e for e in errs if e]
                second = {"transport": other, "error": bad[0]} if bad or not line else {"transport": other, "n_gpus": json.loads(line)["n_gpus"],
                                                                                         "handles_gathered": json.loads(line)["handles_gathered"]}
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_over_ranks": float(t[0]), "images_over_ranks": int(cnt[0]),
                              "shard_of_rank0": [lo, hi], "allreduce": args.allreduce if world > 1 else "none",
                              "handles_gathered": None if gathered is None else len(gathered), "second_transport": second}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from tnml_amd.fixedl import TrainStates
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = 0 if args.share_device else local_rank
    if dev >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs HIP device %d but only %d are visible" % (rank, dev, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    main_tr = args.allreduce
    if args.share_device and world > 1 and not args.leg:
        main_tr = "oneshot"                                      # RCCL refuses two ranks on one device: the test vehicle times the one-shot transport

    labels = synth.synthetic_labels(NT)
    pixels = synth.synthetic_images(N, labels)
    W = synth.random_mps(N, maxm, seed=1)
    single = args.single_label is not None
    if single:
        W[N // 2 - 1] = W[N // 2 - 1][..., 0] * 3.0             # plain weight MPS: no Label index
    ts = TrainStates(labels[lo:hi], N, maxm, pixels=pixels[lo:hi], device=dev, rank=rank, nranks=world,
                     NT_total=NT, dtype=args.dtype, single_label=args.single_label)
    del pixels
    comm_ranks = 1
    if world > 1 and main_tr == "oneshot":
        handles = [None] * world                                 # every rank's IPC handle, in rank order, over the gloo control plane
        dist.all_gather_object(handles, ts.oneshot_export())
        ts.oneshot_connect(handles)
    elif world > 1:
        uid = [TrainStates.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ts.comm_init(uid[0])
    if args.env_budget_gb > 0:
        ts.set_option("env_budget_mb", int(args.env_budget_gb * 1024))
    ts.set_mps(W)
    if world > 1:
        comm_ranks = ts.replica_check()                          # ncclCommCount == world and bit-identical W replicas
        assert comm_ranks == world, (comm_ranks, world)
        pass                                                     # replica check: mode 1 (folded into the packed all-reduce; a mismatch is an error)
    t_init = time.time()
    ts.init()
    ts.synchronize()
    t_init = time.time() - t_init

    def sync():
        drain()
        ts.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    b, ha = 1, 1
    reports = []

    inflight = 0
    pipelined = os.environ.get("TNML_BENCH_PIPELINE", "1") != "0"

    def step():
        """one bond update; pipelined: bond k+1 is enqueued before the report of bond k is fetched, as the sweep drivers do"""
        nonlocal b, ha, inflight
        if pipelined:
            ts.bond_update_begin(b, ha, maxm, minm, cutoff, npass, lam, cconv, report_costs=single)
            inflight += 1
            if inflight == 2:
                reports.append(ts.bond_update_end())
                inflight -= 1
        else:
            reports.append(ts.bond_update(b, ha, maxm, minm, cutoff, npass, lam, cconv, report_costs=single))
        b, ha = lib.sweepnext(b, ha, N)
        if ha > 2:
            b, ha = 1, 1

    def drain():
        nonlocal inflight
        while inflight:
            reports.append(ts.bond_update_end())
            inflight -= 1

    full_sweeps = args.steps >= 2 * (N - 1)
    if args.workload == "8d":
        for _ in range(2 * (N - 1)):                             # the warm-up sweep of SURVEY.md 8(d): bonds grow to their trained size
            step()
        drain()
        reports.clear()
        if not full_sweeps and N >= 64:                          # timed window: consecutive interior bonds of the second sweep, b < N/2
            for _ in range(max(8, min(N // 4, N // 2 - 8 - args.warmup - args.steps))):
                step()
            drain()
            reports.clear()
    elif not full_sweeps and N >= 64:
        # left environments up to the window start, as a sweep would have left them (setup, untimed)
        b0 = max(N // 2 + 8, min(N // 2 + 8 + (N // 2 - 24 - args.warmup - args.steps) // 2, N - 1))
        for bb in range(1, b0):
            ts.shiftE(bb, True)
        b = b0

    for _ in range(args.warmup):
        step()
    # timed region: HIP events only around the roofline kernel (two event records per launch cost host
    # time; timing all ~70 launches of a bond update would lower the very throughput being measured)
    # (the split is one event pair per bond update around the whole of svd_split: it costs nothing and, unlike the other classes, its time
    # depends on WHICH bonds are timed -- the rank-adaptive tridiagonalisation forms ~131 reflectors on bonds whose neighbours are still
    # random-init and 14-27 later in a long window -- so it is taken inside the timed region, not on the breakdown steps after it)
    # (the gradient GEMM and the shift are timed live as well -- roofline_kernels -- where a bond update is GPU-bound; on toy workloads, where
    # it is bound by the host's launch rate, ten more event records per bond update would move the figure being measured)
    live = "fgemm_fwd,fwd_fused,fwd_res,svd" + (",bgemm,grad_quad,fgemm_shift" if (hi - lo) * maxm * maxm >= 1e8 else "")
    ts.profile(os.environ.get("TNML_BENCH_NOPROF", "0") != "1", only=live)
    ts.profile_reset()
    # no cyclic garbage collection inside the timed region: a generation-2 pass over this process's heap takes ~40 ms -- the time of
    # 30 bond updates of an 8-GPU shard -- and where it lands depends on the allocation count (seen as one 40 ms step in an otherwise
    # flat 1.4 ms series, profiles/r03_gc_pause_in_timed_region.txt)
    import gc
    gc.collect()
    gc.disable()
    sync()
    coll0 = ts.collective_stats()
    split0 = ts.split_stats()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        step()
        if os.environ.get("TNML_BENCH_STEPTIMES"):
            step_marks.append(time.perf_counter() - t0)
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    coll1 = ts.collective_stats()
    split1 = ts.split_stats()
    if step_marks and rank == 0:
        print("host time at the end of each timed step (ms):", " ".join("%.2f" % (1e3 * t) for t in step_marks), "| total %.2f" % (1e3 * elapsed), file=sys.stderr)
    ts.profile(False)
    prof = read_prof(ts)
    n_timed_end = len(reports)                                   # sync() has drained the pipeline: every timed report is in
    # untimed extra steps with every kernel class timed: the per-class breakdown.  After whole sweeps the next bonds are the
    # chain start (bond dimensions 2, 4, 8, ...): move on to interior bonds first
    if full_sweeps and N >= 64:
        for _ in range(N // 4):
            step()
        drain()
    ts.profile(True)
    ts.profile_reset()
    nbreak = min(args.steps, 10)
    for _ in range(nbreak):
        step()
    drain()
    ts.profile(False)
    prof_all = read_prof(ts)
    # the reference's literal evaluation order (every forward pass of fixedL.cc:374-421 executed), timed the same way
    nlit = min(args.steps, 100) if args.literal_steps is None else args.literal_steps
    elapsed_lit = None
    lit_same_bonds = False
    if nlit > 0:
        drain()
        if not full_sweeps and N >= 64 and args.workload == "default" and args.literal_steps is None:
            # the SAME bonds as `value`: W and the environments back to where the main window started, the same warm-up, the same count
            ts.set_mps(W)
            ts.init()
            for bb in range(1, b0):
                ts.shiftE(bb, True)
            b, ha = b0, 1
            nlit = args.steps
            lit_same_bonds = True
        ts.set_option("fast_cg", 0)
        ts.set_option("reuse_p", 0)
        for _ in range(args.warmup if lit_same_bonds else 1):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(nlit):
            step()
        sync()
        elapsed_lit = time.perf_counter() - t1
        ts.set_option("fast_cg", 1)
        ts.set_option("reuse_p", 1)
    # the two kernels k_fwd_fused replaces, timed in the same run (untimed steps): stand-alone feature GEMM and label dot
    unfused = None
    if prof_all.get("fwd_fused", (0, 0))[0] > 0 and not args.plain:
        drain()
        ts.set_option("fused_fwd", 0)
        ts.set_option("fwd_res", 0)
        ts.profile(True, only="fgemm_fwd,labeldot")
        ts.profile_reset()
        for _ in range(nbreak):
            step()
        drain()
        ts.profile(False)
        unfused = ts.profile_read()
        ts.set_option("fused_fwd", 1)
        ts.set_option("fwd_res", 1)
    # ---- secondary measurements of the default line: the two Label-on-B centre bonds, and the 8(d) workload itself
    centre_ms = None
    rate_8d = None
    if not args.no_extras and not single and N >= 64 and args.workload == "default":
        drain()
        c0 = N // 2
        for attempt in range(2):                                 # the first round loads the Label-on-B kernel instantiations; the second is timed
            ts.set_mps(W)
            ts.init()
            for bb in range(1, c0 - 1):
                ts.shiftE(bb, True)
            ts.bond_update(c0 - 2, 1, maxm, minm, cutoff, npass, lam, cconv)        # brings P/dP up to date like a sweep arriving here
            cm = []
            for bc in (c0 - 1, c0):                              # Label on B: c0 = b + 1, then c0 = b (fixedL.cc:482-496)
                sync()
                tc = time.perf_counter()
                rc_ = ts.bond_update(bc, 1, maxm, minm, cutoff, npass, lam, cconv)
                ts.synchronize()
                cm.append(1e3 * (time.perf_counter() - tc))
                assert rc_["label_on_B"]
            centre_ms = cm
        if world > 1:
            tcm = torch.tensor(centre_ms, dtype=torch.float64)
            dist.all_reduce(tcm, op=dist.ReduceOp.MAX)
            centre_ms = [float(x) for x in tcm]
        # SURVEY.md 8(d) literally: one untimed warm-up sweep with minm = maxm/2 from the random-init W, then interior bonds of the second sweep
        ts.set_mps(W)
        ts.init()
        b, ha = 1, 1
        minm_main, minm = minm, maxm // 2
        for _ in range(2 * (N - 1) + N // 4):
            step()
        sync()
        n8 = 60
        t8 = time.perf_counter()
        for _ in range(n8):
            step()
        sync()
        rate_8d = n8 / (time.perf_counter() - t8)
        minm = minm_main
        if world > 1:
            t8t = torch.tensor([rate_8d], dtype=torch.float64)
            dist.all_reduce(t8t, op=dist.ReduceOp.MIN)
            rate_8d = float(t8t[0])
    if world > 1:
        t = torch.tensor([elapsed, elapsed_lit or 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        elapsed_lit = float(t[1]) if nlit > 0 else None

    out = None
    shard_sizes = [hi - lo]
    if world > 1:
        shard_sizes = [None] * world
        dist.all_gather_object(shard_sizes, hi - lo)
    if rank == 0:
        timed = reports[n_timed_end - args.steps:n_timed_end]
        NTl = hi - lo
        # dominant kernel: the feature GEMM (T = X*B_mat).  Algorithmic flops per launch = SURVEY.md 8(d)
        # GEMM term 2*NT*(2mL)*(2mR) for the images one launch processes (x10 on the two Label-on-B bonds).
        n_fg, ms_fg = prof.get("fgemm_fwd", (0, 0.0))
        n_ff, ms_ff = prof.get("fwd_fused", (0, 0.0))
        fused = n_ff > n_fg                                     # the forward pass runs as one persistent kernel (kernels_fused.hip)
        if fused:
            n_fg, ms_fg = n_ff, ms_ff
        # every timed bond calls the feature GEMM with its own (mL, mR); average the flops
        fl = [2.0 * NTl * (2 * r["mL"]) * (2 * r["mR"]) * (10 if (r["label_on_B"] and not single) else 1) for r in timed]
        flops_per_pass = float(np.mean(fl)) if fl else 0.0
        img_fg = img_ld = NTl
        flops_per_launch = flops_per_pass
        avg_ms = ms_fg / max(n_fg, 1)
        achieved_tf = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        peak = F64_MFMA_PEAK_TF if args.dtype in ("f64", "f64_e32") else (F32_MFMA_PEAK_TF if args.dtype == "f32" else BF16_MFMA_PEAK_TF)
        ms_per_step = 1e3 * elapsed / args.steps
        kms = {k: v[1] / nbreak for k, v in prof_all.items() if v[0]}
        svd_after = kms.get("svd", 0.0)
        if prof.get("svd", (0, 0.0))[0]:
            kms["svd"] = prof["svd"][1] / prof["svd"][0]           # the split of the TIMED bond updates
        # whole-step MFMA fraction.  executed: the GEMM launches a bond update really issues (5 forward + 4 gradient with the two
        # shortcuts on) + its shiftE; algorithmic: SURVEY.md 8(d), (3P+1) passes of 2 NT (2m)^2 + 2 NT 2m 10, the shift, 22 (2m)^3
        nf_step = (prof_all.get("fgemm_fwd", (0, 0))[0] + prof_all.get("fwd_fused", (0, 0))[0]) / max(nbreak, 1)
        nb_step = prof_all.get("bgemm", (0, 0))[0] / max(nbreak, 1)
        sh = float(np.mean([shift_flops(r, NTl, N, single) for r in timed])) if timed else 0.0
        exec_gf = ((nf_step + nb_step) * flops_per_pass + sh) / 1e9
        m_avg = float(np.mean([0.5 * (r["mL"] + r["mR"]) for r in timed])) if timed else 0.0
        alg_gf = ((3 * npass + 1) * (flops_per_pass + 2.0 * NTl * 2 * m_avg * (1 if single else 10)) + sh + 22.0 * (2 * m_avg) ** 3) / 1e9
        grad_classes = ("fgemm_fwd", "fwd_fused", "labeldot", "p_update", "bgemm", "slab_reduce", "zprime", "allreduce")
        tr_fg, src_fg = pmc_traffic("fwd_fused" if fused else "fgemm_fwd") if args.dtype == "f64" and maxm == 120 and NT == 60000 and world == 1 else (None, None)
        shortcuts_on = os.environ.get("TNML_FAST_CG", "1") != "0" or os.environ.get("TNML_REUSE_P", "1") != "0"
        out = {
            "metric": "two-site bond updates/sec" if not single else "two-site bond updates/sec (per-label variant, label %d)" % args.single_label,
            "value": args.steps / elapsed,
            "unit": "bond updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "parity": "unpinned-oracle (no reference-held vectors exist and ITensor is absent: GPU results are checked against "
                      "oracle/, a line-cited restatement of fixedL.cc cross-checked by an independent numpy restatement)",
            "config": {"workload": "fixedL N=%d, maxm=%d, %d images (%s), Npass=%d, lambda=%g, minm=%d, "
                                   "%s; %s; timed bonds %d..%d (%s, m=%.0f)" % (N, maxm, NT, baseline_config_name(N, maxm, NT, world, args.dtype), npass, lam, minm,
                                                                               {"f64": "fp64 throughout", "f64_e32": "fp64 MFMA over fp32-stored environments", "f32": "fp32 study mode",
                                                                                "bf16x3": "fp32 storage, forward feature GEMM on bf16 MFMA with hi + lo operands (study mode)",
                                                                                "bf16": "fp32 storage, forward feature GEMM on bf16 MFMA (study mode)"}[args.dtype],
                                                                               "random-init W at m=maxm" if args.workload == "default" else "SURVEY 8(d): after one warm-up sweep from the random-init W",
                                                                               timed[0]["bond"] if timed else 0,
                                                                               timed[-1]["bond"] if timed else 0,
                                                                               "whole sweeps" if full_sweeps else
                                                                               ("consecutive interior bonds, Label-carrying shiftE on each" if args.workload == "default" else "consecutive interior bonds of the second sweep"),
                                                                               m_avg),
                       "global_images": NT, "sites": N, "maxm": maxm, "parallelism": "dp%d (image sharding + %s)" % (world, "RCCL all-reduce" if main_tr == "rccl" or world == 1 else "one-shot all-reduce over IPC-mapped regions"),
                       "rccl_ranks": comm_ranks},
            "roofline": {"bound": "mfma", "kernel": FWD_KERNEL[0] if fused else ("k_fgemm64" if args.dtype in ("f64", "f64_e32") else ("k_fgemm" if args.dtype == "f32" else "k_fgemm_bf16")),
                         "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved_tf / peak,
                         "traffic": tr_fg, "traffic_source": src_fg,
                         "avg_launch_ms": avg_ms, "launches": n_fg, "flops_per_launch": flops_per_launch, "images_per_launch": img_fg,
                         "note": (("k_fwd_res = the whole forward pass B*t.v in one launch: the feature GEMM (these flops; the bond matrix resident in the "
                                   "registers of a pair of workgroups) with the label dot of the previous 32-image tile on four streaming waves of the "
                                   "same workgroup"
                                   if FWD_KERNEL[0] == "k_fwd_res" else
                                   "k_fwd_fused = the feature GEMM (these flops) with the label dot of the previous 64-image tile running on four extra "
                                   "waves of the same workgroup")
                                  + ": its launch time replaces feature GEMM + label dot (kernel_ms_per_step.fwd_fused); "
                                    "bytes streamed beside the flops: roofline_hbm") if fused else None},
            "roofline_step": {"bound": "mfma", "peak": peak, "unit": "TFLOP/s",
                              "achieved": exec_gf / ms_per_step, "frac": exec_gf / ms_per_step / peak,
                              "executed_gflop_per_step": exec_gf, "frac_executed": exec_gf / ms_per_step / peak,
                              "note": "whole bond update (GEMMs + label dot + CG vectors + split + shiftE) against the MFMA peak, on the flops of the GEMM "
                                      "launches really issued (algebraic shortcuts on).  For reference only: SURVEY.md 8(d)'s count for the literal "
                                      "order is algorithmic_gflop_per_step; dividing it by this step time credits flops that were not executed",
                              "algorithmic_gflop_per_step": alg_gf},
            "kernel_ms_per_step": kms,
            "gradient_phase_ms": sum(kms.get(k, 0.0) for k in grad_classes),
            "svd_ms": kms.get("svd", 0.0),
            "svd_ms_on_the_breakdown_steps": svd_after,
            "roofline_hbm": hbm_roofline(prof_all, img_ld, timed, args, world),
            "algebraic_shortcuts": [
                "fast CG: B*t.v is linear in B, so P <- P + a (p*t.v) replaces Npass-1 forward GEMMs per bond (TNML_FAST_CG=0 disables)",
                "the network outputs P_n do not depend on the bond they are evaluated at: the after-SVD quadcost of one bond update "
                "provides the residuals of the next one's first gradient, replacing 1 forward GEMM + label dot per bond (TNML_REUSE_P=0 disables)",
            ] if shortcuts_on else [],
            "unfused_forward": None if not unfused or not unfused.get("fgemm_fwd", (0, 0))[0] else {
                "note": "the same forward pass as two kernels (tnml_set_option fused_fwd = 0), timed in this run on untimed extra steps",
                "fgemm_fwd_ms_per_launch": unfused["fgemm_fwd"][1] / unfused["fgemm_fwd"][0],
                "fgemm_fwd_frac_of_mfma_peak": flops_per_pass / (unfused["fgemm_fwd"][1] / unfused["fgemm_fwd"][0] * 1e-3) / 1e12 / peak,
                "labeldot_ms_per_launch": unfused["labeldot"][1] / max(unfused["labeldot"][0], 1),
                "fused_ms_per_launch": avg_ms},
            "value_literal_order": (nlit / elapsed_lit) if elapsed_lit else None,
            "literal_order_steps": nlit,
            "literal_order_note": ("timed on the SAME bonds as `value` (weights and environments reset to the start of the main window, same warm-up): "
                                   "every forward pass of fixedL.cc:374-421 executed" if lit_same_bonds else
                                   "timed on the bonds FOLLOWING the main window (not like for like with `value`: the split is cheaper there)") if elapsed_lit else None,
            "env_init_s": t_init,
            "device_gb": ts.device_bytes() / 1e9,
            "env_host_tier": None if args.env_budget_gb <= 0 else dict(budget_gb=args.env_budget_gb, **ts.env_stats()),
            "last_cost_per_image": timed[-1]["cost"] / NT if timed else None,
            "svd_stats": ts.svd_stats(),
            "speculative_split": {
                "splits_in_timed_region": split1["speculative_splits"] - split0["speculative_splits"],
                "roll_backs_in_timed_region": split1["roll_backs"] - split0["roll_backs"],
                "roll_back_ms_in_timed_region": split1["roll_back_ms"] - split0["roll_back_ms"],
                "roll_backs_per_sweep": (split1["roll_backs"] - split0["roll_backs"]) * 2.0 * (N - 1) / args.steps,
                "ms_per_roll_back": ((split1["roll_back_ms"] - split0["roll_back_ms"]) / (split1["roll_backs"] - split0["roll_backs"])) if split1["roll_backs"] > split0["roll_backs"] else None,
                "note": "a failed deferred check of a speculative split repeats that bond update with the synchronous split and the one begun after it "
                        "(tnml_split_stats: device time of the repeated work, inside `value`'s window)"},
            "replica_repairs": ts.replica_repairs(),
            "centre_bond_ms": None if centre_ms is None else {
                "label_on_B_bond_%d" % (N // 2 - 1): centre_ms[0], "label_on_B_bond_%d" % (N // 2): centre_ms[1],
                "note": "the two bond updates whose bond tensor carries the Label index (10x the GEMM columns, split of a 240 x 2400 matrix), "
                        "timed one by one after the main window (SURVEY.md 8(d))"},
            "value_8d": None if rate_8d is None else {
                "value": rate_8d, "unit": "bond updates/s",
                "note": "SURVEY.md 8(d) workload: one warm-up sweep from the random-init W with minm = maxm/2, then 60 consecutive interior bonds "
                        "of the second sweep (trained bonds shrink towards maxm/2, so this is NOT the shape of `value`)"},
            "images_per_rank": shard_sizes,
            "collectives": None if world == 1 else {
                "allreduces_per_bond_update": (coll1[0] - coll0[0]) / args.steps, "broadcasts_per_bond_update": (coll1[1] - coll0[1]) / args.steps,
                "allreduce_ms_per_bond_update": kms.get("allreduce", 0.0),
                "ms_per_allreduce": (prof_all["allreduce"][1] / prof_all["allreduce"][0]) if prof_all.get("allreduce", (0, 0))[0] else None,
                "mode": ts.allreduce_mode() if hasattr(ts, "allreduce_mode") else "rccl",
                "note": "sum all-reduces of the packed [scalars | gradient or A p] buffer (merged CG passes, carried after-SVD scalars) and "
                        "broadcasts of rank 0's eigenvalues, per bond update of the timed region; allreduce ms from HIP events on the breakdown steps"},
        }
        if world > 1:
            # both transports side by side: this process's own (`value`) now, the other one after its child leg below
            cm = out["collectives"]
            cm["value_from"] = main_tr
            cm[main_tr] = {"value": out["value"], "ms_per_step": ms_per_step, "allreduces_per_bond_update": cm["allreduces_per_bond_update"],
                           "broadcasts_per_bond_update": cm["broadcasts_per_bond_update"], "allreduce_ms_per_bond_update": cm["allreduce_ms_per_bond_update"],
                           "ms_per_allreduce": cm["ms_per_allreduce"], "gradient_phase_ms": out["gradient_phase_ms"], "svd_ms": out["svd_ms"], "mode": cm["mode"],
                           "ranks": comm_ranks}
            ref, ref_file = n1_reference(N, maxm, NT, args.dtype) if (not full_sweeps and args.workload == "default" and not single) else (None, None)
            out["speedup_vs_n1"] = None if not ref else {
                "bond_updates": out["value"] / ref["value"], "gradient_phase": ref["gradient_phase_ms"] / out["gradient_phase_ms"] if out["gradient_phase_ms"] else None,
                "n1_value": ref["value"], "n1_gradient_phase_ms": ref["gradient_phase_ms"], "n1_svd_ms": ref.get("svd_ms"), "n1_source": ref_file,
                "note": "against the committed one-GPU line of the same workload and window (same command at --gpus 1 on a box of the same pool, "
                        "not re-measured in this run); the driver computes scaling efficiency from its own per-N runs"}
        # The three matrix-pipe kernels of a bond update, each from its own HIP events over the TIMED region, with the stamped PMC
        # traffic where it is current; `roofline` is the one with the most time per bond update (the block above describes the forward
        # kernel when that is it).
        if args.dtype in ("f64", "f64_e32"):
            pmc_ok = args.dtype == "f64" and maxm == 120 and NT == 60000 and world == 1
            rk = {}
            for cls, kname, per_step_flops, alg_bytes in (
                    ("fwd_fused" if fused else "fgemm_fwd", out["roofline"]["kernel"], flops_per_pass, float(np.mean([NTl * (11 * min(r["mL"], r["mR"]) * 8 + 2 * 2 * 8 + 4) for r in timed])) if timed else None),
                    ("bgemm", GRAD_KERNEL[0], flops_per_pass, float(np.mean([NTl * (11 * min(r["mL"], r["mR"]) * 8 + 2 * 2 * 8 + 10 * 8) for r in timed])) if timed else None),
                    ("fgemm_shift", "k_shift_res / k_fgemm64 (shiftE)", sh,
                     float(np.mean([NTl * 8.0 * ((r["mL"] if r["half"] == 1 else r["mR"]) + r["newm"]) * (shift_flops(r, NTl, N, single) / (2.0 * NTl * 2 * (r["mL"] if r["half"] == 1 else r["mR"]) * r["newm"])) + NTl * 16.0
                                    for r in timed])) if timed else None)):
                nl_, ms_ = prof.get(cls, (0, 0.0))
                if not nl_:
                    continue
                per_launch_ms = ms_ / nl_
                launches_per_step = nl_ / args.steps
                fl_launch = per_step_flops if cls != "fgemm_shift" else sh / max(launches_per_step, 1e-9)
                tr_, src_ = pmc_traffic(cls) if pmc_ok else (None, None)
                ach = fl_launch / (per_launch_ms * 1e-3) / 1e12
                # both roofs of the same launch: below the ridge (peak flops / peak bytes per second) the HBM stream is the binding one --
                # the trained bonds of the 8(d) workload (m = 59-60: 5.5 flop/B against a ridge of 9.8) -- at m = 120 (10.9 flop/B) the matrix pipe
                hbm_gbs = (alg_bytes / (per_launch_ms * 1e-3) / 1e9) if alg_bytes else None
                ai = (fl_launch / alg_bytes) if alg_bytes else None
                ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
                rk[cls] = {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                           "avg_launch_ms": per_launch_ms, "launches": nl_, "ms_per_step": ms_ / args.steps, "flops_per_launch": fl_launch,
                           "algorithmic_bytes_per_launch": alg_bytes, "traffic": tr_,
                           "traffic_over_algorithmic": (tr_ / alg_bytes) if (tr_ and alg_bytes) else None, "traffic_source": src_,
                           "hbm_achieved_gbs": hbm_gbs, "hbm_frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None,
                           "flop_per_byte": ai, "ridge_flop_per_byte": ridge,
                           "binding_roof": None if ai is None else ("hbm" if ai < ridge else "mfma"),
                           "frac_of_binding_roof": None if ai is None else ((hbm_gbs / HBM_PEAK_GBS) if ai < ridge else ach / peak)}
            out["roofline_kernels"] = rk
            if rk:
                dom = max(rk, key=lambda k: rk[k]["ms_per_step"])
                if dom != ("fwd_fused" if fused else "fgemm_fwd"):
                    d = rk[dom]
                    out["roofline"] = {"bound": "mfma", "kernel": d["kernel"], "achieved": d["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": d["frac"],
                                       "traffic": d["traffic"], "traffic_source": d["traffic_source"], "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
                                       "flops_per_launch": d["flops_per_launch"], "images_per_launch": NTl,
                                       "note": "the kernel with the most time per bond update (%.3f ms; all three matrix-pipe kernels: roofline_kernels)" % d["ms_per_step"]}
                out["roofline"]["dominant_by_ms_per_step"] = True
        st = pmc_step_traffic() if args.dtype == "f64" and maxm == 120 and NT == 60000 and world == 1 and not full_sweeps else None
        if st:
            out["roofline_step"]["traffic"] = st["bytes_per_step"]
            out["roofline_step"]["traffic_gbs"] = st["bytes_per_step"] / (ms_per_step * 1e-3) / 1e9
            out["roofline_step"]["traffic_frac_of_hbm_peak"] = st["bytes_per_step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline_step"]["traffic_source"] = st.get("source")
        if world == 1 and not args.no_cpu_baseline and not single:
            ncore = cpu_quota()
            out["cpu_baseline"] = cpu_baseline(maxm, npass, lam, cutoff, min(16, ncore), NT, full=args.cpu_baseline_full)   # paralleldo.h:55-56 caps at 16
    ts.close()
    if world > 1 and not args.leg:
        # ---- the SAME window over the other transport, in this invocation: one child process per rank (own rendezvous), after this
        # process has released its device memory.  Whatever happens there ends up as a block or an error string under `collectives`.
        other = "oneshot" if main_tr == "rccl" else "rccl"
        block = None
        if args.no_second_transport:
            block = {"skipped": "--no-second-transport"}
        elif args.share_device and other == "rccl":
            block = {"error": "RCCL refuses two ranks on one device (--share-device is the one-GPU test vehicle): not run"}
        else:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            port = [free_port() if rank == 0 else None]
            dist.broadcast_object_list(port, src=0)
            line, err = second_transport_leg(other, port[0], rank, local_rank, world, int(os.environ.get("TNML_BENCH_LEG_TIMEOUT", "900")))
            errs = [None] * world
            dist.all_gather_object(errs, err)
            if rank == 0:
                bad = [e for e in errs if e]
                if bad or not line:
                    block = {"error": bad[0] if bad else "the %s leg printed no line" % other}
                else:
                    d2 = json.loads(line)
                    c2 = d2.get("collectives") or {}
                    block = {"value": d2["value"], "ms_per_step": d2["ms_per_step"], "allreduces_per_bond_update": c2.get("allreduces_per_bond_update"),
                             "broadcasts_per_bond_update": c2.get("broadcasts_per_bond_update"), "allreduce_ms_per_bond_update": c2.get("allreduce_ms_per_bond_update"),
                             "ms_per_allreduce": c2.get("ms_per_allreduce"), "gradient_phase_ms": d2.get("gradient_phase_ms"), "svd_ms": d2.get("svd_ms"),
                             "mode": c2.get("mode"), "ranks": (d2.get("config") or {}).get("rccl_ranks"),
                             "note": "the same command and window in a child process per rank, started after the main measurement had released the devices"}
        if rank == 0:
            out["collectives"][other] = block
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
