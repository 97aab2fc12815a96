"""Randomised differential run of the HIP path against the oracle (a debugging aid, not part of the suites; like everything that
touches oracle/ it lives under tests/):  python tests/debug/fuzz_lockstep.py [ncases] [seed]
Each case draws a chain (4..14 sites, bond dimensions with unequal / odd sizes up to ~40, occasionally up to 160), an image count, a
feature scale, sweep parameters (maxm, minm, cutoff, Npass, lambda) and an arithmetic, runs one sweep of bond updates in lockstep with
the oracle and reports the worst deviations; any mismatch of a kept bond dimension or a cost beyond tolerance is printed with the case.
The oracle runs twice, with 1 and with 3 threads (a different summation order): on badly conditioned bonds -- the reference's own feature
map with a small lambda has CG step sizes of 1e5 and more -- the two oracle runs differ from each other by as much as the HIP path differs
from either (tests/debug/fuzz_case38.py), so a deviation only counts when it exceeds 100x the oracle's own sensitivity."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tnml_amd import lib  # noqa: E402
from tnml_amd.fixedl import TrainStates  # noqa: E402
from test_gpu_parity import _mps_with_dims  # noqa: E402


def one_case(rng, idx):
    huge = bool(os.environ.get("FUZZ_HUGE"))                        # FUZZ_HUGE=1: short chains with bond dimensions up to 320 (the split's size classes)
    N = int(rng.integers(4, 7)) if huge else int(rng.integers(4, 15))
    big = rng.random() < 0.15
    cap = 320 if huge else (160 if big else 40)
    dims = [1]
    for j in range(1, N):
        lim = min(cap, 2 * dims[-1], 2 ** min(N - j, 12)) if not huge else cap
        dims.append(int(rng.integers(100, lim + 1)) if huge else int(rng.integers(1, max(2, min(lim, 2 * dims[-1]) + 1))))
    dims.append(1)
    for j in range(N - 1, 0, -1):
        if not huge:
            dims[j] = min(dims[j], 2 * dims[j + 1])
    NT = int(rng.choice([7, 33])) if huge else int(rng.choice([1, 7, 33, 64, 100, 257]))
    boost = float(rng.choice([1.0, 30.0, 200.0]))
    dtype = str(rng.choice(["f64", "f64", "f64", "f64_e32", "f32"]))
    maxm = int(rng.integers(2, max(3, max(dims) + 3)))
    minm = int(rng.integers(1, maxm + 1))
    cutoff = float(rng.choice([0.0, 1e-12, 1e-8, 1e-4]))
    npass = int(rng.integers(1, 5))
    # lambda >= 1e-3: with lambda = 0 or 1e-6 and fewer images than unknowns the normal equations are (nearly) singular, the CG's step
    # sizes reach 1e5 and more and its later passes are rounding noise in ANY implementation (the oracle run with 3 threads differs from
    # the oracle run with 1 thread by orders of magnitude in cost) -- nothing to compare there
    lam = float(rng.choice([1e-3, 1e-2, 1e-1]))
    if os.environ.get("FUZZ_STRICT"):                                 # FUZZ_STRICT=1: the well-conditioned regime only, where fp64 must follow the oracle to 1e-7
        dtype, boost, lam = "f64", float(rng.choice([30.0, 200.0])), float(rng.choice([1e-2, 1e-1]))
        NT = max(NT, 33)
    tol = {"f64": 1e-7, "f64_e32": 2e-3, "f32": 5e-2}[dtype]
    desc = dict(case=idx, N=N, dims=dims, NT=NT, boost=boost, dtype=dtype, maxm=maxm, minm=minm, cutoff=cutoff, npass=npass, lam=lam)
    only = os.environ.get("FUZZ_ONLY")                                # FUZZ_ONLY=i,j,...: draw every case (the generator advances) but run only these
    if only and idx not in [int(x) for x in only.split(",")]:
        return desc, 0.0, []
    pixels, labels, phi, _ = make_problem(N, NT, 2, 5 + idx, pixel_boost=boost)
    W = _mps_with_dims(dims, 100 + idx)
    ts = TrainStates(labels, N, max(max(dims), maxm), phi=phi, dtype=dtype)
    o = pyoracle.Oracle(phi, labels, W)
    o3 = pyoracle.Oracle(phi, labels, W, nthread=3)
    ts.set_mps(W); o.init(); o3.init(); ts.init()
    tl = None
    if str(idx) in os.environ.get("FUZZ_TRACE", "").split(","):             # a second HIP context in the literal evaluation order (no P recurrence, no carried outputs)
        tl = TrainStates(labels, N, max(max(dims), maxm), phi=phi, dtype=dtype)
        tl.set_option("fast_cg", 0); tl.set_option("reuse_p", 0)
        tl.set_mps(W); tl.init()
    b, ha, worst, bad = 1, 1, 0.0, []
    while ha <= 2:
        r = ts.bond_update(b, ha, maxm, minm, cutoff, npass, lam, 1e-10)
        o.set_bond(b)
        B, tr = o.cgrad(o.bond_tensor(b), npass, lam, 1e-10)
        trace = str(idx) in os.environ.get("FUZZ_TRACE", "").split(",")
        newm, te, sv = o.svd_split(B, b, ha, cutoff, maxm, minm)
        C, lc, cr, nc = o.quadcost(o.bond_tensor(b), lam)
        o.shiftE(b, ha == 1)
        o3.set_bond(b)
        B3, tr3 = o3.cgrad(o3.bond_tensor(b), npass, lam, 1e-10)
        if trace:                                                       # FUZZ_TRACE=<case>: the CG step sizes of the three runs, bond by bond
            rl = tl.bond_update(b, ha, maxm, minm, cutoff, npass, lam, 1e-10)
            tl.set_site(b, o.get_site(b)); tl.set_site(b + 1, o.get_site(b + 1)); tl.shiftE(b, ha == 1)
            f = lambda v: ["%.6g" % x for x in v]
            print("   bond %d half %d: alpha  oracle(1 thread) %s | oracle(3 threads) %s | HIP %s | HIP literal order %s" % (b, ha, f(tr["alpha"]), f(tr3["alpha"]), f(r["cg"]["alpha"]), f(rl["cg"]["alpha"])))
            print("                  |r|    oracle(1 thread) %s | oracle(3 threads) %s | HIP %s | HIP literal order %s" % (f(tr["rnorm"]), f(tr3["rnorm"]), f(r["cg"]["rnorm"]), f(rl["cg"]["rnorm"])))
            print("                  cost   oracle(1 thread) %s | HIP %s | HIP literal order %s" % (f(tr["cost"]), f(r["cg"]["cost"]), f(rl["cg"]["cost"])))
        o3.svd_split(B3, b, ha, cutoff, maxm, minm)
        C3 = o3.quadcost(o3.bond_tensor(b), lam)[0]
        o3.set_site(b, o.get_site(b)); o3.set_site(b + 1, o.get_site(b + 1)); o3.shiftE(b, ha == 1)
        own = abs(C3 - C) / max(abs(C), 1e-300)                        # the oracle's sensitivity to its own summation order
        rel = abs(r["cost"] - C) / max(abs(C), 1e-300)
        if abs(C) < 1e-20 * max(1.0, NT):                               # a perfect fit: relative errors of a zero cost mean nothing
            rel = 0.0
        worst = max(worst, rel)
        if r["newm"] != newm:
            # a kept dimension may legitimately differ where the spectrum sits on the cutoff within round-off: report it with the margin
            p = np.sort(sv ** 2)[::-1]
            tail = np.cumsum(p[::-1])[::-1] / max(p.sum(), 1e-300)
            bad.append(("newm", b, ha, r["newm"], newm, [float(x) for x in tail[max(0, min(newm, r["newm"]) - 1):max(newm, r["newm"]) + 1]]))
        elif rel > tol + 100 * own:
            bad.append(("cost", b, ha, r["cost"], C))
        if dtype == "f64" and r["newm"] == newm and r["ncorrect"] != nc:
            bad.append(("ncorrect", b, ha, r["ncorrect"], nc))
        ts.set_site(b, o.get_site(b)); ts.set_site(b + 1, o.get_site(b + 1))
        ts.shiftE(b, ha == 1)
        b, ha = lib.sweepnext(b, ha, N)
    ts.close()
    return desc, worst, bad


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    nbad = 0
    for idx in range(ncases):
        try:
            desc, worst, bad = one_case(rng, idx)
        except Exception as e:                                         # noqa: BLE001
            print("case %d raised %s: %s" % (idx, type(e).__name__, str(e)[:300]))
            nbad += 1
            continue
        flag = "OK " if not bad else "BAD"
        print("%s case %2d %-7s N=%2d NT=%3d maxm=%3d minm=%3d cutoff=%g npass=%d lam=%g boost=%g dims=%s worst rel. cost error %.1e" %
              (flag, idx, desc["dtype"], desc["N"], desc["NT"], desc["maxm"], desc["minm"], desc["cutoff"], desc["npass"], desc["lam"], desc["boost"], desc["dims"], worst))
        for x in bad[:4]:
            print("     ", x)
        nbad += bool(bad)
    print("%d of %d cases with findings" % (nbad, ncases))


if __name__ == "__main__":
    main()
