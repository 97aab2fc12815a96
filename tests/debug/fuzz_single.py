"""Randomised differential run of the per-label variant (TNML_MODE_SINGLE) against its oracle: random chains with unequal bond dimensions,
image counts, target labels, optimiser (conj / fast_conj), noise (0 / 1e-6 / 1e-3), sweep parameters; one sweep each through mldmrg on
both sides, in lockstep (every bond update starts from the oracle's network).   python tests/debug/fuzz_single.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
from tnml_amd import synth  # noqa: E402
from tnml_amd.fixedl import TrainStates, mldmrg  # noqa: E402


def one_case(rng, idx):
    N = int(rng.integers(4, 13))
    dims = [1]
    for j in range(1, N):
        dims.append(int(rng.integers(1, min(24, 2 * dims[-1]) + 1)))
    dims.append(1)
    for j in range(N - 1, 0, -1):
        dims[j] = min(dims[j], 2 * dims[j + 1])
    NT = int(rng.choice([10, 30, 70, 130]))
    target = int(rng.integers(0, 10))
    normal = bool(rng.integers(0, 2))
    boost = float(rng.choice([30.0, 300.0]))
    method = str(rng.choice(["conj", "conj", "fast_conj"]))
    noise = float(rng.choice([0.0, 0.0, 1e-6, 1e-3]))
    maxm = int(rng.integers(2, max(dims) + 3)); minm = int(rng.integers(1, maxm + 1))
    cutoff = float(rng.choice([0.0, 1e-12, 1e-8])); npass = int(rng.integers(1, 4)); lam = float(rng.choice([1e-3, 1e-2, 1e-1]))
    labels = synth.synthetic_labels(NT, seed=idx, per_label=NT // 10)
    pixels = synth.synthetic_images(N, labels, seed=idx)
    phi = pyoracle.features_single(pixels, normal).copy()
    phi[..., 1] *= boost
    g = np.random.default_rng(1000 + idx)
    W = []
    for j in range(1, N + 1):
        A = g.standard_normal((dims[j - 1], 2, dims[j])) / np.sqrt(2. * max(dims[j - 1], dims[j]))
        A[:, 0] += np.eye(dims[j - 1], dims[j])
        W.append(A)
    ts = TrainStates(labels, N, max(max(dims), maxm), phi=phi, single_label=target)
    o = pyoracle.SingleOracle(phi, labels, target, W)
    ts.set_mps(W); o.init(); ts.init()
    if method == "fast_conj":
        ts.set_option("cg_method", 1); o.set_method("fast_conj")
    if noise:
        ts.set_option_real("noise", noise); o.set_noise(noise)
    from tnml_amd import lib
    bad, worst = [], 0.0
    b, ha = 1, 1
    while ha <= 2:                                                     # in lockstep: every bond update starts from the oracle's network
        r = ts.bond_update(b, ha, maxm, minm, cutoff, npass, lam, 1e-10, report_costs=True)
        o.set_bond(b)
        B0 = o.bond_tensor(b)
        Bo, _ = (o.fast_cgrad if method == "fast_conj" else o.cgrad)(B0, npass, lam, 1e-10)
        if noise:
            newm, te = o.noise_split(Bo, b, ha, noise, cutoff, maxm, minm)
        else:
            newm, te, _ = o.svd_split(Bo, b, ha, cutoff, maxm, minm)
        C = o.quadcost(o.bond_tensor(b), lam)[0]
        o.shiftE(b, ha == 1)
        rel = abs(r["cost"] - C) / max(abs(C), 1e-300)
        worst = max(worst, rel)
        if r["newm"] != newm:
            bad.append(("newm", b, ha, r["newm"], newm))
        elif rel > (1e-7 if method == "conj" else 1e-4):
            bad.append(("cost", b, ha, r["cost"], C))
        ts.set_site(b, o.get_site(b)); ts.set_site(b + 1, o.get_site(b + 1)); ts.shiftE(b, ha == 1)
        b, ha = lib.sweepnext(b, ha, N)
    ts.close()
    desc = "N=%2d NT=%3d target=%d %-9s noise=%g maxm=%2d minm=%2d cutoff=%g npass=%d lam=%g boost=%g %s dims=%s" % (
        N, NT, target, method, noise, maxm, minm, cutoff, npass, lam, boost, "normal" if normal else "series", dims)
    return desc, worst, bad


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    nbad = 0
    for idx in range(ncases):
        try:
            desc, worst, bad = one_case(rng, idx)
        except Exception as e:                                         # noqa: BLE001
            print("case %d raised %s: %s" % (idx, type(e).__name__, str(e)[:300])); nbad += 1; continue
        print("%s case %2d %s worst rel. cost error %.1e" % ("OK " if not bad else "BAD", idx, desc, worst))
        for x in bad:
            print("     ", x)
        nbad += bool(bad)
    print("%d of %d cases with findings" % (nbad, ncases))


if __name__ == "__main__":
    main()
