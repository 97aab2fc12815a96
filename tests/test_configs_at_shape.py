"""The BASELINE configurations at their STATED shapes, and the kernel instantiations bench.py times, against the oracle.

C1  fixedL N=196 (14x14), 100 images per label, maxm=10, Nsweep=2      -- every one of the 780 bond updates
C2  fixedL N=784, maxm=20, 1000 images per label                        -- properties + bond updates vs the oracle
C3  the tile configurations chosen at 60 000 images (feature GEMM 128 x 240 / 12 waves, streaming label dot) run at a
    size the oracle handles, for all three bond kinds
accuracy: a 10 000-image test set, so that "test accuracy within 0.1 %" (north_star) is resolvable.

Why C1 is checked in LOCKSTEP.  A free-running sweep is a chaotic map of its rounding errors: the oracle run with 1 and
with 8 threads (only the summation order of paralleldo.h:51-67 differs) agrees on the per-bond cost to 1e-15 for 10
bonds, 1e-8 after 50, 1e-4 after 250 and 17 % by the end of the first sweep, although both reach the same final cost to
~0.4 % (tests/golden/make_golden_c1.py prints the spread).  No tolerance on a free-running trajectory can therefore
tell a correct implementation from a wrong one after the first ~100 bonds.  The lockstep test removes the accumulation:
after every bond update the GPU's two site tensors are overwritten with the oracle's and the environment is rebuilt
from them, so each of the 780 bond updates starts from the same state on both sides and is compared on its own."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD_C1 = os.path.join(ROOT, "tests", "golden", "fixedl_c1.npz")


def _c1():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_c1 as mg
    g = np.load(GOLD_C1)
    W = [g["W%03d" % j] for j in range(1, int(g["N"]) + 1)]
    return g, mg.features(g["pixels"]), W, dict(mg.PARAMS)


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def test_c1_oracle_reproduces_the_golden_prefix():
    """the oracle with the generator's thread count (chunking is deterministic in nthread, not in the core count)"""
    from oracle import pyoracle
    g, phi, W, p = _c1()
    o = pyoracle.Oracle(phi, g["labels"], W, nthread=8)
    o.init()
    reps = o.mldmrg(p["nsweep"], p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"], max_bonds=60)
    np.testing.assert_allclose([r["cost"] for r in reps], g["cost"][:60], rtol=1e-9)
    assert [r["newm"] for r in reps] == list(g["newm"][:60])
    assert [r["ncorrect"] for r in reps] == list(g["ncorrect"][:60])
    np.testing.assert_allclose(reps[0]["cg"]["cost"][:p["npass"] - 1], g["cg_cost0"], rtol=1e-12)


@pytest.mark.gpu
def test_c1_every_bond_update_in_lockstep_with_the_oracle():
    """BASELINE config 1 at full size: 2 sweeps x 390 bond updates, each compared on its own (cost, per-label costs,
    new bond dimension, truncation error, #correct), then the prediction VECTOR of the trained network
    (fixedL.cc:321-326, util.h:42-57) from tnml_classify against the oracle's toverlap."""
    from oracle import pyoracle
    from tnml_amd import lib
    from tnml_amd.fixedl import TrainStates
    g, phi, W, p = _c1()
    N, NT = int(g["N"]), int(g["NT"])
    ts = TrainStates(g["labels"], N, p["maxm"], phi=phi)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, g["labels"], W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    worst = dict(cost=0.0, lc=0.0, te=0.0)
    nbonds = 0
    ncorr_mismatch = 0
    for sw in range(p["nsweep"]):
        b, ha = 1, 1
        while ha <= 2:
            r = ts.bond_update(b, ha, p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"])
            o.set_bond(b)                                                   # fixedL.cc:488-540 on the oracle
            B, tr = o.cgrad(o.bond_tensor(b), p["npass"], p["lam"], p["cconv"])
            newm, te, _ = o.svd_split(B, b, ha, p["cutoff"], p["maxm"], p["minm"])
            C, lc, cr, nc = o.quadcost(o.bond_tensor(b), p["lam"])
            o.shiftE(b, ha == 1)
            assert r["newm"] == newm, (sw, b, ha)
            worst["cost"] = max(worst["cost"], abs(r["cost"] / C - 1))
            worst["lc"] = max(worst["lc"], np.abs(r["label_cost"] - lc).max() / C)
            worst["te"] = max(worst["te"], abs(r["truncerr"] - te))
            ncorr_mismatch += int(r["ncorrect"] != nc)
            np.testing.assert_allclose(r["cg"]["cost"], tr["cost"], rtol=1e-7, err_msg="CG cost trace, sweep %d bond %d" % (sw, b))
            # same state for the next bond update on both sides
            ts.set_site(b, o.get_site(b))
            ts.set_site(b + 1, o.get_site(b + 1))
            ts.shiftE(b, ha == 1)
            nbonds += 1
            b, ha = lib.sweepnext(b, ha, N)
    print("C1 lockstep: %d bond updates; worst rel. cost error %.2e, per-label cost %.2e (of C), trunc. err. %.2e, "
          "#correct mismatches %d" % (nbonds, worst["cost"], worst["lc"], worst["te"], ncorr_mismatch))
    assert nbonds == 2 * 2 * (N - 1)
    assert worst["cost"] < 1e-7 and worst["lc"] < 1e-7 and worst["te"] < 1e-8
    assert ncorr_mismatch == 0
    # inference on the trained network: both sides now hold the same W
    w, pred, cnt, ninc = ts.classify()
    wo = np.stack([o.toverlap(i) for i in range(NT)])
    assert _rel(w, wo) < 1e-10
    po = np.abs(wo).argmax(axis=1)
    assert (pred == po).all()                                              # the prediction vector, not only its count
    assert int(ninc.sum()) == int((po != g["labels"]).sum())
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64_e32", "f32"])
def test_c1_lockstep_in_the_reduced_precision_modes(dtype):
    """The tolerance study of BASELINE config 5 on the HIP kernels themselves (not an emulation), with the accumulation of a
    free-running sweep taken out: every bond update of C1 starts from the oracle's state and is compared on its own.
      f64_e32  fp64 MFMA over fp32-STORED environments / features       -> gated: per-bond cost within 1e-5
      f32      v_mfma_f32 with fp32 storage (the 2x-rate matrix pipe)    -> interior bonds gated at 1e-3, the Label-on-B
               bonds (where the CG resolves the small Hessian directions only in fp64) reported
    The per-bond-kind maxima are printed; they are the measured basis of the per-bond precision policy in DESIGN.md."""
    from oracle import pyoracle
    from tnml_amd import lib
    from tnml_amd.fixedl import TrainStates
    g, phi, W, p = _c1()
    N = int(g["N"])
    ts = TrainStates(g["labels"], N, p["maxm"], phi=phi, dtype=dtype)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, g["labels"], W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    worst = {"interior": 0.0, "label_on_B": 0.0}
    newm_diff = ncorr_diff = nb = 0
    for sw in range(1):                                                        # one sweep is 390 bond updates of every kind
        b, ha = 1, 1
        while ha <= 2:
            r = ts.bond_update(b, ha, p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"])
            o.set_bond(b)
            B, tr = o.cgrad(o.bond_tensor(b), p["npass"], p["lam"], p["cconv"])
            newm, te, _ = o.svd_split(B, b, ha, p["cutoff"], p["maxm"], p["minm"])
            C, lc, cr, nc = o.quadcost(o.bond_tensor(b), p["lam"])
            o.shiftE(b, ha == 1)
            kind = "label_on_B" if r["label_on_B"] else "interior"
            worst[kind] = max(worst[kind], abs(r["cost"] / C - 1))
            newm_diff += int(r["newm"] != newm)
            ncorr_diff += abs(int(r["ncorrect"]) - int(nc))
            ts.set_site(b, o.get_site(b))
            ts.set_site(b + 1, o.get_site(b + 1))
            ts.shiftE(b, ha == 1)
            nb += 1
            b, ha = lib.sweepnext(b, ha, N)
    print("C1 lockstep in %s: %d bond updates; worst rel. cost error interior %.2e, Label on B %.2e; bond-dimension mismatches %d, "
          "#correct differences (sum) %d" % (dtype, nb, worst["interior"], worst["label_on_B"], newm_diff, ncorr_diff))
    if dtype == "f64_e32":
        assert max(worst.values()) < 1e-5 and newm_diff == 0 and ncorr_diff <= 2
    else:
        assert worst["interior"] < 1e-3
    ts.close()


@pytest.mark.gpu
def test_c1_free_running_against_the_golden_vectors():
    """the same two sweeps free-running against tests/golden/fixedl_c1.npz: tight where the trajectory is still
    determined (first 50 bonds), within the oracle's own summation-order spread afterwards"""
    from tnml_amd.fixedl import TrainStates, mldmrg
    g, phi, W, p = _c1()
    NT = int(g["NT"])
    ts = TrainStates(g["labels"], int(g["N"]), p["maxm"], phi=phi)
    ts.set_mps(W)
    ts.init()
    reps = mldmrg(ts, p["nsweep"], p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"])
    cost = np.array([r["cost"] for r in reps])
    assert len(cost) == len(g["cost"])
    np.testing.assert_allclose(reps[0]["cg"]["cost"], g["cg_cost0"], rtol=1e-9)
    np.testing.assert_allclose(cost[:50], g["cost"][:50], rtol=1e-6)
    assert [r["newm"] for r in reps[:100]] == list(g["newm"][:100])
    assert [r["ncorrect"] for r in reps[:50]] == list(g["ncorrect"][:50])
    dev = np.abs(cost / g["cost"] - 1)
    print("C1 free run vs golden: max rel. dev. bonds 0-49 %.1e, 50-249 %.1e, all %.1e; final cost/NT %.6f vs %.6f"
          % (dev[:50].max(), dev[50:250].max(), dev.max(), cost[-1] / NT, g["cost"][-1] / NT))
    assert dev[50:250].max() < 1e-2                       # oracle 1 thread vs 8 threads: 1.2e-4 here
    assert abs(cost[-1] / g["cost"][-1] - 1) < 0.05       # oracle 1 thread vs 8 threads: 0.4 %
    w, pred, cnt, ninc = ts.classify()
    assert (pred == g["pred"]).mean() >= 0.995            # decision margins of the golden network are >= 0.67
    ts.close()


@pytest.mark.gpu
def test_c3_kernel_instantiations_of_the_benchmark_match_the_oracle():
    """bench.py's 60 000-image run selects k_fgemm64<2,5,4,3,16,...> (128 x 240 tiles, 12 waves) and the streaming
    k_labeldot<4,2,10>; at oracle-sized image counts the library would pick smaller tiles, so they are forced here
    ("fg64_cfg" = 2, "ldot_cfg" = 1) and compared with the oracle for all three bond kinds at m = 120: forward map,
    gradient, cost, the CG and one whole bond update."""
    from oracle import pyoracle
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N, NT, m = 20, 300, 120
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_option("fg64_cfg", 2)
    ts.set_option("ldot_cfg", 1)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    rng = np.random.default_rng(1)
    at = 1
    for b, kind in ((8, "Label on RE"), (9, "Label on B"), (12, "Label on LE")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.05 * rng.standard_normal(B.shape)
        assert B.shape[:4] == (120, 2, 2, 120)
        assert _rel(ts.forward(B), o.forward(B)) < 1e-11, kind
        assert _rel(ts.gradient(B), o.gradient(B)) < 1e-9, kind
        Cg, lg, _, ng = ts.quadcost(B, 1e-3)
        Co, lo, _, no = o.quadcost(B, 1e-3)
        assert Cg == pytest.approx(Co, rel=1e-11) and ng == no, kind
        Bg, tg = ts.cgrad(B, 3, 1e-3, 1e-10)
        Bo, to = o.cgrad(B, 3, 1e-3, 1e-10)
        np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-9, err_msg=kind)
        np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=1e-5, err_msg=kind)
        assert _rel(Bg, Bo) < 1e-5, kind
    # one whole bond update (fast CG + carried outputs on the GPU side) on the last bond
    r = ts.bond_update(12, 1, m, m // 2, 1e-10, 3, 1e-3, 1e-10)
    o.set_bond(12)
    B, _ = o.cgrad(o.bond_tensor(12), 3, 1e-3, 1e-10)
    newm, te, _ = o.svd_split(B, 12, 1, 1e-10, m, m // 2)
    C, lc, cr, nc = o.quadcost(o.bond_tensor(12), 1e-3)
    assert r["newm"] == newm and r["ncorrect"] == nc
    assert r["cost"] == pytest.approx(C, rel=1e-8)
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("NT,kernel", [(300, "fwd_fused"), (1100, "fwd_fused"), (300, "fwd_res"), (1100, "fwd_res"), (2100, "fwd_res")])
def test_fused_forward_kernel_matches_the_oracle(NT, kernel):
    """The two one-launch forms of B*t.v that the BASELINE config 3 shape runs, forced at oracle-sized image counts.
    k_fwd_fused (feature GEMM + label dot of the previous tile in one persistent workgroup; option "fused_fwd"): 300 images = 8
    tiles on 8 workgroups (one tile + the drain round each), 1100 images = 20 tiles on 20 workgroups; with a grid cap of 4
    workgroups (fused_fwd = 4) every workgroup runs several rounds.  k_fwd_res (the bond matrix resident in the registers of a
    pair of workgroups, kernels_res.hip; option "fwd_res") + k_pfinish: 300 images = 16 tiles of 32 on 8 pairs (two rounds + fill
    and drain), 1100 images = 40 tiles on 16 pairs (ragged: 3 rounds on half of the pairs, 2 on the others), 2100 images = 72 tiles
    on 8 pairs (res_grid caps the grid: 9 rounds each).  Forward map, gradient, cost / #correct, the CG (its pAp passes use the |P|^2 mode) and a bond update, for
    both bond kinds the kernels serve."""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N, m = 20, 120
    pixels, labels, phi, W = make_problem(N, NT, m, 7, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    if kernel == "fwd_fused":
        ts.set_option("fwd_res", 0)
        ts.set_option("fused_fwd", 4 if NT > 1000 else 2)
    else:
        ts.set_option("fwd_res", 2)
        ts.set_option("shift_res", 2)                      # k_shift_res builds the Label-carrying environments of init / shiftE (checked below)
        ts.set_option("res_grid", 32 if NT == 1100 else 16)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    for j in (5, 8, 9, 10):                                   # right environments carrying the Label index, built by init (m = 120 from site 8 down)
        assert _rel(ts.env(j), o.env(j)) < 1e-12, j
    rng = np.random.default_rng(1)
    at = 1
    for b, kind in ((8, "Label on RE"), (12, "Label on LE")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
            if bb >= 10:
                assert _rel(ts.env(bb), o.env(bb)) < 1e-12, bb        # left environments carrying the Label index, built by shiftE
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.05 * rng.standard_normal(B.shape)
        assert _rel(ts.forward(B), o.forward(B)) < 1e-11, kind
        assert _rel(ts.gradient(B), o.gradient(B)) < 1e-9, kind
        Cg, lg, _, ng = ts.quadcost(B, 1e-3)
        Co, lo, _, no = o.quadcost(B, 1e-3)
        assert Cg == pytest.approx(Co, rel=1e-11) and ng == no, kind
        np.testing.assert_allclose(lg, lo, rtol=1e-9, atol=1e-12 * Co)
        Bg, tg = ts.cgrad(B, 3, 1e-3, 1e-10)
        Bo, to = o.cgrad(B, 3, 1e-3, 1e-10)
        np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-9, err_msg=kind)
        np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=1e-5, err_msg=kind)
        assert _rel(Bg, Bo) < 1e-5, kind
    prof0 = None
    ts.profile(True)
    r = ts.bond_update(12, 1, m, m // 2, 1e-10, 3, 1e-3, 1e-10)
    ts.profile(False)
    assert ts.profile_read()[kernel][0] > 0                           # the kernel under test really ran
    o.set_bond(12)
    B, _ = o.cgrad(o.bond_tensor(12), 3, 1e-3, 1e-10)
    newm, te, _ = o.svd_split(B, 12, 1, 1e-10, m, m // 2)
    C, lc, cr, nc = o.quadcost(o.bond_tensor(12), 1e-3)
    assert r["newm"] == newm and r["ncorrect"] == nc and r["cost"] == pytest.approx(C, rel=1e-8)
    ts.close()


@pytest.mark.gpu
def test_the_slab_cut_of_the_gradient_gemm_changes_only_the_summation_order():
    """options bgemm_wgs / bgemm_per (how many image slabs the gradient GEMM is cut into, profiles/r05_sweep_gradient_gemm_slabs.txt):
    one slab, the default, one slab per 32 images and a ragged cut give the same gradient to rounding, and repeat bit for bit"""
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N, m, NT = 20, 120, 2100
    pixels, labels, phi, W = make_problem(N, NT, m, 7, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_mps(W)
    ts.init()
    for bb in range(1, 8):
        ts.shiftE(bb, True)
    ts.setBond(8)
    rng = np.random.default_rng(2)
    B = ts.bond_tensor(8)
    B = B + 0.05 * rng.standard_normal(B.shape)
    G0 = ts.gradient(B)
    assert np.array_equal(G0, ts.gradient(B))
    for name, val in (("bgemm_wgs", 4), ("bgemm_wgs", 4096), ("bgemm_per", 5), ("bgemm_per", 72)):
        ts.set_option("bgemm_wgs", 0); ts.set_option("bgemm_per", 0)
        ts.set_option(name, val)
        G = ts.gradient(B)
        assert _rel(G, G0) < 1e-13, (name, val)
        assert np.array_equal(G, ts.gradient(B)), (name, val)
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("NT,wgs", [(300, 0), (1100, 12), (2100, 8), (2100, 0)])
def test_gradient_quad_kernel_matches_the_oracle(NT, wgs):
    """k_grad_quad (kernels_grad.hip; option "grad_quad"): the gradient GEMM dP*dag(t.v) (fixedL.cc:379,418) of the BASELINE config 3 shape
    with the 240 x 240 accumulators resident in a quad of workgroups -- from 4 096 images per rank on it is the kernel every CG pass of
    an m = 120 bond with the Label index on an environment runs, forced here (grad_quad = 2) at oracle-sized image counts.  300 images = 16
    chunks of 32 on 16 quads (ONE stage each: prologue and epilogue only), 1 100 images = 40 chunks on 3 quads (option bgemm_wgs = 12:
    14 + 14 + 12 stages, ragged), 2 100 images = 72 chunks on 2 quads (36 stages each) and on 64 quads (2 stages on 8 of them, 1 on the others).
    Gradient against the oracle for both bond kinds the kernel serves, bit-identical repeats, the same sums as k_bgemm64 to rounding,
    the CG that runs on it, and a whole bond update."""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N, m = 20, 120
    pixels, labels, phi, W = make_problem(N, NT, m, 7, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_option("grad_quad", 2)
    if wgs:
        ts.set_option("bgemm_wgs", wgs)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    rng = np.random.default_rng(1)
    at = 1
    for b, kind in ((8, "Label on RE"), (12, "Label on LE")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.05 * rng.standard_normal(B.shape)
        ts.profile(True, only="bgemm,grad_quad")
        ts.profile_reset()
        G = ts.gradient(B)
        ts.profile(False)
        pr = ts.profile_read()
        assert pr["grad_quad"][0] == 1 and pr.get("bgemm", (0, 0))[0] == 0      # the kernel under test really ran
        assert _rel(G, o.gradient(B)) < 1e-9, kind
        assert np.array_equal(G, ts.gradient(B)), kind                # fixed summation order: the same bits every time
        ts.set_option("grad_quad", 0)
        assert _rel(G, ts.gradient(B)) < 1e-13, kind                   # k_bgemm64: another summation order, the same sums
        ts.set_option("grad_quad", 2)
        Bg, tg = ts.cgrad(B, 3, 1e-3, 1e-10)
        Bo, to = o.cgrad(B, 3, 1e-3, 1e-10)
        np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-9, err_msg=kind)
        np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=1e-5, err_msg=kind)
        assert _rel(Bg, Bo) < 1e-5, kind
    r = ts.bond_update(12, 1, m, m // 2, 1e-10, 3, 1e-3, 1e-10)
    o.set_bond(12)
    B, _ = o.cgrad(o.bond_tensor(12), 3, 1e-3, 1e-10)
    newm, te, _ = o.svd_split(B, 12, 1, 1e-10, m, m // 2)
    C, lc, cr, nc = o.quadcost(o.bond_tensor(12), 1e-3)
    assert r["newm"] == newm and r["ncorrect"] == nc and r["cost"] == pytest.approx(C, rel=1e-8)
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("m,NT,pair,wgs", [(40, 300, 1, 0), (40, 1100, 1, 6), (40, 300, 0, 0), (61, 300, 1, 0), (64, 2100, 1, 4), (64, 300, 0, 0),
                                           (96, 300, 1, 0), (104, 300, 1, 0), (128, 300, 1, 0)])
def test_gradient_quad_kernel_at_other_bond_dimensions(m, NT, pair, wgs):
    """k_grad_quad serves every bond dimension up to 128 (the 256 x 256 tile grid of its quad form is fixed; rows and links beyond the bond
    dimensions are staged as zeros and come out as the zero padding of the M-layout): bonds that have shrunk below maxm (fixedL.cc:593) and
    unequal left / right dimensions (bond 7 of a 20-site chain: 64 x m), dimensions that are multiples of nothing (61).  Bonds up to 64 x 64 run
    its PAIR form (option grad_pair = 1, the default: 128 x 128 tile grid on two workgroups, two row tiles per wave) -- here with one stage per pair
    (300 images), with 12 + 12 + 11 stages on three pairs (1 100 images, bgemm_wgs = 6) and with 36 stages on each of two pairs (2 100 images,
    bgemm_wgs = 4) -- or, with grad_pair = 0, the quad form.  Unforced the quad form takes bonds from 72 x 72 on and the pair form 33^2 <= mI mO
    <= 56^2 from 15 360 images on; forced here (grad_quad = 2)."""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N = 20
    pixels, labels, phi, W = make_problem(N, NT, m, 9, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_option("grad_quad", 2)
    ts.set_option("grad_pair", pair)
    if wgs:
        ts.set_option("bgemm_wgs", wgs)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    rng = np.random.default_rng(4)
    at = 1
    for b, kind in ((7, "Label on RE, 64 x m"), (8, "Label on RE"), (12, "Label on LE"), (13, "Label on LE, m x 64")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.05 * rng.standard_normal(B.shape)
        ts.profile(True, only="bgemm,grad_quad")
        ts.profile_reset()
        G = ts.gradient(B)
        ts.profile(False)
        pr = ts.profile_read()
        assert pr["grad_quad"][0] == 1 and pr.get("bgemm", (0, 0))[0] == 0, kind
        assert _rel(G, o.gradient(B)) < 1e-9, kind
        assert np.array_equal(G, ts.gradient(B)), kind
        ts.set_option("grad_quad", 0)
        assert _rel(G, ts.gradient(B)) < 1e-13, kind                   # k_bgemm64: another summation order, the same sums
        ts.set_option("grad_quad", 2)
        Bg, tg = ts.cgrad(B, 3, 1e-3, 1e-10)                          # the CG's norms run over the PADDED M-layout: the padding must be zeros
        Bo, to = o.cgrad(B, 3, 1e-3, 1e-10)
        np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-9, err_msg=kind)
        np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=1e-5, err_msg=kind)
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("m,NT,grid", [(61, 300, 0), (64, 300, 0), (90, 1100, 16), (96, 300, 0), (104, 300, 0), (110, 2100, 16), (120, 300, 0)])
def test_resident_forward_and_shift_kernels_at_other_bond_dimensions(m, NT, grid):
    """k_fwd_res / k_shift_res (kernels_res.hip) serve input dimensions 33..120 -- one instantiation per reduction length, rows from the bond
    dimension on staged from valid rows against zero rows of the packed matrix -- and every output dimension their tile grid holds (2..15
    column tiles of 8 links on the pair of workgroups of k_fwd_res; up to 8 column tiles of 16 on the waves of k_shift_res, with at most four
    of them every wave takes one 32-image half): trained bonds shrink towards minm = maxm/2 (fixedL.cc:593) and are not multiples of
    anything (61, 90, 110), left and right dimensions differ (bonds 7 and 13 of a 20-site chain: 64 x m, m x 64).  Forced (options
    fwd_res / shift_res = 2; m = 120: fwd_res = 3, the general form on the benchmark's own bond); grid caps make several rounds per workgroup."""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N = 20
    pixels, labels, phi, W = make_problem(N, NT, m, 11, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_option("fwd_res", 3 if m == 120 else 2)
    ts.set_option("shift_res", 2)
    if grid:
        ts.set_option("res_grid", grid)
    ts.set_mps(W)
    ts.profile(True, only="fgemm_shift,fwd_res,fgemm_fwd,labeldot")
    ts.init()
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    for j in (5, 7, 8, 9, 10):                                # right environments carrying the Label index (64 -> m at site 8, m -> m below)
        assert _rel(ts.env(j), o.env(j)) < 1e-12, j
    rng = np.random.default_rng(6)
    at = 1
    for b, kind in ((7, "Label on RE, 64 x m"), (8, "Label on RE"), (12, "Label on LE"), (13, "Label on LE, m x 64")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
            if bb >= 10:
                assert _rel(ts.env(bb), o.env(bb)) < 1e-12, bb        # left environments carrying the Label index, built by shiftE
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.05 * rng.standard_normal(B.shape)
        ts.profile_reset()
        Pg = ts.forward(B)
        pr = ts.profile_read()
        assert pr["fwd_res"][0] == 1 and pr.get("fgemm_fwd", (0, 0))[0] == 0 and pr.get("labeldot", (0, 0))[0] == 0, (kind, pr)
        assert _rel(Pg, o.forward(B)) < 1e-11, kind
        Cg, lg, _, ng = ts.quadcost(B, 1e-3)
        Co, lo, _, no = o.quadcost(B, 1e-3)
        assert Cg == pytest.approx(Co, rel=1e-11) and ng == no, kind
        Bg, tg = ts.cgrad(B, 3, 1e-3, 1e-10)
        Bo, to = o.cgrad(B, 3, 1e-3, 1e-10)
        np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-9, err_msg=kind)
        np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=1e-5, err_msg=kind)
    ts.profile(False)
    ts.close()


@pytest.mark.gpu
def test_a_truncating_sweep_on_the_resident_kernels_in_lockstep_with_the_oracle():
    """Trained bonds shrink towards minm (fixedL.cc:593) to whatever the spectrum leaves: one whole sweep (both halves) of a 24-site chain that
    starts at m = 100 with maxm = 100, minm = 40 and a cutoff that bites, every bond update compared with the oracle's (new bond dimension,
    cost, #correct, CG trace) and then set to the oracle's state.  The resident kernels are forced (fwd_res / shift_res / grad_quad = 2), so the
    forward pass, the gradient GEMM and the Label-carrying shifts meet the pairs of unequal, odd bond dimensions a real run produces -- the
    instantiations by reduction length, the zero-padded M-layout, clamped rows and links -- not only the hand-picked ones of the tests above."""
    from oracle import pyoracle
    from tnml_amd import lib
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N, NT, m = 24, 200, 100
    maxm, minm, cutoff, npass, lam, cconv = 100, 40, 3e-4, 3, 1e-3, 1e-10
    pixels, labels, phi, W = make_problem(N, NT, m, 13, pixel_boost=200.0)
    ts = TrainStates(labels, N, maxm, phi=phi)
    for k in ("fwd_res", "shift_res", "grad_quad"):
        ts.set_option(k, 2)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    dims = set()
    ts.profile(True, only="fwd_res,fgemm_fwd,grad_quad,bgemm")
    b, ha, n = 1, 1, 0
    while ha <= 2:
        r = ts.bond_update(b, ha, maxm, minm, cutoff, npass, lam, cconv)
        o.set_bond(b)
        B, tr = o.cgrad(o.bond_tensor(b), npass, lam, cconv)
        newm, te, _ = o.svd_split(B, b, ha, cutoff, maxm, minm)
        C, lc, cr, nc = o.quadcost(o.bond_tensor(b), lam)
        o.shiftE(b, ha == 1)
        assert r["newm"] == newm, (b, ha, r["newm"], newm)
        assert r["cost"] == pytest.approx(C, rel=1e-8), (b, ha)
        assert r["ncorrect"] == nc, (b, ha)
        np.testing.assert_allclose(r["cg"]["cost"], tr["cost"], rtol=1e-8, err_msg="bond %d half %d" % (b, ha))
        dims.add((r["mL"], r["mR"], newm))
        ts.set_site(b, o.get_site(b))
        ts.set_site(b + 1, o.get_site(b + 1))
        ts.shiftE(b, ha == 1)
        n += 1
        b, ha = lib.sweepnext(b, ha, N)
    ts.profile(False)
    pr = ts.profile_read()
    assert n == 2 * (N - 1)
    shrunk = sorted(d for d in dims if d[2] < 100 and d[2] > 33)
    print("bond updates %d; (mL, mR, newm) met: %s" % (n, sorted(dims)))
    print("launches: %s" % {k: v[0] for k, v in pr.items() if v[0]})
    assert len({d[2] for d in shrunk}) >= 3, shrunk                       # the sweep really produced several truncated dimensions between 33 and 100
    assert pr["fwd_res"][0] >= 50 and pr["grad_quad"][0] >= 50            # (chain ends below 33 x 16 and the two Label-on-B bonds run the generic kernels)
    ts.close()


@pytest.mark.gpu
def test_m60_kernel_instantiations_match_the_oracle():
    """bonds that have shrunk to minm = maxm/2 = 60 (the reference default, fixedL.cc:593) run their own tiles: 128 x 128
    feature-GEMM tiles (forced here as for C3: at 60 000 images they are the default), 128 x 64 gradient-GEMM tiles"""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N, NT, m = 20, 300, 60
    pixels, labels, phi, W = make_problem(N, NT, m, 5, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_option("fg64_cfg", 2)
    ts.set_option("ldot_cfg", 1)
    ts.set_mps(W)
    ts.init()
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    rng = np.random.default_rng(2)
    at = 1
    for b, kind in ((8, "Label on RE"), (9, "Label on B"), (12, "Label on LE")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.05 * rng.standard_normal(B.shape)
        assert B.shape[:4] == (60, 2, 2, 60)
        assert _rel(ts.forward(B), o.forward(B)) < 1e-11, kind
        assert _rel(ts.gradient(B), o.gradient(B)) < 1e-9, kind
        Bg, tg = ts.cgrad(B, 3, 1e-3, 1e-10)
        Bo, to = o.cgrad(B, 3, 1e-3, 1e-10)
        np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-9, err_msg=kind)
        assert _rel(Bg, Bo) < 1e-5, kind
    r = ts.bond_update(12, 1, m, m // 2, 1e-10, 3, 1e-3, 1e-10)
    o.set_bond(12)
    B, _ = o.cgrad(o.bond_tensor(12), 3, 1e-3, 1e-10)
    newm, te, _ = o.svd_split(B, 12, 1, 1e-10, m, m // 2)
    C, lc, cr, nc = o.quadcost(o.bond_tensor(12), 1e-3)
    assert r["newm"] == newm and r["ncorrect"] == nc and r["cost"] == pytest.approx(C, rel=1e-8)
    ts.close()


@pytest.mark.gpu
def test_c2_at_its_stated_shape():
    """BASELINE config 2: N=784, maxm=20, 1000 images per label.  Size-independent properties on the HIP path (linearity
    of the forward map, gradient additivity over image shards, cost independent of the bond) and, against the oracle on
    the same inputs, the first two bond updates of a sweep."""
    from oracle import pyoracle
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, NT, m = 784, 10000, 20
    labels = synth.synthetic_labels(NT, per_label=NT // 10)
    pixels = synth.synthetic_images(N, labels)
    phi = synth.features_series(pixels)
    phi[..., 1] *= 255.0                                                    # [1, x/4] (README.md:74)
    W = synth.random_mps(N, m, seed=5)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_mps(W)
    ts.init()
    B1 = ts.bond_tensor(1)
    rng = np.random.default_rng(0)
    B2 = rng.standard_normal(B1.shape)
    P1, P2, P12 = ts.forward(B1), ts.forward(B2), ts.forward(B1 + 2.0 * B2)
    assert _rel(P12, P1 + 2.0 * P2) < 1e-12
    G = ts.gradient(B1)
    half = NT // 2
    parts = []
    for sl in (slice(0, half), slice(half, NT)):
        t2 = TrainStates(labels[sl], N, m, phi=phi[sl])
        t2.set_mps(W)
        t2.init()
        parts.append(t2.gradient(B1))
        t2.close()
    assert _rel(parts[0] + parts[1], G) < 1e-11
    c1 = ts.quadcost(B1, 0.0)[0]
    # the oracle on the same inputs: initial cost and the first two bond updates
    o = pyoracle.Oracle(phi, labels, W, nthread=min(16, os.cpu_count() or 1))
    o.init()
    assert c1 == pytest.approx(o.quadcost(o.bond_tensor(1), 0.0)[0], rel=1e-11)
    assert _rel(ts.env(3), o.env(3)) < 1e-12 and _rel(ts.env(N), o.env(N)) < 1e-12
    ro = o.mldmrg(1, m, m // 2, 1e-10, 4, 1e-3, 1e-10, max_bonds=2)
    for k, b in enumerate((1, 2)):
        r = ts.bond_update(b, 1, m, m // 2, 1e-10, 4, 1e-3, 1e-10)
        assert r["newm"] == ro[k]["newm"] and r["ncorrect"] == ro[k]["ncorrect"]
        assert r["cost"] == pytest.approx(ro[k]["cost"], rel=1e-8)
    # the cost does not depend on the bond it is evaluated at (gauge invariance), far into the chain
    for b in range(3, 41):
        ts.shiftE(b, True)
    ts.setBond(41)
    c41, _, cr41, n41 = ts.quadcost(ts.bond_tensor(41), 1e-3)
    ts41 = c41 - cr41                                                        # sum_l C_l: a property of the network, not of the bond
    assert ts41 == pytest.approx(ro[1]["cost"] - ro[1]["reg_cost"], rel=1e-9) and n41 == ro[1]["ncorrect"]
    ts.close()


def _hard_images(N, labels, seed, mix=0.8):
    """overlapping classes: every image is its label's template blended with another label's (weight `mix`) plus the
    generator's noise, so that a classifier trained for two sweeps at m = 10 is right on ~90-97 % of the images and a
    0.1 % difference (10 images of 10 000) is visible"""
    from tnml_amd import synth
    labels = np.asarray(labels)
    rng = np.random.default_rng(seed)
    other = (labels + rng.integers(1, 10, size=labels.shape)) % 10
    a = synth.synthetic_images(N, labels, seed=seed).astype(np.float64)
    b = synth.synthetic_images(N, other, seed=seed + 17).astype(np.float64)
    w = rng.uniform(0.0, mix, size=(len(labels), 1))
    return np.clip(np.rint((1.0 - w) * a + w * b), 0, 255).astype(np.uint8)


@pytest.mark.gpu
def test_test_set_accuracy_within_a_tenth_of_a_percent_on_10000_images(tmp_path):
    """north_star: "test-set accuracy within 0.1 % of the CPU reference".  Training set: C1 shape (N=196, 100 per
    label, maxm=10, 2 sweeps) on overlapping synthetic classes; test set: 10 000 images of the same distribution, so
    one image is 0.01 %.  (a) The SAME trained network evaluated by the `fulltest` binary (tnml_classify) and by the
    oracle's toverlap: identical prediction vectors.  (b) The network trained by the `fixedL` binary vs the one trained
    by the oracle from the same initial W: test accuracies within 0.1 % + the oracle's own 1-vs-8-thread spread."""
    from oracle import pyoracle
    from tnml_amd import hostlib, synth
    N, per_label, NTEST = 196, 100, 10000
    ltr = synth.synthetic_labels(10 * per_label, seed=11, per_label=per_label)
    lte = synth.synthetic_labels(NTEST, seed=12, per_label=NTEST // 10)
    pall = _hard_images(N, np.concatenate([ltr, lte]), 11)     # one call: the label templates depend on the seed
    ptr, pte = pall[:len(ltr)], pall[len(ltr):]
    order = np.argsort(ltr, kind="stable")                       # any file order; the driver keeps the first 100 of each label
    data = str(tmp_path / "data")
    synth.write_idx(data, ptr[order], ltr[order], side=14)
    synth.write_idx(data, pte, lte, train=False, side=14)
    keys = "datadir = %s\nfeature_scale = 255\n" % data
    inp = tmp_path / "input"
    inp.write_text("input\n{\n%sNtrain = %d\nNbatch = 10\nNsweep = 2\ncutoff = 1E-10\nmaxm = 10\nminm = 10\nninitial = 5\n"
                   "lambda = 1E-3\nNpass = 4\nseed = 3\n}\n" % (keys, per_label))
    w0 = str(tmp_path / "W0ref")
    hostlib.build_initial_w(data, per_label, 5, 3, w0, feature_scale=255.0)
    run = subprocess.run([os.path.join(ROOT, "tnml_amd", "fixedL"), str(inp)], capture_output=True, text=True, cwd=tmp_path, timeout=900)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    tin = tmp_path / "input_test"
    tin.write_text("input\n{\n%s}\n" % keys)

    def fulltest():
        ev = subprocess.run([os.path.join(ROOT, "tnml_amd", "fulltest"), str(tin)], capture_output=True, text=True, cwd=tmp_path, timeout=900)
        assert ev.returncode == 0, ev.stdout[-1500:] + ev.stderr[-1500:]
        mm = re.search(r"(\d+)/(\d+) correct \(([0-9.]+)%\)", ev.stdout)
        assert mm and int(mm.group(2)) == NTEST
        return int(mm.group(1))
    gpu_trained_correct = fulltest()

    def feats(p):
        gg = p.astype(np.float64) / 255.0
        return np.stack([np.ones_like(gg), 255.0 * ((gg / 255.0) / 4.0)], axis=-1)
    trp, trl, _ = hostlib.read_mnist(data, True, per_label)
    runs = {}
    for nth in (1, 8):
        o = pyoracle.Oracle(feats(trp), trl, hostlib.read_mps(w0), nthread=nth)
        o.init()
        o.mldmrg(2, 10, 10, 1e-10, 4, 1e-3, 1e-10)
        Wo = o.get_mps()
        ot = pyoracle.Oracle(feats(pte), lte, Wo, nthread=1)
        wt = np.stack([ot.toverlap(i) for i in range(NTEST)])
        runs[nth] = dict(W=Wo, pred=np.abs(wt).argmax(axis=1), correct=int((np.abs(wt).argmax(axis=1) == lte).sum()))
    # (a) same network, two inference paths
    hostlib.write_mps(str(tmp_path / "W"), runs[8]["W"])
    same_net_correct = fulltest()
    assert same_net_correct == runs[8]["correct"]
    # (b) two trainings
    spread = abs(runs[1]["correct"] - runs[8]["correct"])
    d = min(abs(gpu_trained_correct - runs[nth]["correct"]) for nth in (1, 8))
    print("10 000 test images: GPU-trained %d correct, oracle-trained %d (1 thread) / %d (8 threads); same network through "
          "fulltest and toverlap: %d / %d" % (gpu_trained_correct, runs[1]["correct"], runs[8]["correct"], same_net_correct, runs[8]["correct"]))
    assert 0.80 * NTEST < runs[8]["correct"] < 0.995 * NTEST             # hard enough to resolve, easy enough to have learned
    assert d <= 0.001 * NTEST + spread


def _forward_with_fp32_stored_environments(phi, W, b, Bt):
    """B*t.v of bond b for a direction Bt with the storage format of TNML_F64_E32 modelled exactly: features and every environment are
    rounded to fp32 where the library stores them (once per shift, fixedL.cc:142-149,221-228), all arithmetic in fp64.
    phi [NT][N][2]; W = the N site tensors [l][s][r]([L]) at the moment the bond is evaluated."""
    r32 = lambda x: x.astype(np.float32).astype(np.float64)
    f = r32(np.asarray(phi, dtype=np.float64))
    NT, N, _ = f.shape
    L = None
    for j in range(1, b):                                             # left environments, built left to right by shiftE
        M = np.einsum('ns,asr...->nar...', f[:, j - 1], W[j - 1])
        if L is None:
            L = M[:, 0]
        else:
            L = np.einsum('narl,na->nrl' if M.ndim == 4 else ('nar,nal->nrl' if L.ndim == 3 else 'nar,na->nr'), M, L)
        L = r32(L)
    R = None
    for j in range(N, b + 1, -1):                                     # right environments, built right to left by init
        M = np.einsum('ns,asr...->nar...', f[:, j - 1], W[j - 1])
        if R is None:
            R = M[:, :, 0]
        else:
            R = np.einsum('narl,nr->nal' if M.ndim == 4 else ('nar,nrl->nal' if R.ndim == 3 else 'nar,nr->na'), M, R)
        R = r32(R)
    fs, ft = f[:, b - 1], f[:, b]
    L = np.ones((NT, 1)) if L is None else L                          # chain ends
    R = np.ones((NT, 1)) if R is None else R
    if Bt.ndim == 5:
        return np.einsum('astrl,na,ns,nt,nr->nl', Bt, L, fs, ft, R, optimize=True)
    if L.ndim == 3:
        return np.einsum('astr,nal,ns,nt,nr->nl', Bt, L, fs, ft, R, optimize=True)
    return np.einsum('astr,na,ns,nt,nrl->nl', Bt, L, fs, ft, R, optimize=True)


@pytest.mark.gpu
def test_c5_bond_updates_at_maxm_300_in_fp64_fp32_and_bf16():
    """BASELINE config 5 (maxm = 300: fp64 vs fp32 vs bf16 bond contraction) on the HIP kernels, bond update by bond update in
    lockstep with the oracle: bonds 10..13 of a 24-site chain at m = 300 (Label on the right environment, on B twice, on the left
    environment), 24 images (the oracle's dense t.v is 29 MB per image at this bond dimension).  Each arithmetic starts every bond
    update from the oracle's state.  The split is the 600 x 600 (6000 on the Label-on-B bonds) problem of eigh_mc.hip.
      f64      fp64 MFMA, fp64 storage                         gated: cost 1e-8, bond dimension and #correct exact
      f64_e32  fp64 MFMA over fp32-stored environments         gated against the storage format itself, not against a fitted number: the
                                                               same evaluation in numpy with features and environments rounded to fp32
                                                               once per shift and fp64 arithmetic (_forward_with_fp32_stored_environments)
                                                               must agree with the HIP path to 1e-7 -- they differ in the fp64 summation
                                                               order only, and a value that lands on the other side of an fp32 rounding
                                                               boundary moves ONE entry of one environment by 6e-8 of itself (measured:
                                                               4e-15).  What the format costs against fp64 on THIS chain (9 + 11 roundings
                                                               through m = 300 sites) is that model's own deviation, 2.2e-7, and the HIP
                                                               path must not exceed twice it; after-SVD cost 1e-6.
                                                               (Round 3 measured 1.04e-5 here and widened its gate to fit: the shifts of
                                                               this mode ran on the fp32 matrix pipe with fp32 accumulation -- the model
                                                               exposed it; they run on the fp64 pipe and round once, on the store, now.)
      f32      v_mfma_f32_16x16x4_f32                          interior bonds gated at 1e-3
      bf16x3   v_mfma_f32_16x16x32_bf16, operands hi + lo      forward map gated at 1e-4 (the kernel's operand layout), costs reported
      bf16     v_mfma_f32_16x16x32_bf16                        forward map gated at 3e-2, costs reported (report-don't-gate)
    The printed table is the m = 300 row set of DESIGN.md's tolerance study."""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    from conftest import make_problem
    N, NT, m = 24, 24, 300
    npass, lam, cconv, cutoff, minm = 4, 1e-3, 1e-10, 1e-10, 150
    pixels, labels, phi, W = make_problem(N, NT, m, 5, pixel_boost=200.0)
    bonds = [10, 11, 12, 13]
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    for bb in range(1, bonds[0]):
        o.shiftE(bb, True)
    ref = []                                                                   # the oracle's bond updates, once for all arithmetics
    rng = np.random.default_rng(0)
    for b in bonds:
        o.set_bond(b)
        B0 = o.bond_tensor(b)
        Bt = B0 + 0.05 * np.abs(B0).max() * rng.standard_normal(B0.shape)       # a direction for the single-evaluation check
        Pt = o.forward(Bt)
        before = (o.get_site(b).copy(), o.get_site(b + 1).copy())
        Pt32 = _forward_with_fp32_stored_environments(phi, o.get_mps(), b, Bt)
        B, tr = o.cgrad(B0, npass, lam, cconv)
        newm, te, _ = o.svd_split(B, b, 1, cutoff, m, minm)
        C, lc, cr, nc = o.quadcost(o.bond_tensor(b), lam)
        o.shiftE(b, True)
        ref.append(dict(b=b, Bt=Bt, Pt=Pt, Pt32=Pt32, before=before, after=(o.get_site(b).copy(), o.get_site(b + 1).copy()), newm=newm, te=te, C=C, nc=nc,
                        cg=tr["cost"]))
    rows = []
    for dtype in ("f64", "f64_e32", "f32", "bf16x3", "bf16"):
        ts = TrainStates(labels, N, m, phi=phi, dtype=dtype)
        ts.set_mps(W)
        ts.init()
        for bb in range(1, bonds[0]):
            ts.shiftE(bb, True)
        worst = {"interior": 0.0, "label_on_B": 0.0}
        fwd = 0.0
        fwd_model = 0.0
        dm = dn = 0
        for r0 in ref:
            b = r0["b"]
            ts.set_site(b, r0["before"][0]); ts.set_site(b + 1, r0["before"][1])
            ts.setBond(b)
            Pg = ts.forward(r0["Bt"])
            fwd = max(fwd, _rel(Pg, r0["Pt"]))
            if dtype == "f64_e32":
                fwd_model = max(fwd_model, _rel(Pg, r0["Pt32"]))
            r = ts.bond_update(b, 1, m, minm, cutoff, npass, lam, cconv)
            kind = "label_on_B" if r["label_on_B"] else "interior"
            worst[kind] = max(worst[kind], abs(r["cost"] / r0["C"] - 1))
            dm += int(r["newm"] != r0["newm"]); dn += abs(int(r["ncorrect"]) - int(r0["nc"]))
            if dtype == "f64":
                np.testing.assert_allclose(r["cg"]["cost"][:npass - 1], r0["cg"][:npass - 1], rtol=1e-8)
                assert r["truncerr"] == pytest.approx(r0["te"], rel=1e-5, abs=1e-16)
            ts.set_site(b, r0["after"][0]); ts.set_site(b + 1, r0["after"][1])   # lockstep: the next bond starts from the oracle's state
            ts.shiftE(b, True)
        stats = ts.svd_stats()
        rows.append((dtype, fwd, worst["interior"], worst["label_on_B"], dm, dn, stats["fallbacks"]))
        if dtype == "f64_e32":
            e32_model = fwd_model
        ts.close()
    print("\nC5 (maxm = 300) lockstep, 4 bond updates per arithmetic: max rel. error of one forward evaluation | of the after-SVD cost, interior bonds | "
          "Label-on-B bonds | bond-dimension mismatches | #correct differences | eigensolver fallbacks")
    for rw in rows:
        print("  %-8s %.2e | %.2e | %.2e | %d | %d | %d" % rw)
    predicted = max(_rel(r0["Pt32"], r0["Pt"]) for r0 in ref)
    print("  fp32 storage modelled in numpy: deviation from fp64 %.2e (the format's cost on this chain) | HIP f64_e32 against the model %.2e" % (predicted, e32_model))
    by = {r[0]: r for r in rows}
    assert e32_model < 1e-7 and by["f64_e32"][1] < 2 * predicted
    assert by["f64"][1] < 1e-11 and max(by["f64"][2], by["f64"][3]) < 1e-8 and by["f64"][4] == 0 and by["f64"][5] == 0 and by["f64"][6] == 0
    assert max(by["f64_e32"][2], by["f64_e32"][3]) < 1e-6 and by["f64_e32"][4] == 0
    assert by["f32"][1] < 1e-4 and by["f32"][2] < 1e-3
    assert by["bf16x3"][1] < 1e-4                                              # a wrong operand layout would give O(1)
    assert by["bf16"][1] < 3e-2
