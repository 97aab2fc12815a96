"""CPU tests of the per-label variant's oracle (oracle/single_oracle.c, restating single.cc / single.h) against the
independently written numpy restatement (oracle/np_restatement.py NpSingle) and its own invariants.
PARITY UNPINNED as for fixedL: the reference ships no tests and cannot be built here."""
import numpy as np
import pytest

from oracle import np_restatement as npr
from oracle import pyoracle
from tnml_amd import synth


def plain_mps(N, m, seed):
    W = synth.random_mps(N, m, seed=seed)
    c0 = N // 2
    W[c0 - 1] = W[c0 - 1][..., 0] * 3.0                   # drop the Label index of the fixedL generator
    return W


def problem(N=10, NT=50, m=4, seed=2, normal=True, boost=1.0):
    labels = synth.synthetic_labels(NT, seed=seed, per_label=NT // 10)
    pixels = synth.synthetic_images(N, labels, seed=seed)
    if boost != 1.0:
        pixels = np.clip(pixels.astype(np.int32) * boost, 0, 255).astype(np.uint8)
    phi = pyoracle.features_single(pixels, normal)
    return pixels, labels, phi, plain_mps(N, m, seed + 3)


def test_features_match_numpy():
    pixels, _, phi, _ = problem()
    np.testing.assert_allclose(phi, npr.features_single(pixels, True), rtol=1e-15)
    np.testing.assert_allclose(pyoracle.features_single(pixels, False), npr.features_single(pixels, False), rtol=1e-15)


@pytest.mark.parametrize("nthread", [1, 3])
def test_single_oracle_matches_numpy_restatement(nthread):
    pixels, labels, phi, W = problem()
    phi = phi.copy(); phi[..., 1] *= 300.0               # well-conditioned second component for a meaningful CG comparison
    o = pyoracle.SingleOracle(phi, labels, 3, W, nthread=nthread)
    n = npr.NpSingle(phi, labels, 3, W)
    o.init(); n.init()
    for j in range(3, 11):
        np.testing.assert_allclose(o.env(j), n.E[j], rtol=1e-12, atol=1e-14)
    for b in (1, 2, 5, 9):
        if b > 1:
            for bb in range(1, b):
                o.shiftE(bb, True); n.shiftE(bb, True)
        o.set_bond(b); n.set_bond(b)
        B = o.bond_tensor(b)
        np.testing.assert_allclose(B, n.bond_tensor(b), rtol=1e-12, atol=1e-14)
        B = B + 0.1 * np.random.default_rng(b).standard_normal(B.shape)
        np.testing.assert_allclose(o.forward(B), n.forward(B), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(o.gradient(B), n.gradient(B), rtol=1e-10, atol=1e-12)
        assert o.quadcost(B, 1e-3)[0] == pytest.approx(n.quadcost(B, 1e-3), rel=1e-12)
        Bo, to = o.cgrad(B, 4, 1e-3, 1e-10)
        Bn, tn = n.cgrad(B, 4, 1e-3, 1e-10)
        np.testing.assert_allclose(to["cost"], tn["cost"], rtol=1e-10)
        np.testing.assert_allclose(to["alpha"], tn["alpha"], rtol=1e-8)
        np.testing.assert_allclose(Bo, Bn, rtol=1e-7, atol=1e-9)
        # rebuild the envs for the next bond from scratch
        o.init(); n.init()


def test_single_sweep_matches_numpy_and_decreases_cost():
    pixels, labels, phi, W = problem(N=8, NT=60, m=3)
    phi = phi.copy(); phi[..., 1] *= 300.0
    o = pyoracle.SingleOracle(phi, labels, 1, W, nthread=2)
    n = npr.NpSingle(phi, labels, 1, W)
    o.init(); n.init()
    ro = o.mldmrg(2, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    rn = n.mldmrg(2, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(ro) == len(rn) == 2 * 2 * 7
    for a, b in zip(ro, rn):
        assert (a["c"], a["half"], a["origm"], a["newm"]) == (b["c"], b["half"], b["origm"], b["newm"])
        assert a["cost_old"] == pytest.approx(b["cost_old"], rel=1e-8)
        assert a["cost_cg"] == pytest.approx(b["cost_cg"], rel=1e-8)
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-8)
        assert a["cost_cg"] <= a["cost_old"] * (1 + 1e-12)              # the CG never increases the cost
    assert ro[-1]["cost"] < 0.8 * ro[0]["cost_old"]
    # decision function = full contraction, and consistent with the cost at any bond
    f = np.array([o.output(i) for i in range(o.NT)])
    np.testing.assert_allclose(f, [n.output(i) for i in range(n.NT)], rtol=1e-6, atol=1e-8)
    y = (labels == 1).astype(float)
    o.set_bond(1)
    B = o.bond_tensor(1)
    assert np.sum((y - f) ** 2) + 1e-3 * np.sum(B * B) == pytest.approx(o.quadcost(B, 1e-3)[0], rel=1e-10)


def test_cgrad_not_optimizing_branch():
    """single.h:202-206: |r| < cconv at entry returns without touching B"""
    pixels, labels, phi, W = problem()
    o = pyoracle.SingleOracle(phi, labels, 0, W)
    o.init()
    B = o.bond_tensor(1)
    B2, tr = o.cgrad(B, 4, 1e-3, 1e30)
    assert tr["skipped"] and np.array_equal(B2, B)


@pytest.mark.parametrize("nthread", [1, 3])
def test_fast_cgrad_matches_numpy_and_is_cgrad_without_the_regulariser(nthread):
    """method = fast_conj (single.h:290-398): the C restatement against the numpy one; with lambda = 0 the residual
    recurrence is exact, so fast_cgrad and cgrad agree to rounding; with lambda > 0 the reference's
    'nr = nr - lambda*B' (:379) makes them differ from the second pass on -- reproduced, not repaired"""
    pixels, labels, phi, W = problem()
    phi = phi.copy(); phi[..., 1] *= 300.0
    o = pyoracle.SingleOracle(phi, labels, 3, W, nthread=nthread)
    n = npr.NpSingle(phi, labels, 3, W)
    o.init(); n.init()
    for b in (1, 5):
        if b > 1:
            for bb in range(1, b):
                o.shiftE(bb, True); n.shiftE(bb, True)
        o.set_bond(b); n.set_bond(b)
        B = o.bond_tensor(b) + 0.1 * np.random.default_rng(b).standard_normal(o.bond_tensor(b).shape)
        for lam in (0.0, 1e-3):
            Bo, to = o.fast_cgrad(B, 4, lam, 1e-10)
            Bn, tn = n.fast_cgrad(B, 4, lam, 1e-10)
            np.testing.assert_allclose(to["alpha"], tn["alpha"], rtol=1e-8)
            np.testing.assert_allclose(to["rnorm"], tn["rnorm"], rtol=1e-7)
            np.testing.assert_allclose(Bo, Bn, rtol=1e-7, atol=1e-9)
            assert to["cost"] == [] and len(to["rnorm"]) == 3
            Bc, tc = o.cgrad(B, 4, lam, 1e-10)
            np.testing.assert_allclose(to["alpha"][0], tc["alpha"][0], rtol=1e-12)      # the first step is the same computation
            if lam == 0.0:
                np.testing.assert_allclose(to["alpha"], tc["alpha"], rtol=1e-7)
                np.testing.assert_allclose(to["rnorm"], tc["rnorm"], rtol=1e-6)
                np.testing.assert_allclose(Bo, Bc, rtol=1e-6, atol=1e-8)
            else:
                assert abs(to["alpha"][1] / tc["alpha"][1] - 1) > 1e-9               # the as-written regulariser term shows
        o.init(); n.init()


def test_fast_conj_sweep_and_entry_check():
    pixels, labels, phi, W = problem(N=8, NT=60, m=3)
    phi = phi.copy(); phi[..., 1] *= 300.0
    o = pyoracle.SingleOracle(phi, labels, 1, W, nthread=2)
    o.init()
    o.set_method("fast_conj")
    ro = o.mldmrg(1, 4, 2, 1e-10, 3, 0.0, 1e-10)
    o2 = pyoracle.SingleOracle(phi, labels, 1, W, nthread=2)
    o2.init()
    rc = o2.mldmrg(1, 4, 2, 1e-10, 3, 0.0, 1e-10)
    assert len(ro) == len(rc) == 14
    assert ro[0]["cg_alpha"] == pytest.approx(rc[0]["cg_alpha"], rel=1e-7)              # lambda = 0: the same optimiser
    assert ro[0]["cost_cg"] == pytest.approx(rc[0]["cost_cg"], rel=1e-8)
    assert all(r["cg_cost"] == [0.0] * len(r["cg_cost"]) for r in ro)                   # fast_cgrad prints no cost
    assert ro[-1]["cost"] < ro[0]["cost_old"]
    B = o.bond_tensor(1)
    o.set_bond(1)
    B2, tr = o.fast_cgrad(B, 4, 1e-3, 1e30)
    assert tr["skipped"] and np.array_equal(B2, B)


@pytest.mark.parametrize("lam", [1e-3, 0.0])
def test_exact_solver_matches_numpy_and_solves_the_normal_equations(lam):
    """method = exact (single.h:117-160): B = y Phi^+ with the filtered inverse s/(s^2 + lambda) above pcut -- C restatement
    (own Jacobi SVD) against numpy.linalg.svd, and the defining property: the regularised residual vanishes"""
    pixels, labels, phi, W = problem(N=10, NT=90, m=3)
    phi = phi.copy(); phi[..., 1] *= 300.0
    o = pyoracle.SingleOracle(phi, labels, 3, W, nthread=2)
    n = npr.NpSingle(phi, labels, 3, W)
    o.init(); n.init()
    for b in (1, 4):
        if b > 1:
            for bb in range(1, b):
                o.shiftE(bb, True); n.shiftE(bb, True)
        o.set_bond(b); n.set_bond(b)
        D = int(np.prod(o.bond_shape(b)))
        sv = np.linalg.svd(n.v.reshape(n.v.shape[0], -1), compute_uv=False)
        pcut = 0.5 * (sv[min(D, len(sv)) - 1] + 0.0) if lam > 0 else 1e-6 * sv[0]    # lambda = 0 needs the cut (rank-deficient Phi)
        Bo = o.exact(b, lam, pcut)
        Bn = n.exact(lam, pcut)
        np.testing.assert_allclose(Bo, Bn, rtol=1e-7, atol=1e-9 * np.abs(Bn).max())
        if lam > 0:
            r = o.gradient(Bo) - lam * Bo
            assert np.abs(r).max() < 1e-8 * np.abs(o.gradient(np.zeros_like(Bo))).max()
            assert o.quadcost(Bo, lam)[0] <= o.cgrad(o.bond_tensor(b), 6, lam, 1e-12)[1]["cost"][-1] * (1 + 1e-9)   # no CG does better
        o.init(); n.init()


def test_exact_method_in_the_sweep():
    pixels, labels, phi, W = problem(N=8, NT=60, m=3)
    phi = phi.copy(); phi[..., 1] *= 300.0
    o = pyoracle.SingleOracle(phi, labels, 1, W, nthread=2)
    o.init()
    o.set_method("exact", pcut=1e-8)
    ro = o.mldmrg(1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(ro) == 14 and all(r["cg_alpha"] == [] for r in ro)
    assert all(r["cost_cg"] <= r["cost_old"] * (1 + 1e-9) for r in ro)     # the exact minimiser never does worse than the incoming tensor
    assert ro[-1]["cost"] < 0.5 * ro[0]["cost_old"]


@pytest.mark.parametrize("noise", [1e-6, 1e-2])
def test_noise_split_matches_numpy_and_is_a_gauge_of_the_bond_tensor(noise):
    """single.h:648-672: the density-matrix split with a noise term, C oracle against the einsum restatement at interior bonds in both
    half sweeps and at the chain ends (no environment there: drho = NT rho).  The basis on site c is orthonormal, the other site is
    UU * B, and without truncation the product of the two sites is B again whatever the noise."""
    pixels, labels, phi, W = problem(N=8, NT=40, m=3)
    phi = phi.copy(); phi[..., 1] *= 300.0
    for ha, bonds in ((1, (1, 3, 5)), (2, (7, 5, 2))):
        o = pyoracle.SingleOracle(phi, labels, 1, W, nthread=2)
        n = npr.NpSingle(phi, labels, 1, W)
        o.init(); n.init()
        if ha == 2:                                                  # half sweep 2 runs right to left: build the left environments first
            for b in range(1, 8):
                o.shiftE(b, True); n.shiftE(b, True)
        walk = range(1, 8) if ha == 1 else range(7, 0, -1)
        for b in walk:
            if b in bonds:
                B = o.bond_tensor(b) * (1.0 + 0.1 * np.cos(np.arange(o.bond_tensor(b).size)).reshape(o.bond_tensor(b).shape))
                full = 2 * max(B.shape[0], B.shape[3])
                mo, teo = o.noise_split(B, b, ha, noise, 0.0, full, full)     # keep everything
                mn, ten = n.noise_split(B, b, ha, noise, 0.0, full, full)
                assert mo == mn
                np.testing.assert_allclose(o.bond_tensor(b), B, rtol=1e-9, atol=1e-11)
                np.testing.assert_allclose(n.bond_tensor(b), B, rtol=1e-9, atol=1e-11)
                A = o.get_site(b if ha == 1 else b + 1)
                G = np.einsum('asg,ash->gh', A, A) if ha == 1 else np.einsum('gtr,htr->gh', A, A)
                np.testing.assert_allclose(G, np.eye(G.shape[0]), atol=1e-10)
                # with truncation both restatements keep the same subspace: compare the truncated bond tensors
                keep = max(1, min(B.shape[0], B.shape[3]))
                mo, teo = o.noise_split(B, b, ha, noise, 1e-12, keep, 1)
                mn, ten = n.noise_split(B, b, ha, noise, 1e-12, keep, 1)
                assert mo == mn and teo == pytest.approx(ten, rel=1e-6, abs=1e-14)
                np.testing.assert_allclose(o.bond_tensor(b), n.bond_tensor(b), rtol=1e-6, atol=1e-9)
                # restore the original sites so that the walk goes on from the same network
                for j in (b, b + 1):
                    o.set_site(j, W[j - 1]); n.W[j] = np.array(W[j - 1])
            o.shiftE(b, ha == 1); n.shiftE(b, ha == 1)


def test_sweep_with_noise_matches_numpy():
    pixels, labels, phi, W = problem(N=8, NT=60, m=3)
    phi = phi.copy(); phi[..., 1] *= 300.0
    o = pyoracle.SingleOracle(phi, labels, 1, W, nthread=2)
    n = npr.NpSingle(phi, labels, 1, W)
    o.set_noise(1e-5); n.noise = 1e-5
    o.init(); n.init()
    ro = o.mldmrg(2, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    rn = n.mldmrg(2, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(ro) == len(rn) == 2 * 2 * 7
    for a, b in zip(ro, rn):
        assert (a["c"], a["half"], a["origm"], a["newm"]) == (b["c"], b["half"], b["origm"], b["newm"])
        assert a["cost_cg"] == pytest.approx(b["cost_cg"], rel=1e-7)
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-7)
    # the noise changes the kept subspace: not the same run as without it
    o0 = pyoracle.SingleOracle(phi, labels, 1, W, nthread=2)
    o0.init()
    r0 = o0.mldmrg(2, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert any(abs(a["cost"] - b["cost"]) > 1e-9 * abs(b["cost"]) for a, b in zip(ro, r0))


@pytest.mark.parametrize("b,r", [(1, 3), (4, 6), (9, 4)])
def test_pinv_matches_numpy_and_approaches_the_exact_solver(b, r):
    """single.h:404-517 from a given start (the reference's is random and time-seeded): C oracle against the numpy restatement -- the V*E
    trace, the singular values, the resulting B -- and, with r = D (the whole space), the iteration is the exact solver"""
    pixels, labels, phi, W = problem(N=10, NT=50, m=3)
    phi = phi.copy(); phi[..., 1] *= 300.0
    o = pyoracle.SingleOracle(phi, labels, 1, W, nthread=1)
    n = npr.NpSingle(phi, labels, 1, W)
    o.init(); n.init()
    for bb in range(1, b):
        o.shiftE(bb, True); n.shiftE(bb, True)
    o.set_bond(b); n.set_bond(b)
    D = int(np.prod(o.bond_shape(b)))
    V0 = np.random.default_rng(b).standard_normal((D, min(r, D)))
    Bo, veo, Do = o.pinv(b, V0, 6, 1e-3)
    Bn, ven, Dn = n.pinv(V0, 6, 1e-3)
    assert len(veo) == len(ven)
    np.testing.assert_allclose(veo, ven, rtol=1e-9)
    np.testing.assert_allclose(Do, Dn, rtol=1e-8)
    np.testing.assert_allclose(Bo, Bn, rtol=1e-6, atol=1e-9 * np.abs(Bn).max())
    assert all(veo[i + 1] >= veo[i] * (1 - 1e-12) for i in range(len(veo) - 1))      # the trace norm of V^T A grows towards the top-r eigenvalue sum
    # the whole space: E = V^T A with V orthogonal, so yUS Einv = y Phi A^+ filtered -- the exact solver's B up to the different filter
    # argument (pinv filters the eigenvalues of A, exact the singular values of Phi): compare with lambda = 0, pcut = 0 on a full-rank case
    if D <= 24:
        Vf = np.random.default_rng(7).standard_normal((D, D))
        Bf, _, _ = o.pinv(b, Vf, 2, 0.0, 0.0)
        Phi = np.stack([x.reshape(-1, order="F") for x in n.v])
        y = (labels == 1).astype(float)
        if np.linalg.matrix_rank(Phi) == D:
            np.testing.assert_allclose(Bf.reshape(-1, order="F"), np.linalg.solve(Phi.T @ Phi, Phi.T @ y), rtol=1e-5, atol=1e-8)
