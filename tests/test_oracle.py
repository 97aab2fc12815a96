"""CPU tests of the oracle: C restatement (oracle/fixedl_oracle.c) vs the independent numpy
restatement (oracle/np_restatement.py), plus the invariants of SURVEY.md section 4.

The reference ships no golden vectors (parity unpinned), so these cross-checks and the committed
fixtures under tests/golden/ are the pins."""
import numpy as np
import pytest

from conftest import make_problem
from oracle import np_restatement as npr
from oracle import pyoracle

RTOL = 1e-10


def both(N=12, NT=60, m=4, seed=3, nthread=1, nbatch=1, boost=200.0):
    pixels, labels, phi, W = make_problem(N, NT, m, seed, pixel_boost=boost)
    o = pyoracle.Oracle(phi, labels, W, nthread=nthread, nbatch=nbatch)
    n = npr.NpFixedL(phi, labels, W)
    o.init()
    n.init()
    return o, n


def test_features_match():
    pixels, labels, phi, W = make_problem()
    np.testing.assert_allclose(pyoracle.features_series(pixels), npr.features_series(pixels), rtol=0, atol=0)
    assert phi[..., 1].max() <= 255 / 260100 + 1e-15


def test_sweepnext_order():
    N = 6
    seq, b, ha = [], 1, 1
    while ha <= 2:
        seq.append((b, ha))
        b, ha = pyoracle.sweepnext(b, ha, N)
    assert seq == [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (5, 2), (4, 2), (3, 2), (2, 2), (1, 2)]
    assert seq == [(x, y) for x, y in _np_seq(N)]


def _np_seq(N):
    b, ha = 1, 1
    while ha <= 2:
        yield b, ha
        b, ha = npr.sweepnext(b, ha, N)


@pytest.mark.parametrize("p,maxm,minm,cutoff,expect", [
    ([4.0, 2.0, 1.0, 1e-12, 1e-13], 10, 1, 1e-10, 3),      # relative cutoff drops the tail
    ([4.0, 2.0, 1.0, 1e-12, 1e-13], 10, 4, 1e-10, 4),      # minm keeps one more
    ([4.0, 2.0, 1.0, 0.5, 0.25], 2, 1, 1e-10, 2),          # maxm wins
    ([1.0], 5, 1, 1e-10, 1),
    ([0.0, 0.0, 0.0], 5, 2, 1e-10, 2),                     # all-zero spectrum: scale -> 1, keeps minm
])
def test_truncate_rule(p, maxm, minm, cutoff, expect):
    m_c, te_c = pyoracle.truncate(p, maxm, minm, cutoff)
    m_n, te_n = npr.truncate(np.array(p), maxm, minm, cutoff)
    assert m_c == m_n == expect
    assert te_c == pytest.approx(te_n, rel=1e-14, abs=0)


def test_envs_after_init():
    o, n = both()
    for j in range(3, o.N + 1):
        np.testing.assert_allclose(o.env(j), n.E[j], rtol=RTOL, atol=1e-300)
    # Label index appears exactly on right envs of sites <= c0
    for j in range(3, o.N + 1):
        assert (o.env(j).ndim == 3) == (j <= o.c0)


@pytest.mark.parametrize("b", [1, 2, 5, 6, 7, 11])
def test_forward_gradient_each_bond_kind(b):
    """bond kinds: edge (b=1,N-1), label on RE (b+1<c0), on B (c0 in {b,b+1}), on LE (b>c0)"""
    o, n = both()
    for bb in range(1, b):                                   # walk envs to bond b without optimising
        o.shiftE(bb, True)
        n.shiftE(bb, True)
    o.set_bond(b)
    n.set_bond(b)
    B = o.bond_tensor(b)
    np.testing.assert_allclose(B, n.bond_tensor(b), rtol=RTOL, atol=1e-300)
    rng = np.random.default_rng(b)
    B = B + 0.1 * rng.standard_normal(B.shape)
    np.testing.assert_allclose(o.forward(B), n.forward(B), rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(o.gradient(B), n.gradient(B), rtol=1e-9, atol=1e-12)
    Cc, lc, cr, nc = o.quadcost(B, 1e-3)
    Cn, ln, crn, ncn = n.quadcost(B, 1e-3)
    assert Cc == pytest.approx(Cn, rel=1e-12)
    np.testing.assert_allclose(lc, ln, rtol=1e-12)
    assert nc == ncn


@pytest.mark.parametrize("lam", [0.0, 1e-3])
def test_cgrad_matches_and_is_monotone(lam):
    o, n = both()
    B0 = o.bond_tensor(1)
    Bc, tc = o.cgrad(B0, 4, lam, 1e-10)
    Bn, tn = n.cgrad(B0, 4, lam, 1e-10)
    np.testing.assert_allclose(Bc, Bn, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(tc["cost"], tn["cost"], rtol=1e-10)
    np.testing.assert_allclose(tc["rnorm"], tn["rnorm"], rtol=1e-8)
    assert len(tc["cost"]) == 3                                    # last pass exits before re-evaluating (fixedL.cc:409)
    assert all(x >= y - 1e-12 * abs(x) for x, y in zip(tc["cost"], tc["cost"][1:]))


def test_full_sweep_reports_match():
    o, n = both(N=10, NT=40, m=4)
    rc = o.mldmrg(1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    rn = n.mldmrg(1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(rc) == len(rn) == 2 * (o.N - 1)
    for a, b in zip(rc, rn):
        assert (a["bond"], a["half"], a["newm"], a["origm"]) == (b["bond"], b["half"], b["newm"], b["origm"])
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-7)
        assert a["ncorrect"] == b["ncorrect"]
        assert a["truncerr"] == pytest.approx(b["truncerr"], rel=1e-5, abs=1e-18)
        assert a["diff"] == pytest.approx(b["diff"], rel=1e-5, abs=1e-12)
    # Label index stays on site N/2 (fixedL.cc:734)
    for j, A in enumerate(o.get_mps(), start=1):
        assert (A.ndim == 4) == (j == o.c0)


def test_svd_split_identities():
    """|B-newB|^2 = sum of discarded sigma^2; newm <= min(maxm, rows, cols)  """
    o, n = both(N=12, NT=60, m=6)
    for b, ha in [(1, 1), (5, 1), (6, 1)]:
        for bb in range(1, b):
            o.shiftE(bb, True)
        o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.3 * np.random.default_rng(b).standard_normal(B.shape)
        m, te, sv = o.svd_split(B, b, ha, 1e-10, 5, 2, )
        newB = o.bond_tensor(b)
        assert m <= 5
        assert np.sum((B - newB) ** 2) == pytest.approx(np.sum(sv[m:] ** 2), rel=1e-8)
        assert te == pytest.approx(np.sum(sv[m:] ** 2) / np.sum(sv ** 2), rel=1e-10)
        o = both(N=12, NT=60, m=6)[0]


def test_env_consistency_with_toverlap():
    """P_n from envs at any bond equals the full contraction of image n with W"""
    o, n = both()
    for b in range(1, o.N):
        o.set_bond(b)
        P = o.forward(o.bond_tensor(b))
        full = np.stack([o.toverlap(i) for i in range(o.NT)])
        np.testing.assert_allclose(P, full, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(full[0], n.toverlap(0), rtol=1e-9, atol=1e-13)
        if b < o.N - 1:
            o.shiftE(b, True)


def test_partition_independence():
    """results independent of (nthread, Nbatch) up to summation order"""
    o1, _ = both(NT=60, nthread=1, nbatch=1)
    o2, _ = both(NT=60, nthread=3, nbatch=4)
    B = o1.bond_tensor(1)
    np.testing.assert_allclose(o1.gradient(B), o2.gradient(B), rtol=1e-11, atol=1e-14)
    assert o1.quadcost(B, 1e-3)[0] == pytest.approx(o2.quadcost(B, 1e-3)[0], rel=1e-13)
    r1 = o1.mldmrg(1, 4, 2, 1e-10, 2, 1e-3, 1e-10, max_bonds=4)
    r2 = o2.mldmrg(1, 4, 2, 1e-10, 2, 1e-3, 1e-10, max_bonds=4)
    for a, b in zip(r1, r2):
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-9)


def test_nbatch_must_divide():
    pixels, labels, phi, W = make_problem(NT=60)
    with pytest.raises(ValueError, match="commensurate"):
        pyoracle.Oracle(phi, labels, W, nthread=1, nbatch=7)      # fixedL.cc:84-89
