"""GPU parity tests: the HIP path (through the C-ABI, tnml_amd.fixedl) against the CPU oracle on
identical seeded inputs.

dtype "f64" (TNML_F64, the default): everything in fp64 like the reference -- the tolerances are those of two
fp64 implementations with different summation orders.  dtype "f64_e32" (TNML_F64_E32): fp64 MFMA and fp64
CG/SVD algebra over fp32-STORED environments -- tolerances set by that storage rounding (~1e-7 per site,
accumulated along the chain and amplified by the CG).  dtype "f32": the looser study figures of SURVEY.md 8(d)."""
import os
import sys

import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu

TOL = {
    #            environments  P_n (rel max|P|)  gradient (rel max|G|)  costs      CG cost trace  CG alpha/|r|/B
    "f64":     dict(E=1e-12,   P=1e-11,          G=1e-9,                C=1e-11,   cgc=1e-9,      cga=1e-5),
    "f64_e32": dict(E=5e-6,    P=5e-6,           G=2e-5,                C=2e-6,    cgc=1e-5,      cga=1e-3),
}
BOTH = ["f64", "f64_e32"]


def _pair(N=12, NT=60, m=4, seed=3, boost=200.0, use_u8=False, maxm=None, dtype="f64"):
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    pixels, labels, phi, W = make_problem(N, NT, m, seed, pixel_boost=boost)
    if use_u8:
        ts = TrainStates(labels, N, maxm or m, pixels=pixels, dtype=dtype)
        phi = pyoracle.features_series(pixels)
    else:
        ts = TrainStates(labels, N, maxm or m, phi=phi, dtype=dtype)
    o = pyoracle.Oracle(phi, labels, W)
    ts.set_mps(W)
    o.init()
    ts.init()
    return ts, o


def _relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _walk(ts, o, b):
    for bb in range(1, b):
        ts.shiftE(bb, True)
        o.shiftE(bb, True)
    ts.setBond(b)
    o.set_bond(b)


@pytest.mark.parametrize("dtype", BOTH)
@pytest.mark.parametrize("use_u8", [False, True])
def test_envs_after_init(use_u8, dtype):
    ts, o = _pair(use_u8=use_u8, boost=1.0 if use_u8 else 200.0, dtype=dtype)
    for j in range(3, o.N + 1):
        Eg, Eo = ts.env(j), o.env(j)
        assert Eg.shape == Eo.shape
        assert _relmax(Eg, Eo) < TOL[dtype]["E"], f"site {j}"


@pytest.mark.parametrize("dtype", BOTH)
@pytest.mark.parametrize("b", [1, 2, 5, 6, 7, 11])
def test_forward_gradient_quadcost_each_bond_kind(b, dtype):
    ts, o = _pair(dtype=dtype)
    P_RTOL, G_RTOL, C_RTOL = TOL[dtype]["P"], TOL[dtype]["G"], TOL[dtype]["C"]
    _walk(ts, o, b)
    B = o.bond_tensor(b)
    assert _relmax(ts.bond_tensor(b), B) < 1e-12
    B = B + 0.1 * np.random.default_rng(b).standard_normal(B.shape)
    assert _relmax(ts.forward(B), o.forward(B)) < P_RTOL
    assert _relmax(ts.gradient(B), o.gradient(B)) < G_RTOL
    Cg, lg, crg, ng = ts.quadcost(B, 1e-3)
    Co, lo, cro, no = o.quadcost(B, 1e-3)
    assert Cg == pytest.approx(Co, rel=C_RTOL)
    np.testing.assert_allclose(lg, lo, rtol=5 * C_RTOL, atol=0.05 * C_RTOL * Co)
    assert crg == pytest.approx(cro, rel=1e-12)
    assert ng == no


@pytest.mark.parametrize("dtype", BOTH)
@pytest.mark.parametrize("b,lam", [(1, 0.0), (3, 1e-3), (6, 1e-3), (9, 1e-3)])
def test_cgrad_matches_oracle(b, lam, dtype):
    ts, o = _pair(dtype=dtype)
    _walk(ts, o, b)
    B0 = o.bond_tensor(b)
    Bg, tg = ts.cgrad(B0, 4, lam, 1e-10)
    Bo, to = o.cgrad(B0, 4, lam, 1e-10)
    assert tg["npass_done"] == to["npass_done"] == 4
    np.testing.assert_allclose(tg["cost"], to["cost"], rtol=TOL[dtype]["cgc"])
    np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=TOL[dtype]["cga"])
    np.testing.assert_allclose(tg["rnorm"], to["rnorm"], rtol=TOL[dtype]["cga"])
    assert _relmax(Bg, Bo) < TOL[dtype]["cga"]
    assert all(x >= y * (1 - 1e-6) for x, y in zip(tg["cost"], tg["cost"][1:]))      # CG monotone


@pytest.mark.parametrize("b,lam", [(1, 0.0), (3, 1e-3), (6, 1e-3), (9, 1e-3)])
def test_merged_cg_passes_match_the_oracle(b, lam):
    """option merged_cg = 2: the CG with ONE image sum per pass -- A p = sum_n (p.v_n) v_n formed from the pAp pass's outputs,
    residual r <- r - a (A p + lambda p), cost partials of a pass reported with the next one (what a multi-rank run uses to
    halve its all-reduces) -- against the oracle's literal cgrad (fixedL.cc:349-445) on every bond kind"""
    ts, o = _pair()
    ts.set_option("merged_cg", 2)
    _walk(ts, o, b)
    B0 = o.bond_tensor(b)
    Bg, tg = ts.cgrad(B0, 4, lam, 1e-10)
    Bo, to = o.cgrad(B0, 4, lam, 1e-10)
    assert tg["npass_done"] == to["npass_done"] == 4
    np.testing.assert_allclose(tg["cost"], to["cost"], rtol=TOL["f64"]["cgc"])
    np.testing.assert_allclose(tg["pAp"], to["pAp"], rtol=TOL["f64"]["cga"])
    np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=TOL["f64"]["cga"])
    np.testing.assert_allclose(tg["rnorm"], to["rnorm"], rtol=TOL["f64"]["cga"])
    assert _relmax(Bg, Bo) < TOL["f64"]["cga"]


@pytest.mark.parametrize("b", [2, 6, 9])
def test_pAp_entry_point(b):
    """tnml_pAp: sum_n |p*t.v_n|^2 + lambda |p|^2 (fixedL.cc:394-403) for a random direction, against the oracle's forward map"""
    ts, o = _pair()
    _walk(ts, o, b)
    p = np.random.default_rng(b).standard_normal(o.bond_shape(b))
    ref = float(np.sum(o.forward(p) ** 2) + 1e-3 * np.sum(p ** 2))
    assert ts.pAp(p, 1e-3) == pytest.approx(ref, rel=1e-11)


@pytest.mark.parametrize("b,ha", [(1, 1), (3, 1), (3, 2), (5, 1), (5, 2), (6, 1), (6, 2), (8, 2), (11, 2)])
def test_svd_split_matches_oracle(b, ha):
    ts, o = _pair(N=12, NT=60, m=6)
    _walk(ts, o, b)
    B = o.bond_tensor(b)
    B = B + 0.3 * np.random.default_rng(10 * b + ha).standard_normal(B.shape)
    mg, teg, svg = ts.svd_split(B, b, ha, 1e-10, 5, 2)
    mo, teo, svo = o.svd_split(B, b, ha, 1e-10, 5, 2)
    assert mg == mo
    np.testing.assert_allclose(svg, svo, rtol=1e-8, atol=1e-10 * svo[0])
    assert teg == pytest.approx(teo, rel=1e-6, abs=1e-18)
    # gauge invariant: the re-formed bond tensor, and isometry of the site the sweep leaves
    assert _relmax(ts.bond_tensor(b), o.bond_tensor(b)) < 1e-9
    A = ts.get_site(b if ha == 1 else b + 1)
    if ha == 1:
        M = np.moveaxis(A, 2, -1).reshape(-1, A.shape[2])
    else:
        M = A.reshape(A.shape[0], -1).T
    np.testing.assert_allclose(M.T @ M, np.eye(M.shape[1]), atol=1e-9)


@pytest.mark.parametrize("dtype,sweep_rtol", [("f64", 1e-8), ("f64_e32", 1e-4)])
def test_full_sweep_reports_match_oracle(dtype, sweep_rtol):
    ts, o = _pair(N=10, NT=40, m=4, dtype=dtype)
    from tnml_amd.fixedl import mldmrg
    rg = mldmrg(ts, 1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    ro = o.mldmrg(1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(rg) == len(ro) == 2 * (o.N - 1)
    for a, b in zip(rg, ro):
        assert (a["bond"], a["half"], a["origm"], a["newm"]) == (b["bond"], b["half"], b["origm"], b["newm"])
        assert a["cost"] == pytest.approx(b["cost"], rel=sweep_rtol)
        assert abs(a["ncorrect"] - b["ncorrect"]) <= (0 if dtype == "f64" else 1)
        np.testing.assert_allclose(a["label_cost"], b["label_cost"], rtol=10 * sweep_rtol, atol=sweep_rtol * b["cost"])
    # first bond: identical (W, data) state -> tight tolerance
    assert rg[0]["cost"] == pytest.approx(ro[0]["cost"], rel=TOL[dtype]["cgc"])
    assert rg[0]["truncerr"] == pytest.approx(ro[0]["truncerr"], rel=1e-3, abs=1e-12)
    for j, A in enumerate(ts.get_mps(), start=1):
        assert (A.ndim == 4) == (j == ts.c0)


@pytest.mark.parametrize("dtype", BOTH)
def test_m120_shapes_forward_gradient(dtype):
    """the maxm=120 specialisations (240-column feature GEMM, 80x80 gradient tiles) vs the oracle"""
    ts, o = _pair(N=20, NT=64, m=120, maxm=120, dtype=dtype)
    P_RTOL, G_RTOL, E_RTOL = TOL[dtype]["P"], TOL[dtype]["G"], TOL[dtype]["E"]
    _walk(ts, o, 8)          # bond 8: 120 x 120, Label on RE (c0 = 10)
    B = o.bond_tensor(8)
    B = B + 0.05 * np.random.default_rng(1).standard_normal(B.shape)
    assert B.shape == (120, 2, 2, 120)
    assert _relmax(ts.forward(B), o.forward(B)) < P_RTOL
    assert _relmax(ts.gradient(B), o.gradient(B)) < G_RTOL
    ts.shiftE(8, True); o.shiftE(8, True)
    ts.setBond(9); o.set_bond(9)     # bond 9: Label on B
    B = o.bond_tensor(9)
    assert B.shape == (120, 2, 2, 120, 10)
    assert _relmax(ts.forward(B), o.forward(B)) < P_RTOL
    assert _relmax(ts.gradient(B), o.gradient(B)) < G_RTOL
    for bb in (9, 10, 11):
        ts.shiftE(bb, True); o.shiftE(bb, True)
    ts.setBond(12); o.set_bond(12)   # bond 12: Label on LE
    B = o.bond_tensor(12)
    assert _relmax(ts.forward(B), o.forward(B)) < P_RTOL
    assert _relmax(ts.gradient(B), o.gradient(B)) < G_RTOL
    assert _relmax(ts.env(11), o.env(11)) < E_RTOL


@pytest.mark.parametrize("dtype", BOTH)
@pytest.mark.parametrize("NT", [1, 37, 257])
def test_ragged_image_count_and_padding(dtype, NT):
    """NT not a multiple of any tile size (and a single image): padding images must not contribute"""
    ts, o = _pair(N=8, NT=NT, m=3, dtype=dtype)
    B = o.bond_tensor(1)
    assert _relmax(ts.gradient(B), o.gradient(B)) < TOL[dtype]["G"]
    assert ts.quadcost(B, 0.0)[3] == o.quadcost(B, 0.0)[3]


def test_properties_at_scale():
    """size-independent properties at a BASELINE-config-2-like shape (N=784 is exercised by bench.py;
    here N=64, NT=10000, m=20): linearity of the forward map in B, gradient additivity over image
    shards, cost independent of the bond it is evaluated at (4-2/4-5)."""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, NT, m = 64, 10000, 20
    labels = synth.synthetic_labels(NT, per_label=NT // 10)
    pixels = synth.synthetic_images(N, labels)
    W = synth.random_mps(N, m, seed=5)
    ts = TrainStates(labels, N, m, pixels=pixels)
    ts.set_mps(W)
    ts.init()
    B1 = ts.bond_tensor(1)
    rng = np.random.default_rng(0)
    B2 = rng.standard_normal(B1.shape)
    P1, P2, P12 = ts.forward(B1), ts.forward(B2), ts.forward(B1 + 2.0 * B2)
    assert _relmax(P12, P1 + 2.0 * P2) < 1e-4
    G = ts.gradient(B1)
    half = NT // 2
    parts = []
    for sl in (slice(0, half), slice(half, NT)):
        t2 = TrainStates(labels[sl], N, m, pixels=pixels[sl])
        t2.set_mps(W)
        t2.init()
        parts.append(t2.gradient(B1))
        t2.close()
    assert _relmax(parts[0] + parts[1], G) < 1e-4
    c1 = ts.quadcost(B1, 0.0)[0]
    for b in range(1, 6):
        ts.shiftE(b, True)
    ts.setBond(6)
    c6 = ts.quadcost(ts.bond_tensor(6), 0.0)[0]
    assert c6 == pytest.approx(c1, rel=1e-4)


def test_f32_study_mode_forward_gradient():
    """TNML_F32 (exact-fp32 MFMA): per-image contractions agree to fp32 round-off; its CG is NOT
    expected to track the fp64 reference (DESIGN.md "why fp64 MFMA"), so only single evaluations
    are checked, with the looser tolerances of SURVEY.md 8(d)."""
    ts, o = _pair(dtype="f32")
    for b in (1, 6, 9):
        if b > 1:
            for bb in range(ts._walked if hasattr(ts, "_walked") else 1, b):
                ts.shiftE(bb, True); o.shiftE(bb, True)
        ts._walked = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b) + 0.1 * np.random.default_rng(b).standard_normal(o.bond_shape(b))
        assert _relmax(ts.forward(B), o.forward(B)) < 1e-4
        assert _relmax(ts.gradient(B), o.gradient(B)) < 2e-4
        assert ts.quadcost(B, 1e-3)[0] == pytest.approx(o.quadcost(B, 1e-3)[0], rel=1e-5)


@pytest.mark.parametrize("ha", [1, 2])
@pytest.mark.parametrize("backend", ["0", "1", "2"])
def test_svd_split_240x240_against_lapack(ha, backend, monkeypatch):
    """the n=240 split (in-house tridiagonalisation + dstedc + back transform, and the stock rocSOLVER
    path) against numpy/LAPACK: singular values, truncation error, optimal rank-120 reconstruction,
    isometry of the site the sweep leaves"""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    monkeypatch.setenv("TNML_SVD_BACKEND", backend)
    N, m, NT = 20, 120, 16
    labels = synth.synthetic_labels(NT, seed=1)
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    W = synth.random_mps(N, m, seed=2)
    ts.set_mps(W)
    b = 8
    rng = np.random.default_rng(ha)
    decay = np.exp(-0.12 * np.arange(240))
    U0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    V0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    M = (U0 * decay) @ V0.T                                   # rows (a,s), cols (t,beta)
    B = M.reshape(120, 2, 2, 120, order="F")
    mg, te, sv = ts.svd_split(B, b, ha, 0.0, 120, 60)       # cutoff 0: maxm decides
    s_ref = np.linalg.svd(M, compute_uv=False)
    assert mg == 120
    # Gram route: sqrt(eps) floor, as in ITensor.  The tridiagonalisation drops a trailing block whose eigenvalues sum to less
    # than 1e-15 trace(G) (here 4.7e-15 sigma_0^2): singular values below 7e-8 sigma_0 may come out as zero.
    np.testing.assert_allclose(sv, s_ref, rtol=1e-7, atol=1e-7 * s_ref[0])
    assert te == pytest.approx(np.sum(s_ref[120:] ** 2) / np.sum(s_ref ** 2), rel=1e-5)
    newB = ts.bond_tensor(b).reshape(240, 240, order="F")
    Ur, sr, Vr = np.linalg.svd(M)
    best = (Ur[:, :120] * sr[:120]) @ Vr[:120]
    assert np.abs(newB - best).max() < 1e-9 * sr[0]
    A = ts.get_site(b if ha == 1 else b + 1)
    Q = A.reshape(240, 120, order="F") if ha == 1 else A.reshape(120, 240, order="F").T
    np.testing.assert_allclose(Q.T @ Q, np.eye(120), atol=1e-10)
    st = ts.svd_stats()
    assert st["fallbacks"] == 0, st


def test_cgrad_early_exit_on_cconv():
    """|r| < cconv (fixedL.cc:432-436) is a device-side flag here: later passes must be no-ops"""
    ts, o = _pair()
    B0 = o.bond_tensor(1)
    _, t_ref = o.cgrad(B0, 4, 1e-3, 1e-10)
    cconv = 0.5 * (t_ref["rnorm"][0] + t_ref["rnorm"][1])            # trips at the second pass
    assert t_ref["rnorm"][1] < cconv < t_ref["rnorm"][0]
    Bg, tg = ts.cgrad(B0, 4, 1e-3, cconv)
    Bo, to = o.cgrad(B0, 4, 1e-3, cconv)
    assert to["converged"] and tg["converged"]
    assert tg["npass_done"] == to["npass_done"] == 2
    np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-9)
    assert _relmax(Bg, Bo) < 1e-6


def test_fixedl_cli_driver_end_to_end(tmp_path):
    """the C++ `fixedL <inputfile>` driver (tnml_amd/host) on a tiny idx dataset: its per-bond
    "--> After SVD, Cost" log lines against the oracle started from the same initial W"""
    import os
    import re
    import subprocess
    from oracle import pyoracle
    from tnml_amd import hostlib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, per_label = 16, 20
    labels = synth.synthetic_labels(10 * per_label, seed=9, per_label=per_label)
    pixels = synth.synthetic_images(N, labels, seed=9)
    pixels = np.clip(pixels.astype(np.int32) * 3, 0, 255).astype(np.uint8)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    inp = tmp_path / "input"
    inp.write_text("input\n{\ndatadir = %s\nNtrain = %d\nNbatch = 4\nNsweep = 1\ncutoff = 1E-10\nmaxm = 6\nminm = 3\n"
                   "ninitial = 3\nlambda = 1E-3\nNpass = 3\nseed = 5\nprecision = f64\n}\n" % (data, per_label))
    run = subprocess.run([os.path.join(root, "tnml_amd", "fixedL"), str(inp)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    log = run.stdout
    assert "Total of %d training images" % (10 * per_label) in log and "%d sites of dimension 2" % N in log
    costs = [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", log)]
    newm = [int(x) for x in re.findall(r"New m=(\d+)", log)]
    assert len(costs) == 2 * (N - 1)
    assert os.path.exists(tmp_path / "W") and os.path.exists(tmp_path / "sites")
    # oracle from the identical initial W (same builder, same seed) and the identical selection of images
    w0 = str(tmp_path / "W0ref")
    hostlib.build_initial_w(data, per_label, 3, 5, w0)
    px, lab, _ = hostlib.read_mnist(data, True, per_label)
    o = pyoracle.Oracle(pyoracle.features_series(px), lab, hostlib.read_mps(w0))
    o.init()
    ro = o.mldmrg(1, 6, 3, 1e-10, 3, 1e-3, 1e-10)
    # strict precision (fp64 environments and features).  With the reference's own feature map every image is
    # almost the same vector and the CG is so ill conditioned that two fp64 implementations with different
    # summation orders agree only to ~1e-5 from the first Label-on-B bond on (alpha differs at 4e-5 there,
    # DESIGN.md section 2); bond dimensions can then differ by one where the tail of the spectrum is noise.
    ref_m = [r["newm"] for r in ro]
    assert newm[:6] == ref_m[:6] and all(abs(a - b) <= 1 for a, b in zip(newm, ref_m))
    ref_c = np.array([r["cost"] / len(lab) for r in ro])
    np.testing.assert_allclose(costs[:6], ref_c[:6], rtol=1e-9)              # before the first Label-on-B bond
    np.testing.assert_allclose(costs, ref_c, rtol=2e-5)
    # the final W on disk carries the Label index on site N/2 only
    Wf = hostlib.read_mps(str(tmp_path / "W"))
    assert [A.ndim == 4 for A in Wf] == [j == N // 2 for j in range(1, N + 1)]
    # `fulltest <inputfile>` (fulltest.cc) on a held-out idx test set with the W just written
    tl = synth.synthetic_labels(130, seed=21)
    tp = np.clip(synth.synthetic_images(N, tl, seed=21).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    synth.write_idx(data, tp, tl, train=False)
    for feat, phi_t in (("series", pyoracle.features_series(tp)),
                        ("normal", np.stack([np.cos(np.pi / 2 * tp / 65025.), np.sin(np.pi / 2 * tp / 65025.)], axis=-1))):
        tin = tmp_path / ("input_test_" + feat)
        tin.write_text("input\n{\ndatadir = %s\nfname = W\nfeature = %s\nprecision = f64\n}\n" % (data, feat))
        run = subprocess.run([os.path.join(root, "tnml_amd", "fulltest"), str(tin)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
        ot = pyoracle.Oracle(phi_t, tl, Wf)
        Wt = np.stack([ot.toverlap(i) for i in range(len(tl))])
        pred = np.abs(Wt).argmax(axis=1)
        ncor = int((pred == tl).sum())
        m = re.search(r"(\d+)/(\d+) correct \(([0-9.]+)%\), (\d+)/(\d+) incorrect", run.stdout)
        assert m and int(m.group(2)) == len(tl) and int(m.group(1)) == ncor and int(m.group(4)) == len(tl) - ncor
        digits = re.findall(r"Digit (\d) (\d+)/(\d+) correct", run.stdout)
        assert [(int(a), int(b), int(c)) for a, b, c in digits] == \
            [(l, int(((pred == tl) & (tl == l)).sum()), int((tl == l).sum())) for l in range(10) if (tl == l).any()]
        assert "Total # test images = %d" % len(tl) in run.stdout


def test_fixedl_cli_resumes_from_the_W_it_wrote(tmp_path):
    """checkpoint / resume (fixedL.cc:671-681, :565): a run that finds `W` and `sites` in its directory continues from them.  One run of
    two sweeps against two runs of one sweep each in the same directory: the second run reads the first one's W, starts at the cost the
    first one ended with and reproduces the second sweep of the long run (W is stored in fp64, the environments are rebuilt by the same
    kernels)."""
    import re
    import shutil
    import subprocess
    from tnml_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, per_label = 16, 8
    labels = synth.synthetic_labels(10 * per_label, seed=9, per_label=per_label)
    pixels = np.clip(synth.synthetic_images(N, labels, seed=9).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)

    def run(wd, nsweep):
        wd.mkdir(exist_ok=True)
        (wd / "input").write_text("input\n{\ndatadir = %s\nfeature_scale = 255\nNtrain = %d\nNbatch = 4\nNsweep = %d\ncutoff = 1E-10\nmaxm = 6\n"
                                  "minm = 3\nninitial = 2\nlambda = 1E-3\nNpass = 2\nseed = 5\nbond_log = bonds.csv\n}\n" % (data, per_label, nsweep))
        r = subprocess.run([os.path.join(root, "tnml_amd", "fixedL"), str(wd / "input")], capture_output=True, text=True, cwd=wd, timeout=300)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        return r.stdout, [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", r.stdout)]
    log_a, cost_a = run(tmp_path / "long", 2)
    log_b1, cost_b1 = run(tmp_path / "short", 1)
    assert "Reading W from disk" not in log_b1
    log_b2, cost_b2 = run(tmp_path / "short", 1)
    assert "Reading W from disk" in log_b2 and "Done making initial W" not in log_b2
    nb = 2 * (N - 1)
    assert len(cost_a) == 2 * nb and len(cost_b1) == nb and len(cost_b2) == nb
    np.testing.assert_allclose(cost_b1, cost_a[:nb], rtol=1e-12)
    start = float(re.search(r"Before starting DMRG Cost = ([0-9.eE+-]+)", log_b2).group(1))
    assert start == pytest.approx(cost_b1[-1], rel=1e-8)
    np.testing.assert_allclose(cost_b2, cost_a[nb:], rtol=1e-8)
    # the machine-readable log (extension key bond_log): one CSV line per bond update with the numbers of the text log
    import csv
    rows = list(csv.DictReader(open(tmp_path / "long" / "bonds.csv")))
    assert len(rows) == 2 * nb and [int(r["sweep"]) for r in rows] == [1] * nb + [2] * nb
    np.testing.assert_allclose([float(r["cost_after_svd"]) for r in rows], cost_a, rtol=1e-9)
    assert [int(r["new_m"]) for r in rows] == [int(x) for x in re.findall(r"New m=(\d+)", log_a)]
    assert all(float(r["seconds"]) >= 0. and int(r["ntrain"]) == 10 * per_label for r in rows)


@pytest.mark.parametrize("pipeline", ["yes", "no"])
def test_fixedl_cli_file_hooks(tmp_path, pipeline):
    """the WRITE_WF / LAMBDA hooks of mldmrg (fixedL.cc:542-559): files dropped in the working directory are noticed after a bond
    update, removed, W is written, the new lambda takes effect for the following bond updates -- with the pipelined sweep loop
    (one bond later than in the reference, by construction) and with `pipeline = no`"""
    import re
    import subprocess
    from tnml_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, per_label = 16, 6
    labels = synth.synthetic_labels(10 * per_label, seed=9, per_label=per_label)
    pixels = np.clip(synth.synthetic_images(N, labels, seed=9).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    inp = tmp_path / "input"
    inp.write_text("input\n{\ndatadir = %s\nfeature_scale = 255\nNtrain = %d\nNbatch = 3\nNsweep = 1\ncutoff = 1E-10\nmaxm = 5\nminm = 2\n"
                   "ninitial = 2\nlambda = 1E-3\nNpass = 2\nseed = 5\npipeline = %s\n}\n" % (data, per_label, pipeline))
    (tmp_path / "WRITE_WF").write_text("")
    (tmp_path / "LAMBDA").write_text("0.02\n")
    run = subprocess.run([os.path.join(root, "tnml_amd", "fixedL"), str(inp)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-2000:]
    log = run.stdout
    assert log.count("File WRITE_WF found") == 1 and log.count("new lambda = 0.02") == 1
    assert not (tmp_path / "WRITE_WF").exists() and not (tmp_path / "LAMBDA").exists()
    assert (tmp_path / "W").exists()
    lams = [float(x) for x in re.findall(r"In cgrad, lambda = ([0-9.eE+-]+)", log)]
    assert len(lams) == 2 * (N - 1)
    first_new = lams.index(2e-2)
    assert lams[:first_new] == [1e-3] * first_new and lams[first_new:] == [2e-2] * (len(lams) - first_new)
    assert first_new == (1 if pipeline == "no" else 2)        # strict order: the second bond update; pipelined: the third (its predecessor was already issued)


@pytest.mark.parametrize("precision,rtol", [("f64", 1e-6), ("mixed", 2e-2)])
def test_fixedl_cli_imglen_and_feature_scale(tmp_path, precision, rtol):
    """driver extensions: `imglen` (8x8 -> 4x4 block means) and `feature_scale = 255` (the README's [1, x/4] map, well
    conditioned) -- log costs and the fulltest table against the oracle fed with the same reduced images"""
    import os
    import re
    import subprocess
    from oracle import pyoracle
    from tnml_amd import hostlib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    per_label = 20
    labels = synth.synthetic_labels(10 * per_label, seed=4, per_label=per_label)
    tl = synth.synthetic_labels(90, seed=22)
    allpx = synth.synthetic_images(64, np.concatenate([labels, tl]), seed=4)     # one seed = one set of class templates
    pixels, tp = allpx[:len(labels)], allpx[len(labels):]
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    synth.write_idx(data, tp, tl, train=False)
    inp = tmp_path / "input"
    keys = "datadir = %s\nimglen = 4\nfeature_scale = 255\nprecision = %s\n" % (data, precision)
    inp.write_text("input\n{\n%sNtrain = %d\nNbatch = 4\nNsweep = 2\ncutoff = 1E-10\nmaxm = 6\nminm = 3\nninitial = 3\n"
                   "lambda = 1E-3\nNpass = 3\nseed = 5\n}\n" % (keys, per_label))
    run = subprocess.run([os.path.join(root, "tnml_amd", "fixedL"), str(inp)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-2000:]
    assert "16 sites of dimension 2" in run.stdout
    costs = [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", run.stdout)]
    newm = [int(x) for x in re.findall(r"New m=(\d+)", run.stdout)]

    def feats(px):
        g = hostlib.reduce(px, 8, 4) / 255.
        return np.stack([np.ones_like(g), 255. * ((g / 255.) / 4.)], axis=-1)
    w0 = str(tmp_path / "W0ref")
    hostlib.build_initial_w(data, per_label, 3, 5, w0, imglen=4, feature_scale=255.)
    px, lab, _ = hostlib.read_mnist(data, True, per_label)
    o = pyoracle.Oracle(feats(px), lab, hostlib.read_mps(w0))
    o.init()
    ro = o.mldmrg(2, 6, 3, 1e-10, 3, 1e-3, 1e-10)
    assert len(costs) == len(ro) == 4 * 15
    np.testing.assert_allclose(costs, [r["cost"] / len(lab) for r in ro], rtol=rtol, atol=2e-10)   # atol: 10 printed digits
    if precision == "f64":
        assert newm == [r["newm"] for r in ro]
    # evaluator with the same keys
    tin = tmp_path / "input_test"
    tin.write_text("input\n{\n%s}\n" % keys)
    run = subprocess.run([os.path.join(root, "tnml_amd", "fulltest"), str(tin)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    Wf = hostlib.read_mps(str(tmp_path / "W"))
    ot = pyoracle.Oracle(feats(tp), tl, Wf)
    Wt = np.stack([ot.toverlap(i) for i in range(len(tl))])
    ncor = int((np.abs(Wt).argmax(axis=1) == tl).sum())
    m = re.search(r"(\d+)/(\d+) correct", run.stdout)
    assert m and int(m.group(2)) == len(tl) and abs(int(m.group(1)) - ncor) <= (0 if precision == "f64" else 1)
    assert ncor > 45                                           # the model has learned something on this toy set


@pytest.mark.parametrize("b", [1, 3, 6, 9])
def test_reference_feature_map_single_bond(b):
    """TNML_F64 with the reference's own feature map [1, byte/260100] (raw bytes through
    tnml_set_data_u8): environments, P, gradient, CG trace to ~1e-9"""
    ts, o = _pair(use_u8=True, boost=1.0)
    for j in range(3, o.N + 1):
        assert _relmax(ts.env(j), o.env(j)) < 1e-12
    _walk(ts, o, b)
    B0 = o.bond_tensor(b)
    assert _relmax(ts.forward(B0), o.forward(B0)) < 1e-11
    assert _relmax(ts.gradient(B0), o.gradient(B0)) < 1e-9
    Bg, tg = ts.cgrad(B0, 4, 1e-3, 1e-10)
    Bo, to = o.cgrad(B0, 4, 1e-3, 1e-10)
    np.testing.assert_allclose(tg["cost"], to["cost"], rtol=1e-10)
    np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=1e-3)      # alpha ~ 1e3 on the Label-on-B bond: ill conditioned
    assert _relmax(Bg, Bo) < 1e-3


def test_reference_feature_map_full_sweep():
    ts, o = _pair(N=10, NT=40, m=4, use_u8=True, boost=1.0)
    from tnml_amd.fixedl import mldmrg
    rg = mldmrg(ts, 1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    ro = o.mldmrg(1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert [r["newm"] for r in rg] == [r["newm"] for r in ro]
    np.testing.assert_allclose([r["cost"] for r in rg], [r["cost"] for r in ro], rtol=1e-8)
    assert [r["ncorrect"] for r in rg] == [r["ncorrect"] for r in ro]


@pytest.mark.parametrize("dtype,tol", [("f64_e32", 2e-6), ("f64", 1e-12), ("f32", 2e-5)])
@pytest.mark.parametrize("N,m", [(12, 4), (4, 2), (17, 6)])
def test_classify_matches_toverlap(dtype, tol, N, m):
    """tnml_classify (fulltest.cc / util.h toverlap + fullTest) against the oracle's per-image full contraction"""
    ts, o = _pair(N=N, NT=70, m=m, dtype=dtype)
    W_or = np.stack([o.toverlap(i) for i in range(o.NT)])
    w, pred, cnt, ninc = ts.classify()
    assert _relmax(w, W_or) < tol
    pred_or = np.abs(W_or).argmax(axis=1)                      # numpy argmax = first maximum, util.h:42-57
    labels = np.asarray(o.labels)
    if dtype == "f64":
        np.testing.assert_array_equal(pred, pred_or)
    else:                                                      # a fp32-rounded near-tie may flip
        gap = np.sort(np.abs(W_or), axis=1)
        clear = (gap[:, -1] - gap[:, -2]) > 10 * tol * np.abs(W_or).max()
        np.testing.assert_array_equal(pred[clear], pred_or[clear])
    np.testing.assert_array_equal(cnt, np.bincount(labels, minlength=10))
    np.testing.assert_array_equal(ninc, np.bincount(labels[pred != labels], minlength=10))
    # the training environments survive a classify call: a bond update afterwards still matches the oracle
    if dtype == "f64":
        ts.setBond(1)
        o.set_bond(1)
        B0 = o.bond_tensor(1)
        assert _relmax(ts.forward(B0), o.forward(B0)) < 1e-11


def test_classify_after_training_agrees_with_quadcost_count():
    """after a sweep, the number of correct training images from the full contraction equals quadcost's count
    at any bond (both are argmax_l |W_l| of the same network)"""
    ts, o = _pair(N=10, NT=80, m=4)
    from tnml_amd.fixedl import mldmrg
    rg = mldmrg(ts, 1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    w, pred, cnt, ninc = ts.classify()
    assert int(cnt.sum() - ninc.sum()) == rg[-1]["ncorrect"]


@pytest.mark.parametrize("rank,noise", [(130, 0.0), (130, 1e-9), (60, 1e-13), (236, 0.0)])
def test_rank_adaptive_exit_of_the_tridiagonalisation(rank, noise):
    """the split's Householder chain stops when the trailing block of the Gram matrix is numerically zero (eigh.hip,
    k_sytrd_v3 psd_tol): same singular values, truncation error, reconstruction and isometry as the full chain
    (option sytrd_exit = 0) and as LAPACK, for the shape a sweep produces (rank m + a few of n = 2m), with a noise
    floor above and below the threshold, and for a nearly full-rank tensor where the exit must not fire early"""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, m, NT = 20, 120, 16
    labels = synth.synthetic_labels(NT, seed=1)
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    ts.set_mps(synth.random_mps(N, m, seed=2))
    rng = np.random.default_rng(rank)
    U0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    V0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    sv0 = np.zeros(240)
    sv0[:rank] = np.exp(-0.03 * np.arange(rank))
    sv0[rank:] = noise * rng.uniform(0.5, 1.0, 240 - rank)
    M = (U0 * sv0) @ V0.T
    B = M.reshape(120, 2, 2, 120, order="F")
    s_ref = np.linalg.svd(M, compute_uv=False)
    Ur, sr, Vr = np.linalg.svd(M)
    best = (Ur[:, :120] * sr[:120]) @ Vr[:120]
    te_ref = np.sum(s_ref[120:] ** 2) / np.sum(s_ref ** 2)
    out = {}
    for ex in (0, 1):
        ts.set_option("sytrd_exit", ex)
        for ha in (1, 2):
            mg, te, sv = ts.svd_split(B, 8, ha, 0.0, 120, 60)        # cutoff 0: maxm decides
            assert mg == 120
            np.testing.assert_allclose(sv, s_ref, rtol=1e-7, atol=1.5e-7 * s_ref[0])
            assert abs(te - te_ref) < 1e-12 + 1e-6 * te_ref
            newB = ts.bond_tensor(8).reshape(240, 240, order="F")
            assert np.abs(newB - best).max() < 2e-7 * sr[0]           # ties at the cut (sr[119] ~ sr[120]) leave this much freedom
            assert abs(np.linalg.norm(newB - M) - np.linalg.norm(best - M)) < 1e-9 * sr[0]   # ... but the truncation is optimal
            A = ts.get_site(8 if ha == 1 else 9)
            Q = A.reshape(240, 120, order="F") if ha == 1 else A.reshape(120, 240, order="F").T
            np.testing.assert_allclose(Q.T @ Q, np.eye(120), atol=1e-10)
            out[(ex, ha)] = (te, sv)
    for ha in (1, 2):
        np.testing.assert_allclose(out[(1, ha)][1], out[(0, ha)][1], rtol=1e-9, atol=1.5e-7 * s_ref[0])
    ts.set_option("sytrd_exit", 1)
    assert ts.svd_stats()["fallbacks"] == 0


@pytest.mark.parametrize("rank", [30, 119])
def test_svd_split_rank_deficient_cluster(rank):
    """a bond tensor of numerical rank < minm: the kept basis contains a cluster of numerically zero eigenvalues.
    The in-house eigensolver must still return an isometry (Cholesky-QR repair of the cluster, no rocSOLVER
    fallback) and the optimal reconstruction."""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, m, NT = 20, 120, 16
    labels = synth.synthetic_labels(NT, seed=1)
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    ts.set_mps(synth.random_mps(N, m, seed=2))
    rng = np.random.default_rng(rank)
    U0, _ = np.linalg.qr(rng.standard_normal((240, rank)))
    V0, _ = np.linalg.qr(rng.standard_normal((240, rank)))
    sv0 = np.exp(-0.05 * np.arange(rank))
    M = (U0 * sv0) @ V0.T
    B = M.reshape(120, 2, 2, 120, order="F")
    for ha in (1, 2):
        mg, te, sv = ts.svd_split(B, 8, ha, 1e-10, 120, 120)        # minm = maxm: nothing may be dropped
        assert mg == 120
        np.testing.assert_allclose(sv[:rank], sv0, rtol=1e-7, atol=1e-7)
        assert np.all(sv[rank:] < 1e-6)
        newB = ts.bond_tensor(8).reshape(240, 240, order="F")
        assert np.abs(newB - M).max() < 1e-9
        A = ts.get_site(8 if ha == 1 else 9)
        Q = A.reshape(240, 120, order="F") if ha == 1 else A.reshape(120, 240, order="F").T
        np.testing.assert_allclose((Q.T @ Q)[:rank, :rank], np.eye(rank), atol=1e-10)
        if ha == 1:                                                   # the Gram-side factor is a full isometry
            np.testing.assert_allclose(Q.T @ Q, np.eye(120), atol=1e-10)
    st = ts.svd_stats()
    assert st["fallbacks"] == 0, st


def test_svd_split_graded_spectrum_with_close_pairs():
    """a graded spectrum (13 decades inside the kept 120) with pairs of eigenvalues far closer than eps*lambda_max --
    what a trained bond looks like.  The eigensolver splits the tridiagonal matrix into unreduced blocks and repairs
    what inverse iteration leaves non-orthogonal; no rocSOLVER fallback, full isometry."""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, m, NT = 20, 120, 16
    labels = synth.synthetic_labels(NT, seed=1)
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    ts.set_mps(synth.random_mps(N, m, seed=2))
    rng = np.random.default_rng(11)
    sv0 = np.concatenate([10.0 ** (-np.arange(16) * 0.45), 1e-7 * (1 + 0.01 * np.arange(104) // 2), 1e-12 * rng.random(120)])
    sv0 = np.sort(sv0)[::-1]
    U0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    V0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    M = (U0 * sv0) @ V0.T
    B = M.reshape(120, 2, 2, 120, order="F")
    for ha in (1, 2):
        mg, te, sv = ts.svd_split(B, 8, ha, 0.0, 120, 120)
        assert mg == 120
        np.testing.assert_allclose(sv[:16] ** 2, sv0[:16] ** 2, rtol=1e-7, atol=1e-15)    # Gram route: eps*lambda_max floor
        newB = ts.bond_tensor(8).reshape(240, 240, order="F")
        assert np.abs(newB - M).max() < 1e-7
        A = ts.get_site(8 if ha == 1 else 9)
        Q = A.reshape(240, 120, order="F") if ha == 1 else A.reshape(120, 240, order="F").T
        if ha == 1:
            np.testing.assert_allclose(Q.T @ Q, np.eye(120), atol=1e-9)
    st = ts.svd_stats()
    assert st["fallbacks"] == 0, st


# ---------------------------------------------------------------------------------------------------
# per-label variant (single.cc / single.h): TNML_MODE_SINGLE against oracle/single_oracle.c
def _single_pair(N=12, NT=60, m=4, target=3, seed=3, boost=300.0, normal=True, dtype="f64", maxm=None):
    from oracle import pyoracle
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    labels = synth.synthetic_labels(NT, seed=seed, per_label=NT // 10)
    pixels = synth.synthetic_images(N, labels, seed=seed)
    phi = pyoracle.features_single(pixels, normal).copy()
    phi[..., 1] *= boost
    W = synth.random_mps(N, m, seed=seed + 7)
    W[N // 2 - 1] = W[N // 2 - 1][..., 0] * 3.0            # plain MPS: no Label index
    ts = TrainStates(labels, N, maxm or m, phi=phi, dtype=dtype, single_label=target)
    o = pyoracle.SingleOracle(phi, labels, target, W)
    ts.set_mps(W)
    o.init()
    ts.init()
    return ts, o


@pytest.mark.parametrize("dtype", BOTH)
@pytest.mark.parametrize("b", [1, 2, 6, 11])
def test_single_forward_gradient_cost_cgrad(b, dtype):
    ts, o = _single_pair(dtype=dtype)
    T = TOL[dtype]
    for j in range(3, o.N + 1):
        assert _relmax(ts.env(j), o.env(j)) < T["E"]
    for bb in range(1, b):
        ts.shiftE(bb, True); o.shiftE(bb, True)
    ts.setBond(b); o.set_bond(b)
    B0 = o.bond_tensor(b)
    assert _relmax(ts.bond_tensor(b), B0) < 1e-12
    B = B0 + 0.1 * np.random.default_rng(b).standard_normal(B0.shape)
    assert _relmax(ts.forward(B), o.forward(B)) < T["P"]
    assert _relmax(ts.gradient(B), o.gradient(B)) < T["G"]
    Cg, lg, crg, ng = ts.quadcost(B, 1e-3)
    Co, cro = o.quadcost(B, 1e-3)
    assert Cg == pytest.approx(Co, rel=T["C"]) and lg.sum() + crg == pytest.approx(Cg, rel=1e-12) and crg == pytest.approx(cro, rel=1e-12)
    y = (np.asarray(o.labels) == o.target)
    assert ng == int(((o.forward(B) > 0.5) == y).sum())
    Bg, tg = ts.cgrad(B0, 4, 1e-3, 1e-10)
    Bo, to = o.cgrad(B0, 4, 1e-3, 1e-10)
    assert not tg["skipped"] and not to["skipped"]
    np.testing.assert_allclose(tg["cost"], to["cost"], rtol=T["cgc"])
    np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=T["cga"])
    assert _relmax(Bg, Bo) < T["cga"]
    # single.h:202-206: |r| < cconv at entry -> B untouched
    Bs, tsk = ts.cgrad(B0, 4, 1e-3, 1e30)
    assert tsk["skipped"] and tsk["npass_done"] == 0 and np.array_equal(Bs, B0)


@pytest.mark.parametrize("lam", [0.0, 1e-3])
@pytest.mark.parametrize("b", [1, 6, 11])
def test_single_fast_conj_matches_the_oracle(b, lam):
    """method = fast_conj of the per-label variant (single.h:290-398; tnml_set_option cg_method = 1): one image sum per CG
    step, residual by recurrence, the reference's regulariser term as written -- against oracle/single_oracle.c"""
    ts, o = _single_pair()
    for bb in range(1, b):
        ts.shiftE(bb, True); o.shiftE(bb, True)
    ts.setBond(b); o.set_bond(b)
    B0 = o.bond_tensor(b)
    ts.set_option("cg_method", 1)
    Bg, tg = ts.cgrad(B0, 4, lam, 1e-10)
    Bo, to = o.fast_cgrad(B0, 4, lam, 1e-10)
    assert not tg["skipped"] and tg["npass_done"] == 4
    np.testing.assert_allclose(tg["alpha"], to["alpha"], rtol=1e-7)
    np.testing.assert_allclose(tg["rnorm"], to["rnorm"], rtol=1e-6)
    assert _relmax(Bg, Bo) < 1e-6
    if lam > 0.:                                            # ... and it is NOT the conj optimiser once the regulariser enters
        _, tc = o.cgrad(B0, 4, lam, 1e-10)
        assert abs(tg["alpha"][1] / tc["alpha"][1] - 1) > 1e-9
    Bs, tsk = ts.cgrad(B0, 4, lam, 1e30)                    # entry check (single.h:328-332)
    assert tsk["skipped"] and np.array_equal(Bs, B0)
    ts.set_option("cg_method", 0)
    Bc, tcg = ts.cgrad(B0, 4, lam, 1e-10)
    _, toc = o.cgrad(B0, 4, lam, 1e-10)
    np.testing.assert_allclose(tcg["alpha"], toc["alpha"], rtol=1e-7)    # the option switches back cleanly


@pytest.mark.parametrize("lam", [1e-3, 0.0])
def test_single_exact_solver_matches_the_oracle(lam):
    """method = exact of the per-label variant (single.h:117-160; tnml_exact): the dense least-squares solution through
    the D x D normal matrix (columns from the forward / weighted-gradient kernels, rocSOLVER dsyevd, filtered inverse on
    the host) against the oracle's SVD of the design matrix; with lambda > 0 the regularised residual vanishes"""
    ts, o = _single_pair(N=10, NT=90, m=3, boost=300.0)
    for b in (1, 4):
        for bb in range(1 if b == 1 else 1, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
        ts.setBond(b); o.set_bond(b)
        import numpy.linalg as la
        B0 = o.bond_tensor(b)
        pcut = 1e-8 if lam > 0 else 1e-3                     # lambda = 0 needs a cut well above the Gram route's sqrt(eps) floor
        Bo = o.exact(b, lam, pcut)
        Bg = ts.exact(b, lam, pcut)
        assert _relmax(Bg, Bo) < (1e-6 if lam > 0 else 1e-4), (b, lam, _relmax(Bg, Bo))
        if lam > 0:
            r = ts.gradient(Bg) - lam * Bg
            assert np.abs(r).max() < 1e-7 * np.abs(ts.gradient(np.zeros_like(Bg))).max()
            assert ts.quadcost(Bg, lam)[0] <= o.cgrad(B0, 8, lam, 1e-12)[1]["cost"][-1] * (1 + 1e-9)
        break_out = b
        # walk on for the next bond from fresh environments
        ts.init(); o.init()


def test_single_exact_method_in_the_sweep():
    ts, o = _single_pair(N=8, NT=60, m=3, target=1, maxm=4)
    from tnml_amd.fixedl import mldmrg
    ts.set_option("cg_method", 2)
    ts.set_option_real("pcut", 1e-8)
    o.set_method("exact", pcut=1e-8)
    rg = mldmrg(ts, 1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    ro = o.mldmrg(1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(rg) == len(ro) == 14
    for a, b in zip(rg, ro):
        assert (a["c"], a["half"], a["origm"], a["newm"]) == (b["c"], b["half"], b["origm"], b["newm"])
        assert a["cost_cg"] == pytest.approx(b["cost_cg"], rel=1e-6)
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-6)


def test_single_fast_conj_sweep_matches_the_oracle():
    ts, o = _single_pair(N=10, NT=80, m=3, target=7, maxm=5)
    from tnml_amd.fixedl import mldmrg
    ts.set_option("cg_method", 1)
    o.set_method("fast_conj")
    rg = mldmrg(ts, 2, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    ro = o.mldmrg(2, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(rg) == len(ro) == 2 * 2 * 9
    for a, b in zip(rg, ro):
        assert (a["c"], a["half"], a["origm"], a["newm"]) == (b["c"], b["half"], b["origm"], b["newm"])
        assert a["cost_old"] == pytest.approx(b["cost_old"], rel=1e-7)
        assert a["cost_cg"] == pytest.approx(b["cost_cg"], rel=1e-7)
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-7)
    assert ro[-1]["cost"] < ro[0]["cost_old"]


@pytest.mark.parametrize("b,r", [(1, 3), (5, 6), (11, 4)])
def test_single_pinv_matches_the_oracle(b, r):
    """tnml_pinv (single.h:404-517 from a given start; the reference's is random and time-seeded): E = A V through the CG's own forward +
    weighted-gradient passes, the r x r algebra on the host, against the oracle's dense iteration -- V*E trace, singular values, B"""
    ts, o = _single_pair(N=12, NT=70, m=4, target=3)
    _walk(ts, o, b)
    D = int(np.prod(o.bond_shape(b)))
    V0 = np.random.default_rng(10 + b).standard_normal((D, min(r, D)))
    Bg, veg, Dg = ts.pinv(b, V0, 5, 1e-3)
    Bo, veo, Do = o.pinv(b, V0, 5, 1e-3)
    assert len(veg) == len(veo)
    np.testing.assert_allclose(veg, veo, rtol=1e-9)
    np.testing.assert_allclose(np.sort(Dg)[::-1], np.sort(Do)[::-1], rtol=1e-8)
    assert _relmax(Bg, Bo) < 1e-6
    Cg = ts.quadcost(Bg, 1e-3)[0]                              # what the reference prints: "After pinv, Cost"
    assert Cg == pytest.approx(o.quadcost(Bo, 1e-3)[0], rel=1e-7)


@pytest.mark.parametrize("noise", [1e-6, 1e-3])
def test_single_noise_split_matches_the_oracle(noise):
    """the density-matrix split with a noise term of the per-label variant (single.h:648-672) on the device -- rho from the Gram matrix
    the split forms anyway, drho from one GEMM over the images, a streaming kernel for the 2 x 2 weights and three weighted Gram matrices
    of the environment -- against the oracle: interior bonds in both half sweeps, the chain ends (no environment: drho = NT rho),
    kept dimension, truncation error and the two site tensors' product"""
    N = 10
    for ha in (1, 2):
        ts, o = _single_pair(N=N, NT=60, m=4, target=3, maxm=8)
        ts.set_option_real("noise", noise)
        if ha == 2:
            for b in range(1, N):
                ts.shiftE(b, True); o.shiftE(b, True)
        for b in (range(1, N) if ha == 1 else range(N - 1, 0, -1)):
            ts.setBond(b); o.set_bond(b)
            B0 = o.bond_tensor(b)
            B = B0 * (1.0 + 0.2 * np.cos(1.0 + np.arange(B0.size)).reshape(B0.shape))
            keep = max(2, min(B.shape[0], B.shape[3]))
            mg, teg, _ = ts.svd_split(B, b, ha, 1e-12, keep, 1)
            mo, teo = o.noise_split(B, b, ha, noise, 1e-12, keep, 1)
            assert mg == mo, (ha, b)
            assert teg == pytest.approx(teo, rel=1e-5, abs=1e-13), (ha, b)
            assert _relmax(ts.bond_tensor(b), o.bond_tensor(b)) < 1e-7, (ha, b)
            A = ts.get_site(b if ha == 1 else b + 1)
            G = np.einsum('asg,ash->gh', A, A) if ha == 1 else np.einsum('gtr,htr->gh', A, A)
            assert np.abs(G - np.eye(G.shape[0])).max() < 1e-10, (ha, b)
            ts.set_site(b, o.get_site(b)); ts.set_site(b + 1, o.get_site(b + 1))      # lockstep
            ts.shiftE(b, ha == 1); o.shiftE(b, ha == 1)
        ts.close()


def test_single_sweeps_with_noise_match_the_oracle():
    ts, o = _single_pair(N=10, NT=80, m=3, target=7, maxm=5)
    from tnml_amd.fixedl import mldmrg
    ts.set_option_real("noise", 1e-5)
    o.set_noise(1e-5)
    rg = mldmrg(ts, 2, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    ro = o.mldmrg(2, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(rg) == len(ro) == 2 * 2 * 9
    for a, b in zip(rg, ro):
        assert (a["c"], a["half"], a["origm"], a["newm"]) == (b["c"], b["half"], b["origm"], b["newm"])
        assert a["cost_cg"] == pytest.approx(b["cost_cg"], rel=1e-7)
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-7)
        assert a["truncerr"] == pytest.approx(b["truncerr"], rel=1e-3, abs=1e-12)
    with pytest.raises(Exception):                             # the fixedL variant has no such split (fixedL.cc has no noise)
        ts2, _ = _pair(N=8, NT=20, m=2)
        ts2.set_option_real("noise", 1e-5)


@pytest.mark.parametrize("normal", [True, False])
def test_single_full_sweeps_and_decision_function(normal):
    ts, o = _single_pair(N=10, NT=80, m=3, target=7, normal=normal, maxm=5)
    from tnml_amd.fixedl import mldmrg
    rg = mldmrg(ts, 2, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    ro = o.mldmrg(2, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    assert len(rg) == len(ro) == 2 * 2 * 9
    for a, b in zip(rg, ro):
        assert (a["c"], a["half"], a["origm"], a["newm"]) == (b["c"], b["half"], b["origm"], b["newm"])
        assert a["cost_old"] == pytest.approx(b["cost_old"], rel=1e-7)
        assert a["cost_cg"] == pytest.approx(b["cost_cg"], rel=1e-7)
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-7)
        assert a["norm_oB"] == pytest.approx(b["norm_oB"], rel=1e-7)
        assert a["truncerr"] == pytest.approx(b["truncerr"], rel=1e-3, abs=1e-12)
    assert ro[-1]["cost"] < 0.9 * ro[0]["cost_old"]
    w, pred, cnt, ninc = ts.classify()                      # decision function f(x) of separate_fulltest.cc
    f_or = np.array([o.output(i) for i in range(o.NT)])
    assert w.shape == (o.NT, 1) and _relmax(w[:, 0], f_or) < 1e-6
    y = (np.asarray(o.labels) == 7)
    np.testing.assert_array_equal(pred, (f_or > 0.5).astype(np.int32))
    assert int(ninc.sum()) == int(((f_or > 0.5) != y).sum())


def test_single_and_separate_fulltest_cli(tmp_path):
    """the C++ `single <inputfile>` driver for all ten labels and `separate_fulltest` on the resulting L<n>/W<n> files,
    against the oracle started from the same initial W and image order (labels round-robin, single.cc:156-181)"""
    import os
    import re
    import subprocess
    from oracle import pyoracle
    from tnml_amd import hostlib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, per_label = 16, 16
    labels = synth.synthetic_labels(10 * per_label, seed=6, per_label=per_label)
    tl = synth.synthetic_labels(120, seed=23)
    allpx = np.clip(synth.synthetic_images(N, np.concatenate([labels, tl]), seed=6).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    pixels, tp = allpx[:len(labels)], allpx[len(labels):]
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    synth.write_idx(data, tp, tl, train=False)
    keys = "datadir = %s\nfeature_scale = 1\n" % data
    # the image order of the driver: label 0's first image, label 1's first image, ...
    px, lab, _ = hostlib.read_mnist(data, True, per_label)
    by = [list(np.flatnonzero(lab == l)) for l in range(10)]
    order = [by[l][k] for k in range(per_label) for l in range(10)]
    phi = pyoracle.features_single(px[order], True)
    # BASELINE config 4: the ten per-label trainings through the driver's own launcher (`labels = all`: one child `single` per label in
    # its directory L<n>; on this one-GPU box `share_device = yes` runs them one after the other on device 0)
    allinp = tmp_path / "input_all"
    allinp.write_text("input\n{\n%slabels = all\nshare_device = yes\nNtrain = %d\nNsweep = 1\ncutoff = 1E-10\nmaxm = 5\nminm = 2\nninitial = 3\n"
                      "lambda = 1E-3\nNpass = 3\nseed = 4\nnthread = 2\n}\n" % (keys, per_label))
    launch = subprocess.run([os.path.join(root, "tnml_amd", "single"), str(allinp)], capture_output=True, text=True, cwd=tmp_path, timeout=900)
    assert launch.returncode == 0, launch.stdout[-1500:] + launch.stderr[-1500:]
    assert "10 of 10 per-label trainings finished" in launch.stdout
    for L in range(10):
        wd = tmp_path / ("L%d" % L)
        assert "label %d -> device 0, directory L%d" % (L, L) in launch.stdout and "label %d on device 0: done" % L in launch.stdout

        class run:                                          # the child's log, where the per-label run used to be captured
            stdout = (wd / "log").read_text()
        assert os.path.exists(wd / ("W%d" % L)) and "%d training images with selected label L=%d" % (per_label, L) in run.stdout
        if L in (3, 8):                                     # full trajectory check on two of the ten
            w0 = str(tmp_path / ("W0ref%d" % L))
            hostlib.build_initial_single(data, per_label, L, 3, 4, True, w0)
            o = pyoracle.SingleOracle(phi, lab[order], L, hostlib.read_mps(w0))
            o.init()
            ro = o.mldmrg(1, 5, 2, 1e-10, 3, 1e-3, 1e-10)
            c_old = [float(a) for a, _ in re.findall(r"Cost = ([0-9.eE+-]+) --> ([0-9.eE+-]+)", run.stdout)]
            c_cg = [float(b) for _, b in re.findall(r"Cost = ([0-9.eE+-]+) --> ([0-9.eE+-]+)", run.stdout)]
            c_svd = [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+) \(", run.stdout)]
            newm = [int(x) for x in re.findall(r"New m=(\d+)", run.stdout)]
            assert len(c_svd) == len(ro) == 2 * (N - 1)
            nt = float(len(lab))
            np.testing.assert_allclose(c_old, [r["cost_old"] / nt for r in ro], rtol=2e-5, atol=2e-10)
            np.testing.assert_allclose(c_cg, [r["cost_cg"] / nt for r in ro], rtol=2e-5, atol=2e-10)
            np.testing.assert_allclose(c_svd, [r["cost"] / nt for r in ro], rtol=2e-5, atol=2e-10)
            assert all(abs(a - r["newm"]) <= 1 for a, r in zip(newm, ro))
            m0 = re.search(r"Before DMRG, Cost = ([0-9.eE+-]+)", run.stdout)
            assert m0 and float(m0.group(1)) == pytest.approx(ro[0]["cost_old"] / per_label, rel=1e-6)   # divides by the Ntrain key (single.cc:218)
    # method = fast_conj through the command line (single.cc:44, single.h:599): its own log format (pass and |r| on one line,
    # no cost per pass) and the oracle's fast_cgrad trajectory
    wd = tmp_path / "Lfast"
    wd.mkdir()
    inp = wd / "input"
    inp.write_text("input\n{\n%slabel = 3\nNtrain = %d\nNsweep = 1\ncutoff = 1E-10\nmaxm = 5\nminm = 2\nninitial = 3\n"
                   "lambda = 1E-3\nNpass = 3\nseed = 4\nnthread = 2\nmethod = fast_conj\n}\n" % (keys, per_label))
    run = subprocess.run([os.path.join(root, "tnml_amd", "single"), str(inp)], capture_output=True, text=True, cwd=wd, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    o = pyoracle.SingleOracle(phi, lab[order], 3, hostlib.read_mps(str(tmp_path / "W0ref3")))
    o.init()
    o.set_method("fast_conj")
    ro = o.mldmrg(1, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    c_cg = [float(b) for _, b in re.findall(r"Cost = ([0-9.eE+-]+) --> ([0-9.eE+-]+)", run.stdout)]
    c_svd = [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+) \(", run.stdout)]
    assert len(c_svd) == len(ro) == 2 * (N - 1)
    # the as-written regulariser term (:379) makes the iteration erratic (the cost goes UP on some bonds), so round-off
    # differences grow faster than with conj: tight on the first bonds, loose on the whole sweep
    np.testing.assert_allclose(c_cg[:12], [r["cost_cg"] / float(len(lab)) for r in ro][:12], rtol=2e-5, atol=2e-10)
    np.testing.assert_allclose(c_svd[:12], [r["cost"] / float(len(lab)) for r in ro][:12], rtol=2e-5, atol=2e-10)
    np.testing.assert_allclose(c_cg, [r["cost_cg"] / float(len(lab)) for r in ro], rtol=3e-2)
    np.testing.assert_allclose(c_svd, [r["cost"] / float(len(lab)) for r in ro], rtol=3e-2)
    assert re.search(r"^  Conj grad pass 1   \|r\| = [0-9.]+E[+-][0-9]+$", run.stdout, re.M) and not re.search(r"^  1 C = ", run.stdout, re.M)
    # method = exact through the command line (single.h:600): the dense solver at these toy sizes, against the oracle
    wde = tmp_path / "Lexact"                               # a fresh directory: the driver continues from an existing W<label>
    wde.mkdir()
    ex = wde / "input_exact"
    ex.write_text(inp.read_text().replace("method = fast_conj", "method = exact\npcut = 1E-8"))
    run = subprocess.run([os.path.join(root, "tnml_amd", "single"), str(ex)], capture_output=True, text=True, cwd=wde, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    o = pyoracle.SingleOracle(phi, lab[order], 3, hostlib.read_mps(str(tmp_path / "W0ref3")))
    o.init()
    o.set_method("exact", pcut=1e-8)
    ro = o.mldmrg(1, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    c_cg = [float(b) for _, b in re.findall(r"Cost = ([0-9.eE+-]+) --> ([0-9.eE+-]+)", run.stdout)]
    assert len(c_cg) == len(ro)
    np.testing.assert_allclose(c_cg[:10], [r["cost_cg"] / float(len(lab)) for r in ro][:10], rtol=1e-4, atol=1e-9)
    assert "Conj grad pass" not in run.stdout
    # noise = 1E-5 through the command line: the density-matrix split (single.h:648-672), "Trunc err" instead of "SVD trunc err" (:669)
    wdn = tmp_path / "Lnoise"
    wdn.mkdir()
    nz = wdn / "input_noise"
    nz.write_text(inp.read_text().replace("method = fast_conj", "method = conj\nnoise = 1E-5"))
    run = subprocess.run([os.path.join(root, "tnml_amd", "single"), str(nz)], capture_output=True, text=True, cwd=wdn, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    o = pyoracle.SingleOracle(phi, lab[order], 3, hostlib.read_mps(str(tmp_path / "W0ref3")))
    o.init()
    o.set_noise(1e-5)
    ro = o.mldmrg(1, 5, 2, 1e-10, 3, 1e-3, 1e-10)
    c_svd = [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+) \(", run.stdout)]
    assert len(c_svd) == len(ro) and "SVD trunc err" not in run.stdout and run.stdout.count("Trunc err = ") == len(ro)
    np.testing.assert_allclose(c_svd, [r["cost"] / float(len(lab)) for r in ro], rtol=1e-6, atol=1e-10)
    # method = pinv (single.h:596-604): the pseudo-inverse solution is a diagnostic -- its cost is printed -- and the update is cgrad: the
    # same sweep as method = conj, plus one block of diagnostic lines per bond update
    logs = {}
    for meth in ("conj", "pinv"):
        wdm = tmp_path / ("L" + meth)
        wdm.mkdir()
        im = wdm / "input_m"
        im.write_text(inp.read_text().replace("method = fast_conj", "method = %s\nNtarget = 4" % meth))
        run = subprocess.run([os.path.join(root, "tnml_amd", "single"), str(im)], capture_output=True, text=True, cwd=wdm, timeout=300)
        assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
        logs[meth] = run.stdout
    after = {k: re.findall(r"--> After SVD, Cost = [0-9.eE+-]+ \(", v) for k, v in logs.items()}
    assert after["pinv"] == after["conj"] and len(after["pinv"]) == 2 * (N - 1)
    pc = [float(x) for x in re.findall(r"After pinv, Cost = ([0-9.eE+-]+)", logs["pinv"])]
    assert len(pc) == 2 * (N - 1) and all(np.isfinite(pc)) and "After pinv" not in logs["conj"]
    assert logs["pinv"].count("Using pcut = 1.00E-08") == 2 * (N - 1) and logs["pinv"].count("Initial V*E = ") == 2 * (N - 1)
    bad = wd / "input_bad"
    bad.write_text(inp.read_text().replace("fast_conj", "newton"))
    run = subprocess.run([os.path.join(root, "tnml_amd", "single"), str(bad)], capture_output=True, text=True, cwd=wd, timeout=300)
    assert run.returncode != 0 and "not recognized" in run.stdout
    # evaluator on the ten weight files
    (tmp_path / "sites").write_bytes((tmp_path / "L0" / "sites").read_bytes())
    tin = tmp_path / "input_test"
    tin.write_text("input\n{\n%simglen = 4\n}\n" % keys)
    run = subprocess.run([os.path.join(root, "tnml_amd", "separate_fulltest"), str(tin)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    phit = pyoracle.features_single(tp, True)
    O = np.zeros((10, len(tl)))
    for L in range(10):
        oo = pyoracle.SingleOracle(phit, tl, L, hostlib.read_mps(str(tmp_path / ("L%d" % L) / ("W%d" % L))))
        O[L] = [oo.output(i) for i in range(len(tl))]
    pred = np.abs(O).argmax(axis=0)
    m = re.search(r"(\d+)/(\d+) correct", run.stdout)
    assert m and int(m.group(2)) == len(tl) and int(m.group(1)) == int((pred == tl).sum())
    costs = [float(x) for x in re.findall(r"Digit \d C = ([0-9.eE+-]+)", run.stdout)]
    ref = [float(np.sum((O[n] - (tl == n)) ** 2)) for n in range(10)]
    np.testing.assert_allclose(costs, ref, rtol=1e-8)
    assert float(re.search(r"Total C = ([0-9.eE+-]+)", run.stdout).group(1)) == pytest.approx(sum(ref), rel=1e-8)


def test_rccl_path_with_a_one_rank_communicator(monkeypatch):
    """TNML_FORCE_COMM=1: a 1-rank RCCL communicator is built and every gradient / cost all-reduce goes through
    ncclAllReduce on the compute stream -- the results must be unchanged (exercises the multi-GPU code path on one GPU)"""
    from tnml_amd.fixedl import TrainStates, mldmrg
    monkeypatch.setenv("TNML_FORCE_COMM", "1")
    ts, o = _pair(N=10, NT=40, m=4)
    ts.comm_init(TrainStates.comm_unique_id())
    ts.setBond(1)
    B = o.bond_tensor(1)
    assert _relmax(ts.gradient(B), o.gradient(B)) < 1e-9
    assert ts.quadcost(B, 1e-3)[0] == pytest.approx(o.quadcost(B, 1e-3)[0], rel=1e-11)
    rg = mldmrg(ts, 1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    ro = o.mldmrg(1, 4, 2, 1e-10, 3, 1e-3, 1e-10)
    np.testing.assert_allclose([r["cost"] for r in rg], [r["cost"] for r in ro], rtol=1e-8)


def _spectra():
    rng = np.random.default_rng(77)
    yield "identity (240-fold degenerate)", np.ones(240)
    yield "two plateaus", np.concatenate([np.full(60, 3.0), np.full(180, 0.5)])
    yield "rank one", np.concatenate([[2.0], np.zeros(239)])
    yield "rank 119 then exact zeros", np.concatenate([np.linspace(2.0, 1.0, 119), np.zeros(121)])
    yield "pairs of equal values", np.repeat(np.exp(-0.08 * np.arange(120)), 2)
    yield "geometric 1e0..1e-14", np.logspace(0, -14, 240)
    yield "cluster of 40 within 1e-13", np.concatenate([np.linspace(2, 1, 100), 0.5 + 1e-13 * rng.random(40), np.logspace(-1, -6, 100)])
    yield "random uniform", np.sort(rng.random(240))[::-1]


@pytest.mark.parametrize("name,sv0", list(_spectra()), ids=[n for n, _ in _spectra()])
def test_svd_split_hard_spectra(name, sv0):
    """degenerate, clustered, rank-deficient and widely graded singular value spectra through the in-house eigensolver
    (split, inverse iteration, Newton-Schulz / Cholesky-QR repair, rocSOLVER fallback): the singular values, the optimal
    rank-120 reconstruction and the isometry of the site the sweep leaves must come out right in every case"""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, m, NT = 20, 120, 16
    labels = synth.synthetic_labels(NT, seed=1)
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    ts.set_mps(synth.random_mps(N, m, seed=2))
    rng = np.random.default_rng(len(name))
    sv0 = np.sort(np.asarray(sv0, dtype=float))[::-1]
    U0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    V0, _ = np.linalg.qr(rng.standard_normal((240, 240)))
    M = (U0 * sv0) @ V0.T
    B = M.reshape(120, 2, 2, 120, order="F")
    best_err = np.sqrt(np.sum(sv0[120:] ** 2))                      # Eckart-Young: Frobenius error of the best rank-120 approximation
    for ha in (1, 2):
        mg, te, sv = ts.svd_split(B, 8, ha, 0.0, 120, 120)
        assert mg == 120
        np.testing.assert_allclose(sv ** 2, sv0 ** 2, rtol=1e-7, atol=3e-14 * sv0[0] ** 2)   # Gram route: eps*lambda_max floor
        assert te == pytest.approx(np.sum(sv0[120:] ** 2) / np.sum(sv0 ** 2), rel=1e-6, abs=1e-13)
        newB = ts.bond_tensor(8).reshape(240, 240, order="F")
        err = np.linalg.norm(newB - M)
        assert err <= best_err * (1 + 1e-6) + 2e-7 * sv0[0], (err, best_err)
        A = ts.get_site(8 if ha == 1 else 9)
        Q = A.reshape(240, 120, order="F") if ha == 1 else A.reshape(120, 240, order="F").T
        if ha == 1:                                                   # the Gram-side factor is an isometry whatever the spectrum
            np.testing.assert_allclose(Q.T @ Q, np.eye(120), atol=1e-9)
    st = ts.svd_stats()
    assert st["fallbacks"] <= 2, st


def _spectra_300():
    rng = np.random.default_rng(78)
    yield "identity (300-fold degenerate)", np.ones(300)
    yield "two plateaus", np.concatenate([np.full(70, 3.0), np.full(230, 0.5)])
    yield "rank one", np.concatenate([[2.0], np.zeros(299)])
    yield "rank 149 then exact zeros", np.concatenate([np.linspace(2.0, 1.0, 149), np.zeros(151)])
    yield "geometric 1e0..1e-14", np.logspace(0, -14, 300)
    yield "cluster of 50 within 1e-13", np.concatenate([np.linspace(2, 1, 120), 0.5 + 1e-13 * rng.random(50), np.logspace(-1, -6, 130)])
    yield "random uniform", np.sort(rng.random(300))[::-1]


@pytest.mark.parametrize("name,sv0", list(_spectra_300()), ids=[n for n, _ in _spectra_300()])
def test_svd_split_hard_spectra_on_the_workgroup_cluster(name, sv0):
    """the same hard spectra at maxm = 150 (Gram side n = 300 > 240): tridiagonalisation on the cluster of workgroups (eigh_mc.hip,
    with and without its rank-adaptive exit), tridiagonal kernels in their n <= 640 instantiations, Cholesky QR on rocSOLVER dpotrf"""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, m, NT = 24, 150, 16                                             # bond 10 of 24 sites: 150 x 150, Label on the right environment (c0 = 12)
    labels = synth.synthetic_labels(NT, seed=1)
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    ts.set_mps(synth.random_mps(N, m, seed=2))
    rng = np.random.default_rng(len(name))
    sv0 = np.sort(np.asarray(sv0, dtype=float))[::-1]
    U0, _ = np.linalg.qr(rng.standard_normal((300, 300)))
    V0, _ = np.linalg.qr(rng.standard_normal((300, 300)))
    M = (U0 * sv0) @ V0.T
    B = M.reshape(150, 2, 2, 150, order="F")
    best_err = np.sqrt(np.sum(sv0[150:] ** 2))
    for ha, exit_on in ((1, 1), (2, 1), (1, 0)):
        ts.set_option("sytrd_exit", exit_on)
        mg, te, sv = ts.svd_split(B, 10, ha, 0.0, 150, 150)
        assert mg == 150
        np.testing.assert_allclose(sv ** 2, sv0 ** 2, rtol=1e-7, atol=3e-14 * sv0[0] ** 2)
        assert te == pytest.approx(np.sum(sv0[150:] ** 2) / np.sum(sv0 ** 2), rel=1e-6, abs=1e-13)
        newB = ts.bond_tensor(10).reshape(300, 300, order="F")
        err = np.linalg.norm(newB - M)
        assert err <= best_err * (1 + 1e-6) + 2e-7 * sv0[0], (err, best_err)
        A = ts.get_site(10 if ha == 1 else 11)
        Q = A.reshape(300, 150, order="F") if ha == 1 else A.reshape(150, 300, order="F").T
        if ha == 1:
            np.testing.assert_allclose(Q.T @ Q, np.eye(150), atol=1e-9)
    st = ts.svd_stats()
    # The rocSOLVER fallback is a correct path (everything above holds through it), this bounds how often it is needed.  On an exactly
    # degenerate plateau that straddles the kept basis whether inverse iteration + Cholesky QR reaches 1e-6 before the polish step is
    # decided by the last bits of the Gram matrix (round 5: the in-house k_dgemm_small and rocBLAS dgemm give 3 and 2 fallbacks in 3 splits)
    assert st["fallbacks"] <= (3 if "plateau" in name or "degenerate" in name else 2), st


def test_workgroup_cluster_that_gives_up_falls_back_to_rocsolver():
    """k_sytrd_mc never hangs the GPU: a workgroup that waits in vain for a peer sets the abort word, every workgroup leaves, the
    host sees the status and redoes the split with rocsolver_dsyevd.  Option mc_spin_max = 0 makes the very first failed poll give up."""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, m, NT = 24, 150, 16                                             # bond 10 of 24 sites: 150 x 150, Label on the right environment (c0 = 12)
    labels = synth.synthetic_labels(NT, seed=1)
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    ts.set_mps(synth.random_mps(N, m, seed=2))
    ts.set_option("mc_spin_max", 0)
    rng = np.random.default_rng(5)
    sv0 = np.exp(-0.05 * np.arange(300))
    U0, _ = np.linalg.qr(rng.standard_normal((300, 300)))
    V0, _ = np.linalg.qr(rng.standard_normal((300, 300)))
    M = (U0 * sv0) @ V0.T
    for rep in range(2):                                               # the second call finds the abort word cleared
        mg, te, sv = ts.svd_split(M.reshape(150, 2, 2, 150, order="F"), 10, 1, 0.0, 150, 150)
        assert mg == 150
        np.testing.assert_allclose(sv, sv0, rtol=1e-7, atol=1e-7)
        assert np.abs(ts.bond_tensor(10).reshape(300, 300, order="F") - (U0[:, :150] * sv0[:150]) @ V0[:, :150].T).max() < 1e-9
    assert ts.svd_stats()["fallbacks"] == 2


def test_invariants_of_the_split_and_of_the_gauge():
    """SURVEY.md section 4, invariants 2 and 4 on the HIP path itself (no oracle involved):
    |B - newB|^2 = sum of the discarded sigma^2 = truncerr * sum sigma^2; the data cost evaluated with the
    un-optimised bond tensor of the NEXT bond equals the "After SVD" data cost of this bond (gauge invariance)."""
    ts, o = _pair(N=12, NT=80, m=6, maxm=6)
    lam = 1e-3
    for b in range(1, 7):
        r = ts.bond_update(b, 1, 4, 2, 1e-10, 3, lam, 1e-10)           # maxm 4 < 6: real truncation on the interior bonds
        data_cost = r["cost"] - r["reg_cost"]
        ts.setBond(b + 1)
        Bn = ts.bond_tensor(b + 1)
        C, lc, cr, nc = ts.quadcost(Bn, lam)
        assert C - cr == pytest.approx(data_cost, rel=1e-10)            # same network, different bond: same outputs
        assert nc == r["ncorrect"]
    # the split alone on a fresh random bond tensor
    ts2, o2 = _pair(N=12, NT=40, m=6, maxm=6)
    for bb in range(1, 4):
        ts2.shiftE(bb, True)
    ts2.setBond(4)
    B = ts2.bond_tensor(4) + 0.2 * np.random.default_rng(4).standard_normal((6, 2, 2, 6))
    mnew, te, sv = ts2.svd_split(B, 4, 1, 0.0, 5, 1)
    newB = ts2.bond_tensor(4)
    assert mnew == 5
    assert np.sum((B - newB) ** 2) == pytest.approx(np.sum(sv[5:] ** 2), rel=1e-8)
    assert te == pytest.approx(np.sum(sv[5:] ** 2) / np.sum(sv ** 2), rel=1e-8)


def test_fixedl_initial_w_from_ten_per_label_files(tmp_path):
    """fixedL.cc:682-701: with files W0..W9 present (and no W) the initial weight MPS is the sum of the ten per-label MPS,
    each with the Label index attached on site N/2.  The ten files come from the `single` driver (feature = series, the
    map fixedL uses); the summed network must output f_l(x) on label component l for every image."""
    import os
    import re
    import shutil
    import subprocess
    from oracle import pyoracle
    from tnml_amd import hostlib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, per_label = 16, 8
    labels = synth.synthetic_labels(10 * per_label, seed=12, per_label=per_label)
    pixels = np.clip(synth.synthetic_images(N, labels, seed=12).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    comb = tmp_path / "comb"
    comb.mkdir()
    for L in range(10):
        wd = tmp_path / ("L%d" % L)
        wd.mkdir()
        (wd / "input").write_text("input\n{\ndatadir = %s\nlabel = %d\nNtrain = %d\nNsweep = 1\ncutoff = 1E-10\nmaxm = 4\nminm = 2\nninitial = 2\n"
                                  "lambda = 1E-3\nNpass = 2\nseed = 2\nfeature = series\nfeature_scale = 255\n}\n" % (data, L, per_label))
        run = subprocess.run([os.path.join(root, "tnml_amd", "single"), "input"], capture_output=True, text=True, cwd=wd, timeout=300)
        assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-1000:]
        shutil.copy(wd / ("W%d" % L), comb / ("W%d" % L))
    (comb / "input").write_text("input\n{\ndatadir = %s\nNtrain = %d\nNbatch = 4\nNsweep = 0\nmaxm = 10\nlambda = 1E-3\nfeature_scale = 255\n}\n" % (data, per_label))
    run = subprocess.run([os.path.join(root, "tnml_amd", "fixedL"), "input"], capture_output=True, text=True, cwd=comb, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    assert "Found separate W0,W1,...,W9 MPS: summing" in run.stdout and "Summing all 10 label states together" in run.stdout
    W = hostlib.read_mps(str(comb / "W"))
    assert [A.ndim == 4 for A in W] == [j == N // 2 for j in range(1, N + 1)]
    px, lab, _ = hostlib.read_mnist(data, True, per_label)
    g = px.astype(np.float64) / 255.0
    phi = np.stack([np.ones_like(g), 255.0 * ((g / 255.0) / 4.0)], axis=-1)
    o = pyoracle.Oracle(phi, lab, W)
    out = np.stack([o.toverlap(i) for i in range(len(lab))])                     # [n, 10]
    for L in range(10):
        so = pyoracle.SingleOracle(phi, lab, L, hostlib.read_mps(str(comb / ("W%d" % L))))
        fL = np.array([so.output(i) for i in range(len(lab))])
        np.testing.assert_allclose(out[:, L], fL, rtol=1e-5, atol=1e-6 * np.abs(fL).max())     # the sum is truncated at Cutoff 1E-10
    o.init()
    C0 = o.quadcost(o.bond_tensor(1), 1e-3)[0] / len(lab)
    m0 = re.search(r"Before starting DMRG Cost = ([0-9.eE+-]+)", run.stdout)
    assert m0 and float(m0.group(1)) == pytest.approx(C0, rel=1e-8)
    y = np.eye(10)[lab]
    assert C0 == pytest.approx((np.sum((y - out) ** 2) + 1e-3 * np.sum(o.bond_tensor(1) ** 2)) / len(lab), rel=1e-9)


def test_python_multi_gpu_driver_matches_the_cpp_driver(tmp_path):
    """`python -m tnml_amd.train <inputfile>` (one process per GPU under torch.distributed.run; here a single process) prints
    the same per-bond lines as the C++ `fixedL` driver for the same input: both are thin loops over tnml_bond_update"""
    import os
    import re
    import subprocess
    import sys
    from tnml_amd import hostlib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, per_label = 16, 12
    labels = synth.synthetic_labels(10 * per_label, seed=15, per_label=per_label)
    pixels = np.clip(synth.synthetic_images(N, labels, seed=15).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    text = ("input\n{\ndatadir = %s\nNtrain = %d\nNbatch = 4\nNsweep = 1\ncutoff = 1E-10\nmaxm = 5\nminm = 2\nninitial = 3\n"
            "lambda = 1E-3\nNpass = 3\nseed = 5\nfeature_scale = 255\n}\n" % (data, per_label))
    outs = {}
    for name, cmd in (("cpp", [os.path.join(root, "tnml_amd", "fixedL"), "input"]), ("py", [sys.executable, "-m", "tnml_amd.train", "input"])):
        wd = tmp_path / name
        wd.mkdir()
        (wd / "input").write_text(text)
        env = dict(os.environ, PYTHONPATH=root)
        run = subprocess.run(cmd, capture_output=True, text=True, cwd=wd, timeout=600, env=env)
        assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
        outs[name] = run.stdout
        assert os.path.exists(wd / "W") and os.path.exists(wd / "sites")
    pick = lambda s: re.findall(r"^(SVD trunc err.*|Original m=.*|--> After SVD.*|Percent correct.*|  Cost = .*|Before starting DMRG.*)$", s, re.M)
    a, b = pick(outs["cpp"]), pick(outs["py"])
    assert len(a) == len(b) and len(a) > 100
    assert a == b
    Wc, Wp = hostlib.read_mps(str(tmp_path / "cpp" / "W")), hostlib.read_mps(str(tmp_path / "py" / "W"))
    assert all(np.array_equal(x, y) for x, y in zip(Wc, Wp))


def test_properties_at_the_full_baseline_size():
    """BASELINE config 3 itself (N=784 sites, 60 000 images, m=120, fp64: 250 GB on the one GPU): size-independent
    properties of the HIP path at the size the benchmark runs -- linearity of the forward map in B, the data cost seen
    from two neighbouring bonds (gauge invariance), additivity of the gradient over image shards, run-to-run determinism."""
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, NT, m, b0 = 784, 60000, 120, 400
    labels = synth.synthetic_labels(NT)
    pixels = synth.synthetic_images(N, labels)
    W = synth.random_mps(N, m, seed=1)

    def walked(sl):
        t = TrainStates(labels[sl], N, m, pixels=pixels[sl])
        t.set_mps(W)
        t.init()
        for bb in range(1, b0):
            t.shiftE(bb, True)
        t.setBond(b0)
        return t
    ts = walked(slice(0, NT))
    assert ts.device_bytes() > 200e9
    B1 = ts.bond_tensor(b0)
    assert B1.shape == (120, 2, 2, 120)
    B2 = np.random.default_rng(0).standard_normal(B1.shape) * np.abs(B1).max()
    P1, P2, P12 = ts.forward(B1), ts.forward(B2), ts.forward(B1 + 2.0 * B2)
    assert _relmax(P12, P1 + 2.0 * P2) < 1e-12
    G = ts.gradient(B1)
    assert _relmax(ts.gradient(B1), G) == 0.0                                  # deterministic reductions: bit-identical
    C, lc, cr, nc = ts.quadcost(B1, 1e-3)
    ts.shiftE(b0, True)
    ts.setBond(b0 + 1)
    C2, lc2, cr2, nc2 = ts.quadcost(ts.bond_tensor(b0 + 1), 1e-3)
    assert C2 - cr2 == pytest.approx(C - cr, rel=1e-11) and nc2 == nc
    np.testing.assert_allclose(lc2, lc, rtol=1e-10)
    ts.close()
    Gs = np.zeros_like(G)
    for sl in (slice(0, NT // 2), slice(NT // 2, NT)):
        t2 = walked(sl)
        Gs += t2.gradient(B1)
        t2.close()
    assert _relmax(Gs, G) < 1e-11


@pytest.mark.parametrize("N,NT,m,dtype", [(20, 300, 8, "f64"), (16, 700, 24, "f64_e32"), (24, 48, 150, "f64")])
def test_memory_estimate_covers_what_a_sweep_allocates(N, NT, m, dtype):
    """tnml_estimate_bytes is what the drivers size maxm with (tnml_plan_maxm) before any environment exists; the
    environment slabs are allocated lazily during the first sweep.  An estimate below the real footprint would surface
    as a failed hipMalloc in the middle of a sweep: after a full sweep the context must hold no more than estimated
    (and not absurdly less: the plan would give memory away)."""
    from tnml_amd.fixedl import TrainStates, mldmrg
    pixels, labels, phi, W = make_problem(N, NT, min(m, 8), 3, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, pixels=pixels, dtype=dtype)
    ts.set_mps(W)
    ts.init()
    mldmrg(ts, 1, m, max(2, m // 2), 1e-10, 2, 1e-3, 1e-10)
    used, est = ts.device_bytes(), ts.estimate_bytes()
    ts.close()
    assert est > 0 and used <= est, (used, est)
    assert used > 0.25 * est, (used, est)


def test_cg_on_ill_conditioned_bonds_is_at_least_as_close_to_extended_precision_as_the_oracle():
    """Label-on-B bonds have CG step sizes of 10^2 and the late passes are conditioning-limited: two fp64 implementations agree on the
    first three step sizes to 1e-9 and differ in the fourth by percents.  Which one is right?  The numpy restatement of the same CG run in
    80-bit extended precision (np.longdouble) is the referee: the HIP path's third residual norm must be at least as close to it as the C
    oracle's (it is 2-50x closer: blocked MFMA accumulation and fixed-order tree sums against sequential sums over 1e4-1e6 terms)."""
    from oracle import np_restatement as npr
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    if np.finfo(np.longdouble).eps > 1e-18:
        pytest.skip("np.longdouble is not an extended type here")
    for (N, dims, NT, boost, lam, seed) in ((4, [1, 2, 2, 1, 1], 257, 200.0, 1e-2, 7), (6, [1, 2, 2, 4, 2, 1, 1], 100, 30.0, 1e-2, 13)):
        pixels, labels, phi, _ = make_problem(N, NT, 2, seed, pixel_boost=boost)
        W = _mps_with_dims(dims, 100 + seed)
        for b in (1, 2):
            o = pyoracle.Oracle(phi, labels, W); o.init()
            ts = TrainStates(labels, N, max(dims), phi=phi); ts.set_mps(W); ts.init()
            n = npr.NpFixedL(phi, labels, W)
            n.phi = n.phi.astype(np.longdouble); n.W = [None] + [x.astype(np.longdouble) for x in n.W[1:]]; n.delta = n.delta.astype(np.longdouble)
            n.init()
            for bb in range(1, b):
                o.shiftE(bb, True); ts.shiftE(bb, True); n.shiftE(bb, True)
            o.set_bond(b); ts.setBond(b); n.set_bond(b)
            B0 = o.bond_tensor(b)
            _, to = o.cgrad(B0, 4, lam, 0.0)
            _, tg = ts.cgrad(B0, 4, lam, 0.0)
            _, tx = n.cgrad(n.bond_tensor(b), 4, np.longdouble(lam), 0.0)
            np.testing.assert_allclose(tg["alpha"][:3], [float(x) for x in tx["alpha"][:3]], rtol=1e-8)
            np.testing.assert_allclose(to["alpha"][:3], [float(x) for x in tx["alpha"][:3]], rtol=1e-8)
            rx = float(tx["rnorm"][2])
            err_hip, err_orc = abs(tg["rnorm"][2] - rx) / rx, abs(to["rnorm"][2] - rx) / rx
            assert err_hip <= 1.5 * err_orc + 1e-12, (N, b, err_hip, err_orc)
            ts.close()


@pytest.mark.parametrize("b", [2, 6, 9])
def test_forward_gradient_and_cost_against_an_extended_precision_evaluation(b):
    """absolute accuracy, not just agreement of two fp64 codes: the numpy restatement evaluated in 80-bit extended precision is the
    referee for B*t.v, the gradient and the cost of one bond (Label on the right environment, on B, on the left environment) -- the HIP
    path and the C oracle must both sit within a few hundred fp64 ulps of it"""
    from oracle import np_restatement as npr
    if np.finfo(np.longdouble).eps > 1e-18:
        pytest.skip("np.longdouble is not an extended type here")
    ts, o = _pair(N=12, NT=200, m=8)
    pixels, labels, phi, W = make_problem(12, 200, 8, 3, pixel_boost=200.0)
    n = npr.NpFixedL(phi, labels, W)
    n.phi = n.phi.astype(np.longdouble); n.W = [None] + [x.astype(np.longdouble) for x in n.W[1:]]; n.delta = n.delta.astype(np.longdouble)
    n.init()
    for bb in range(1, b):
        ts.shiftE(bb, True); o.shiftE(bb, True); n.shiftE(bb, True)
    ts.setBond(b); o.set_bond(b); n.set_bond(b)
    B = o.bond_tensor(b) + 0.05 * np.random.default_rng(b).standard_normal(o.bond_tensor(b).shape)
    Bx = B.astype(np.longdouble)
    Px, Gx = n.forward(Bx), n.gradient(Bx)
    Cx = float(n.quadcost(Bx, np.longdouble(1e-3))[0])
    for name, impl in (("HIP", ts), ("oracle", o)):
        P, G = impl.forward(B), impl.gradient(B)
        assert float(np.abs(P - Px).max() / np.abs(Px).max()) < 2e-14, name
        assert float(np.abs(G - Gx).max() / np.abs(Gx).max()) < 1e-13, name
        assert impl.quadcost(B, 1e-3)[0] == pytest.approx(Cx, rel=1e-13), name
    ts.close()


def _mps_with_dims(dims, seed):
    """random weight MPS with the given bond dimensions d_0 = 1, d_1, ..., d_N = 1 (Label index on site N/2), any shapes"""
    rng = np.random.default_rng(seed)
    N = len(dims) - 1
    W = []
    for j in range(1, N + 1):
        ml, mr = dims[j - 1], dims[j]
        shape = (ml, 2, mr, 10) if j == N // 2 else (ml, 2, mr)
        A = rng.standard_normal(shape) / np.sqrt(2. * max(ml, mr) * (10 if j == N // 2 else 1))
        A[:, 0] += (np.eye(ml, mr) if A.ndim == 3 else np.eye(ml, mr)[:, :, None] / np.sqrt(10.))
        W.append(A)
    return W


@pytest.mark.parametrize("dtype,ftol,gtol,boost", [("f64", 1e-10, 1e-8, 200.0), ("f64", 1e-10, 1e-8, 1.0), ("f64_e32", 5e-6, 5e-6, 200.0), ("f32", 2e-5, 5e-4, 200.0)])   # boost 1: the reference's own feature map [1, byte/260100]
@pytest.mark.parametrize("dims", [[1, 2, 3, 5, 9, 17, 33, 65, 120, 2, 1],
                                  [1, 2, 120, 97, 64, 60, 61, 128, 33, 2, 1],
                                  [1, 2, 4, 150, 129, 200, 300, 257, 16, 2, 1]])
def test_bonds_with_unequal_and_odd_dimensions(dims, dtype, ftol, gtol, boost):
    """real sweeps leave bonds of every size (minm <= m <= maxm, left and right dimension different, odd): every kernel dispatch on
    (Kp, Np) -- feature / gradient GEMM tile classes, label-dot variants, pack / unpack, the shift forms -- must agree with the
    oracle on shapes it was not tuned for.  Walks a 10-site chain with prescribed bond dimensions and checks environments, forward
    map, gradient and cost at every bond."""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates
    N, NT = len(dims) - 1, 40
    pixels, labels, phi, _ = make_problem(N, NT, 2, 5, pixel_boost=boost)
    W = _mps_with_dims(dims, 11)
    ts = TrainStates(labels, N, max(dims), phi=phi, dtype=dtype)
    o = pyoracle.Oracle(phi, labels, W)
    ts.set_mps(W)
    o.init()
    ts.init()
    rng = np.random.default_rng(2)
    exact = dtype == "f64"
    w = ts.classify()[0]                                       # inference (tnml_classify) on the same odd shapes
    wo = np.stack([o.toverlap(i) for i in range(NT)])
    assert _relmax(w, wo) < (1e-11 if exact else 10 * ftol)
    for b in range(1, N):
        ts.setBond(b)
        o.set_bond(b)
        B = o.bond_tensor(b)
        assert B.shape[0] == dims[b - 1] and B.shape[3] == dims[b + 1]
        B = B + 0.1 * np.abs(B).max() * rng.standard_normal(B.shape)
        assert _relmax(ts.forward(B), o.forward(B)) < ftol, b
        assert _relmax(ts.gradient(B), o.gradient(B)) < gtol, b
        Cg, Co = ts.quadcost(B, 1e-3), o.quadcost(B, 1e-3)
        assert Cg[0] == pytest.approx(Co[0], rel=1e-10 if exact else 10 * ftol), b
        assert Cg[3] == Co[3] or not exact, b
        ts.shiftE(b, True)
        o.shiftE(b, True)
        assert _relmax(ts.env(b), o.env(b)) < (1e-10 if exact else ftol), b
    ts.close()


@pytest.mark.parametrize("dims,maxm,minm", [([1, 2, 4, 7, 13, 24, 31, 17, 8, 4, 2, 1], 29, 5),
                                            ([1, 2, 4, 8, 16, 97, 120, 64, 33, 2, 1], 101, 60),
                                            ([1, 2, 4, 8, 16, 150, 129, 200, 64, 4, 2, 1], 161, 80)])     # splits of 258 ... 400 rows on the workgroup cluster
def test_a_sweep_over_bonds_of_unequal_dimensions_in_lockstep(dims, maxm, minm):
    """whole bond updates (CG, split with truncation to an odd maxm, after-SVD cost, environment shift) on a chain whose bond
    dimensions differ left and right and change under the sweep: one full sweep in lockstep with the oracle"""
    from oracle import pyoracle
    from tnml_amd import lib
    from tnml_amd.fixedl import TrainStates
    N, NT = len(dims) - 1, 50
    pixels, labels, phi, _ = make_problem(N, NT, 2, 5, pixel_boost=200.0)
    W = _mps_with_dims(dims, 4)
    ts = TrainStates(labels, N, max(max(dims), maxm), phi=phi)
    o = pyoracle.Oracle(phi, labels, W)
    ts.set_mps(W)
    o.init()
    ts.init()
    b, ha, n = 1, 1, 0
    while ha <= 2:
        r = ts.bond_update(b, ha, maxm, minm, 1e-10, 3, 1e-3, 1e-10)
        o.set_bond(b)
        B, tr = o.cgrad(o.bond_tensor(b), 3, 1e-3, 1e-10)
        newm, te, _ = o.svd_split(B, b, ha, 1e-10, maxm, minm)
        C, lc, cr, nc = o.quadcost(o.bond_tensor(b), 1e-3)
        o.shiftE(b, ha == 1)
        assert r["newm"] == newm, (b, ha)
        assert r["cost"] == pytest.approx(C, rel=1e-8), (b, ha)
        assert r["ncorrect"] == nc, (b, ha)
        np.testing.assert_allclose(r["cg"]["cost"], tr["cost"], rtol=1e-8)
        ts.set_site(b, o.get_site(b))
        ts.set_site(b + 1, o.get_site(b + 1))
        ts.shiftE(b, ha == 1)
        n += 1
        b, ha = lib.sweepnext(b, ha, N)
    assert n == 2 * (N - 1)
    ts.close()


@pytest.mark.parametrize("m", [121, 123, 124, 125, 127, 128, 129, 160, 161, 200, 224, 225, 240, 256, 257, 300, 319, 320, 321, 384, 385, 400, 448, 500, 512])
def test_split_at_every_size_between_the_one_workgroup_kernel_and_the_cluster_limit(m):
    """every Gram side n = 2m from 242 to 1 024 (round 6: 642..1 024, 17..32 workgroups) takes some combination of the cluster tridiagonalisation's workgroup count (4..32), the
    inverse iteration's vectors per workgroup (16 / 8 / 4, with or without byte flags) and the back transformation's reflectors per pass:
    the singular values of a random bond tensor against numpy at the sizes on either side of each switch (n = 256 once fell into a gap
    between two LDS layouts and failed in the middle of a sweep)."""
    from tnml_amd.fixedl import TrainStates
    N, NT, b = 24, 8, 10
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)
    ts = TrainStates(labels, N, m, phi=phi)
    ts.set_mps(W)
    ts.init()
    for bb in range(1, b):
        ts.shiftE(bb, True)
    ts.setBond(b)
    rng = np.random.default_rng(m)
    Bn = rng.standard_normal((m, 2, 2, m)) * np.logspace(0, -9, 2 * m)[rng.permutation(2 * m)].reshape(m, 2, 1, 1)   # graded rows: a spectrum over 9 decades
    ref = (lambda U, S, Vt: (U[:, :m] * S[:m]) @ Vt[:m])(*np.linalg.svd(Bn.reshape(2 * m, 2 * m)))
    sv = np.linalg.svd(Bn.reshape(2 * m, 2 * m), compute_uv=False)
    # above 128 kept columns the Cholesky QR of the basis is a block Gram-Schmidt over <= 128-column blocks (2 blocks up to 256, 3 above);
    # bgs_chol = 0 is the stock dpotrf + dtrsm it replaced -- both at the sizes either side of a block-count switch
    for bgs in ((1, 0) if m in (129, 256, 257, 300, 385) else (1,)):
        ts.set_option("bgs_chol", bgs)
        mg, teg, svg = ts.svd_split(Bn, b, 1, 1e-12, m, m)
        assert mg == m
        np.testing.assert_allclose(svg[:m], sv[:m], rtol=1e-7, atol=1e-9 * sv[0])
        assert _relmax(ts.bond_tensor(b).reshape(2 * m, 2 * m), ref) < 1e-7
        Ab = ts.get_site(b).reshape(2 * m, m)                       # the left factor is an isometry
        assert np.abs(Ab.T @ Ab - np.eye(m)).max() < 1e-11, bgs
    assert ts.svd_stats()["fallbacks"] == 0
    ts.close()


@pytest.mark.parametrize("m,NT", [(150, 48), (300, 24), (124, 40), (125, 40), (128, 40)])     # n = 248 / 250 / 256: either side of the inverse iteration's 16-vector LDS layout
def test_bond_dimension_above_120_splits_on_the_workgroup_cluster(m, NT):
    """maxm > 120 (BASELINE config 5 goes to 300): Gram side n = 2m > 240, beyond the one-workgroup tridiagonalisation;
    the split runs on the multi-workgroup kernel (eigh_mc.hip) with the Cholesky QR on rocSOLVER dpotrf, the GEMMs on the
    generic tiles.  One bond update at m = 150 and at m = 300 (config 5's bond dimension) against the oracle."""
    ts, o = _pair(N=24, NT=NT, m=m, maxm=m)
    _walk(ts, o, 10)                                           # bond 10 of 24: m x m, Label on the right environment (c0 = 12)
    B = o.bond_tensor(10)
    assert B.shape == (m, 2, 2, m)
    assert _relmax(ts.forward(B), o.forward(B)) < 1e-11
    assert _relmax(ts.gradient(B), o.gradient(B)) < 1e-9
    Bn = B + 0.05 * np.random.default_rng(3).standard_normal(B.shape)
    mg, teg, svg = ts.svd_split(Bn, 10, 1, 1e-10, m, m // 2)
    mo, teo, svo = o.svd_split(Bn, 10, 1, 1e-10, m, m // 2)
    assert mg == mo
    np.testing.assert_allclose(svg[:mg], svo[:mo], rtol=1e-7, atol=1e-8 * svo[0])
    assert _relmax(ts.bond_tensor(10), o.bond_tensor(10)) < 1e-8


@pytest.mark.parametrize("dtype,gate_on,gate_off", [("bf16x3", 2e-4, 2e-5), ("bf16", 5e-2, 5e-3)])
def test_gradient_gemm_on_the_bf16_pipe(dtype, gate_on, gate_off):
    """dP*dag(t.v) (fixedL.cc:379,418) in the bf16 study modes (BASELINE config 5: bf16 MFMA bond contraction): k_bgemm_bf16 rounds both
    operands of the gradient GEMM to bf16 (plain: 8 mantissa bits; hi + lo: ~16) and accumulates in fp32; with option bf16_grad = 0
    the fp32 kernel of round 3 runs instead.  All three bond kinds; the gates are the operand precision (a wrong fragment layout gives
    O(1)); the residuals dP that weight the sum come from the mode's own forward pass, so the fp32 kernel's result carries that error too."""
    ts, o = _pair(N=12, NT=300, m=24, dtype=dtype)
    at = 1
    for b, kind in ((3, "Label on RE"), (6, "Label on B"), (9, "Label on LE")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        Go = o.gradient(B)
        ts.set_option("bf16_grad", 1)
        e_on = _relmax(ts.gradient(B), Go)
        ts.set_option("bf16_grad", 0)
        e_off = _relmax(ts.gradient(B), Go)
        print(dtype, kind, "gradient error with the bf16 kernel %.2e, with the fp32 kernel %.2e" % (e_on, e_off))
        assert e_on < gate_on, kind
        assert e_off < gate_off, kind                      # the fp32 gradient kernel; its weights dP come from the mode's forward pass (bf16: 1e-3)
        assert e_on != e_off, kind                         # the two kernels really are different code paths
    ts.close()


@pytest.mark.parametrize("dtype,gate", [("bf16x3", 2e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("m", [24, 61, 150])
def test_forward_gemm_of_the_bf16_modes_with_operands_converted_once(dtype, gate, m):
    """B*t.v (fixedL.cc:318,377,399,416) in the bf16 study modes with k_fgemm_bf16e (round 6: the Label-free environment converted to bf16
    once per bond, the bond vector once per launch, both site features applied to the fp32 accumulators) against the oracle and against
    the round-5 kernel that rounds E * phi while staging (option bf16_once = 0): both within the operand precision of the oracle, close
    to each other, and not the same code path.  Bond dimensions that fill their last reduction chunk / link tile (24 -> 32 of 32, 64 of 64
    unused), that are odd (61) and that need several of each (150); the copy of the environment must follow every shift."""
    NT = 300 if m <= 61 else 130
    ts, o = _pair(N=24, NT=NT, m=m, dtype=dtype)
    at = 1
    rng = np.random.default_rng(m)
    for b, kind in ((8, "Label on RE"), (9, "Label on RE, next bond"), (15, "Label on LE")):
        for bb in range(at, b):
            ts.shiftE(bb, True); o.shiftE(bb, True)
        at = b
        ts.setBond(b); o.set_bond(b)
        B = o.bond_tensor(b)
        B = B + 0.05 * np.abs(B).max() * rng.standard_normal(B.shape)
        Po = o.forward(B)
        ts.set_option("bf16_once", 1)
        P1 = ts.forward(B)
        assert np.array_equal(P1, ts.forward(B)), kind                    # the cached copy of the environment gives the same bits
        ts.set_option("bf16_once", 0)
        P0 = ts.forward(B)
        e1, e0, d = _relmax(P1, Po), _relmax(P0, Po), _relmax(P1, P0)
        print(dtype, m, kind, "forward error converted once %.2e, rounding while staging %.2e, between them %.2e" % (e1, e0, d))
        assert e1 < gate and e0 < gate, kind
        assert 0 < d < 2 * gate, kind
    ts.close()


def test_environments_spill_to_host_memory_and_the_sweep_does_not_notice():
    """the host tier of the environments (the reference's Nbatch / proj_images spill, fixedL.cc:115-120,153,177-178,216,231): with
    env_budget_mb the environment slabs on the device are capped, slabs farthest from the current bond are copied to host memory and
    copied back when setBond / shiftE needs them.  Two sweeps under a budget of FOUR slabs (the chain needs more than twice that without one) give
    bit-identical costs, bond dimensions and site tensors; every environment -- resident or spilled -- still equals the oracle's; the
    test-set pass (its chain buffers come from the same slabs) and a second context without a budget agree."""
    from tnml_amd.fixedl import TrainStates, mldmrg
    from conftest import make_problem
    N, NT, m = 24, 300, 12
    pixels, labels, phi, W = make_problem(N, NT, m, 11, pixel_boost=200.0)
    args = (2, m, m // 2, 1e-10, 3, 1e-3, 1e-10)

    def run(budget_slabs, async_copies=1):
        ts = TrainStates(labels, N, m, phi=phi)
        if budget_slabs:
            ts.set_option("env_budget_mb", 2)                   # 2 MiB = 4.27 slabs of 10 x 12 x 512 doubles: at most four on the device
            ts.set_option("env_async", async_copies)            # 1: evictions and look-ahead copies on a second stream beside the bond update; 0: on the compute stream
        ts.set_mps(W)
        ts.init()
        reps = mldmrg(ts, *args)
        envs = {}
        for j in range(1, N + 1):
            try:
                envs[j] = ts.env(j)
            except Exception:                                   # sites whose environment was never built
                pass
        acc = ts.classify()
        out = dict(cost=[r["cost"] for r in reps], newm=[r["newm"] for r in reps], W=[ts.get_site(j) for j in range(1, N + 1)],
                   envs=envs, acc=acc, stats=ts.env_stats(), dev=ts.device_bytes())
        ts.close()
        return out
    free = run(0)
    assert free["stats"]["spills"] == 0 and free["stats"]["slabs"] >= 8
    for async_copies in (1, 0):
        tight = run(4, async_copies)
        assert tight["stats"]["slabs"] <= 4 and tight["stats"]["spills"] > 20 and tight["stats"]["fetches"] > 20, async_copies
        assert tight["dev"] < free["dev"]
        assert tight["cost"] == free["cost"] and tight["newm"] == free["newm"], async_copies
        for a, b in zip(tight["W"], free["W"]):
            assert np.array_equal(a, b)
        assert tight["envs"].keys() == free["envs"].keys()
        for j in free["envs"]:
            assert np.array_equal(tight["envs"][j], free["envs"][j]), (j, async_copies)
        assert np.array_equal(tight["acc"][0], free["acc"][0]) and np.array_equal(tight["acc"][1], free["acc"][1])


def test_fixedl_driver_with_an_environment_budget_prints_the_same_log(tmp_path):
    """`env_budget_gb` of the C++ driver (the reference keeps what does not fit in its `proj_images` files, fixedL.cc:115-120; here
    host memory): a 25-site chain under a budget of one MiB = eight of the ~16 slabs it needs -- every cost line identical to the run
    with everything resident."""
    import os
    import re
    import subprocess
    from tnml_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, per_label = 25, 20                                      # (idx images are rows x cols: 5 x 5)
    labels = synth.synthetic_labels(10 * per_label, seed=4, per_label=per_label)
    pixels = np.clip(synth.synthetic_images(N, labels, seed=4).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    logs = []
    for budget in ("", "env_budget_gb = 0.001\n"):
        wd = tmp_path / ("run_budget" if budget else "run_resident")      # (a second run in the same directory would resume from the first one's W)
        wd.mkdir()
        inp = wd / "input"
        inp.write_text("input\n{\ndatadir = %s\nNtrain = %d\nNsweep = 2\ncutoff = 1E-10\nmaxm = 6\nminm = 3\nninitial = 3\nlambda = 1E-3\nNpass = 3\nseed = 5\n"
                       "feature_scale = 255\n%s}\n" % (data, per_label, budget))
        run = subprocess.run([os.path.join(root, "tnml_amd", "fixedL"), str(inp)], capture_output=True, text=True, cwd=wd, timeout=300)
        assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
        logs.append(re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", run.stdout))
    assert len(logs[0]) == 2 * 2 * (N - 1) and logs[0] == logs[1]


@pytest.mark.gpu
@pytest.mark.parametrize("pipelined", [False, True])
def test_speculative_split_and_its_roll_back(pipelined):
    """minm >= the columns the split may keep: the new bond dimension does not depend on the spectrum, so tnml_bond_update_begin enqueues
    the split WITHOUT its host synchronisation (option spec_split, default on) -- eigenvalues and check values reach tnml_bond_update_end
    through pinned mirrors, the new site tensors sit in spare buffers until the deferred check has passed.  Same numbers as the
    synchronous split, bit for bit; with the test hook debug_fail_split the k-th speculative split reports a failed check, the bond
    update and the one begun after it are rolled back and repeated (the failed one with the synchronous split), and the sweep still
    follows the oracle."""
    from oracle import pyoracle
    from tnml_amd.fixedl import TrainStates, mldmrg
    from conftest import make_problem
    N, NT, m = 12, 200, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 5, pixel_boost=200.0)
    args = (1, m, m, 1e-10, 3, 1e-3, 1e-10)                            # minm = maxm

    def run(spec, fail=None):
        ts = TrainStates(labels, N, m, phi=phi)
        ts.set_option("spec_split", spec)
        if fail is not None:
            ts.set_option("debug_fail_split", fail)
        ts.set_mps(W)
        ts.init()
        reps = mldmrg(ts, *args, pipelined=pipelined)
        st = ts.svd_stats()
        Wf = ts.get_mps()
        ts.close()
        return reps, st, Wf
    sync, st_sync, W_sync = run(0)
    spec, st_spec, W_spec = run(1)
    assert [r["newm"] for r in spec] == [r["newm"] for r in sync]
    assert [r["cost"] for r in spec] == [r["cost"] for r in sync]     # the same kernels in the same order
    assert [r["ncorrect"] for r in spec] == [r["ncorrect"] for r in sync]
    np.testing.assert_allclose([r["truncerr"] for r in spec], [r["truncerr"] for r in sync], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose([r["cg"]["cost"][0] for r in spec], [r["cg"]["cost"][0] for r in sync], rtol=0, atol=0)
    for a, b in zip(W_spec, W_sync):
        assert np.array_equal(a, b)
    for fail in (0, 4, 17):                                            # first, an interior and the last speculative split of the sweep (the four chain-end splits are 2 x 2: stock solver)
        redo, st_redo, W_redo = run(1, fail)
        assert st_redo["fallbacks"] >= st_spec["fallbacks"] + 1, (fail, st_redo, st_spec)
        assert [r["bond"] for r in redo] == [r["bond"] for r in sync] and [r["newm"] for r in redo] == [r["newm"] for r in sync]
        np.testing.assert_allclose([r["cost"] for r in redo], [r["cost"] for r in sync], rtol=1e-8)
    o = pyoracle.Oracle(phi, labels, W, nthread=2)
    o.init()
    ro = o.mldmrg(*args)
    np.testing.assert_allclose([r["cost"] for r in spec], [r["cost"] for r in ro], rtol=1e-8)
    assert [r["newm"] for r in spec] == [r["newm"] for r in ro] and [r["ncorrect"] for r in spec] == [r["ncorrect"] for r in ro]
