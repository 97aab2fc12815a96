"""Golden-vector tests: tests/golden/fixedl_small.npz (made by tests/golden/make_golden.py).
CPU: the oracle reproduces its committed vectors (guards the oracle against regressions).
GPU: the HIP path reproduces them (same bar as the oracle parity tests)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixedl_small.npz")


def _load():
    g = np.load(GOLD)
    from tnml_amd import synth
    N = int(g["N"])
    phi = synth.features_series(g["pixels"])
    phi[..., 1] *= float(g["boost"])
    W = [g["W%02d" % j] for j in range(1, N + 1)]
    params = {k[6:]: g[k].item() for k in g.files if k.startswith("param_")}
    return g, phi, W, params


def test_oracle_reproduces_golden():
    from oracle import pyoracle
    g, phi, W, p = _load()
    o = pyoracle.Oracle(phi, g["labels"], W)
    o.init()
    np.testing.assert_allclose(o.env(3), g["env3"], rtol=1e-12)
    np.testing.assert_allclose(o.forward(g["B1"]), g["P1"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(o.gradient(g["B1"]), g["G1"], rtol=1e-10, atol=1e-13)
    C, lc, cr, nc = o.quadcost(g["B1"], p["lam"])
    assert C == pytest.approx(float(g["C1"]), rel=1e-12) and nc == int(g["ncorrect1"])
    reps = o.mldmrg(p["nsweep"], p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"])
    assert [r["bond"] for r in reps] == list(g["bond"]) and [r["half"] for r in reps] == list(g["half"])
    assert [r["newm"] for r in reps] == list(g["newm"])                       # exact
    assert [r["ncorrect"] for r in reps] == list(g["ncorrect"])               # exact
    np.testing.assert_allclose([r["cost"] for r in reps], g["cost"], rtol=1e-10)
    np.testing.assert_allclose(np.stack([r["label_cost"] for r in reps]), g["label_cost"], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    from tnml_amd.fixedl import TrainStates, mldmrg
    g, phi, W, p = _load()
    ts = TrainStates(g["labels"], int(g["N"]), p["maxm"], phi=phi)
    ts.set_mps(W)
    ts.init()

    def rel(a, b):
        return np.abs(a - b).max() / np.abs(b).max()
    assert rel(ts.env(3), g["env3"]) < 1e-12
    assert rel(ts.forward(g["B1"]), g["P1"]) < 1e-11
    assert rel(ts.gradient(g["B1"]), g["G1"]) < 1e-9
    C, lc, cr, nc = ts.quadcost(g["B1"], p["lam"])
    assert C == pytest.approx(float(g["C1"]), rel=1e-11) and nc == int(g["ncorrect1"])
    reps = mldmrg(ts, p["nsweep"], p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"])
    assert [r["newm"] for r in reps] == list(g["newm"])
    np.testing.assert_allclose([r["cost"] for r in reps], g["cost"], rtol=1e-8)
    assert [r["ncorrect"] for r in reps] == list(g["ncorrect"])
    np.testing.assert_allclose(np.stack([r["label_cost"] for r in reps]), g["label_cost"], rtol=1e-7, atol=1e-8 * float(g["cost"].max()))
    np.testing.assert_allclose(reps[0]["cg"]["cost"], g["cg_cost"][0], rtol=1e-9)


# ---- per-label variant: tests/golden/single_small.npz (tests/golden/make_golden_single.py) ----------------
GOLD_S = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "single_small.npz")


def _load_single():
    g = np.load(GOLD_S)
    from oracle import np_restatement as npr
    phi = npr.features_single(g["pixels"], True)
    phi[..., 1] *= float(g["boost"])
    W = [g["W%02d" % j] for j in range(1, int(g["N"]) + 1)]
    params = {k[6:]: g[k].item() for k in g.files if k.startswith("param_")}
    return g, phi, W, params


def test_single_oracle_reproduces_golden():
    from oracle import pyoracle
    g, phi, W, p = _load_single()
    o = pyoracle.SingleOracle(phi, g["labels"], int(g["target"]), W)
    o.init()
    np.testing.assert_allclose(o.forward(g["B1"]), g["P1"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(o.gradient(g["B1"]), g["G1"], rtol=1e-10, atol=1e-13)
    assert o.quadcost(g["B1"], p["lam"])[0] == pytest.approx(float(g["C1"]), rel=1e-12)
    reps = o.mldmrg(p["nsweep"], p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"])
    assert [r["c"] for r in reps] == list(g["c"]) and [r["newm"] for r in reps] == list(g["newm"])
    for k in ("cost_old", "cost_cg", "cost"):
        np.testing.assert_allclose([r[k] for r in reps], g[k], rtol=1e-10)
    np.testing.assert_allclose([o.output(i) for i in range(int(g["NT"]))], g["f_final"], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_hip_path_reproduces_single_golden():
    from tnml_amd.fixedl import TrainStates, mldmrg
    g, phi, W, p = _load_single()
    ts = TrainStates(g["labels"], int(g["N"]), p["maxm"], phi=phi, single_label=int(g["target"]))
    ts.set_mps(W)
    ts.init()

    def rel(a, b):
        return np.abs(a - b).max() / np.abs(b).max()
    assert rel(ts.forward(g["B1"]), g["P1"]) < 1e-11
    assert rel(ts.gradient(g["B1"]), g["G1"]) < 1e-9
    assert ts.quadcost(g["B1"], p["lam"])[0] == pytest.approx(float(g["C1"]), rel=1e-11)
    reps = mldmrg(ts, p["nsweep"], p["maxm"], p["minm"], p["cutoff"], p["npass"], p["lam"], p["cconv"])
    assert [r["c"] for r in reps] == list(g["c"]) and [r["newm"] for r in reps] == list(g["newm"])
    for k in ("cost_old", "cost_cg", "cost"):
        np.testing.assert_allclose([r[k] for r in reps], g[k], rtol=1e-7)
    w = ts.classify()[0]
    assert rel(w[:, 0], g["f_final"]) < 1e-6
