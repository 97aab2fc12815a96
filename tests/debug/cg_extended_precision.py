"""Which of two fp64 CG traces is "right" on a badly conditioned bond?  The reference's CG (fixedL.cc:349-445) on one Label-on-B bond of a
4-site chain, run four ways: the C oracle, the numpy restatement in float64, the same numpy code in 80-bit extended precision
(np.longdouble: 64-bit mantissa), and the HIP path.  If extended precision reproduces neither fp64 trace in the late step sizes, the
difference between oracle and HIP path is conditioning, not a defect of either.   python tests/debug/cg_extended_precision.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from oracle import pyoracle
from oracle import np_restatement as npr
from tnml_amd.fixedl import TrainStates
from test_gpu_parity import _mps_with_dims

def cg_np(phi, labels, W, b, npass, lam, dtype):
    n = npr.NpFixedL(phi, labels, W)
    n.phi = n.phi.astype(dtype); n.W = [None] + [a.astype(dtype) for a in n.W[1:]]; n.delta = n.delta.astype(dtype)
    n.init()
    for bb in range(1, b):
        n.shiftE(bb, True)
    n.set_bond(b)
    B, tr = n.cgrad(n.bond_tensor(b).astype(dtype), npass, dtype(lam), 0.0)
    return [float(x) for x in tr["alpha"]], [float(x) for x in tr["rnorm"]]

for (N, dims, NT, boost, lam, seed) in ((4, [1, 2, 2, 1, 1], 257, 200.0, 1e-2, 7), (6, [1, 2, 2, 4, 2, 1, 1], 100, 30.0, 1e-2, 13)):
    pixels, labels, phi, _ = make_problem(N, NT, 2, seed, pixel_boost=boost)
    W = _mps_with_dims(dims, 100 + seed)
    for b in (1, 2):
        o = pyoracle.Oracle(phi, labels, W); o.init()
        ts = TrainStates(labels, N, max(dims), phi=phi); ts.set_mps(W); ts.init()
        for bb in range(1, b):
            o.shiftE(bb, True); ts.shiftE(bb, True)
        o.set_bond(b); ts.setBond(b)
        B0 = o.bond_tensor(b)
        _, to = o.cgrad(B0, 4, lam, 0.0)
        _, tg = ts.cgrad(B0, 4, lam, 0.0)
        a64, r64 = cg_np(phi, labels, W, b, 4, lam, np.float64)
        a80, r80 = cg_np(phi, labels, W, b, 4, lam, np.longdouble)
        f = lambda v: ["%.9g" % x for x in v]
        print("N=%d bond %d (NT=%d, feature scale %g, lambda=%g)" % (N, b, NT, boost, lam))
        print("   alpha  C oracle fp64      ", f(to["alpha"]))
        print("   alpha  numpy fp64         ", f(a64))
        print("   alpha  numpy 80-bit       ", f(a80))
        print("   alpha  HIP fp64           ", f(tg["alpha"]))
        print("   |r|    C oracle / numpy64 / numpy80 / HIP:", f(to["rnorm"]), f(r64), f(r80), f(tg["rnorm"]))
        ts.close()
