import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import make_problem
from tnml_amd.fixedl import TrainStates
import os
N, NT, m = 20, int(os.environ.get("DBG_NT", "300")), 120
pixels, labels, phi, W = make_problem(N, NT, m, 7, pixel_boost=200.0)
ts = TrainStates(labels, N, m, phi=phi)
ts.set_mps(W); ts.init()
for bb in range(1, 8): ts.shiftE(bb, True)
ts.setBond(8)
B = ts.bond_tensor(8) + 0.05 * np.random.default_rng(1).standard_normal((120, 2, 2, 120))
G0 = ts.gradient(B)
ts.set_option("bgemm_ps", 2)
G1 = ts.gradient(B)
# G[a,s,t,q] -> M layout rows (a,s) cols (q,t)
D = np.abs(G1 - G0)
print("max err", D.max(), "max G", np.abs(G0).max())
# error by a-range and q-range
for a0 in range(0, 120, 30):
    print("a", a0, [float("%.2e" % D[a0:a0+30, :, :, q0:q0+30].max()) for q0 in range(0, 120, 30)])
print("by s,t:", [[float("%.2e" % D[:, s, t, :].max()) for t in range(2)] for s in range(2)])
r = G1 / np.where(np.abs(G0) > 1e-9 * np.abs(G0).max(), G0, np.nan)
print("ratio quantiles", np.nanquantile(r, [0.01, 0.25, 0.5, 0.75, 0.99]))
bad = np.argwhere(D > 1e-9 * np.abs(G0).max())
print("bad entries:", len(bad), "of", D.size)
import collections
print("a values:", sorted(collections.Counter(bad[:,0]).items())[:40])
print("q values:", sorted(collections.Counter(bad[:,3]).items()))
print("(s,t):", collections.Counter(map(tuple, bad[:,1:3])))
G2 = ts.gradient(B)
print("repeat identical:", np.array_equal(G1, G2), np.abs(G2-G0).max())
