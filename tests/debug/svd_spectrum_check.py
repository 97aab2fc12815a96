import sys, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/tnml_amd') else '.')
import numpy as np
from tnml_amd import synth
from tnml_amd.fixedl import TrainStates
N, m, NT = 16, 6, 16
labels = synth.synthetic_labels(NT, seed=1)
for be in ["0", "2", "1"]:
    os.environ["TNML_SVD_BACKEND"] = be
    ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
    ts.set_mps(synth.random_mps(N, m, seed=2))
    rng = np.random.default_rng(0)
    p = np.array([9.92883076e-01, 6.62180509e-03, 3.19129020e-04, 1.75990183e-04, 4.35142625e-23, 2.90535593e-24, 0, 0])
    U0, _ = np.linalg.qr(rng.standard_normal((8, 8))); V0, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    M = (U0 * np.sqrt(p)) @ V0[:, :8].T
    B = M.reshape(4, 2, 2, 6, order="F")
    for ha in (1, 2):
        mg, te, sv = ts.svd_split(B, 3, ha, 1e-10, 6, 3)
        print("backend", be, "ha", ha, "m", mg, "p", (sv**2/np.sum(sv**2))[:6], ts.svd_stats())
    ts.close()
