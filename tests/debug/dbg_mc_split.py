import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tnml_amd import synth
from tnml_amd.fixedl import TrainStates
variant = sys.argv[1]
N, m, NT = (24 if variant == "n24" else 22), 150, 16
labels = synth.synthetic_labels(NT, seed=1)
ts = TrainStates(labels, N, m, pixels=synth.synthetic_images(N, labels, seed=1))
ts.set_mps(synth.random_mps(N, m, seed=2))
if variant == "init":
    ts.init()
if variant == "small_first":
    ts2 = TrainStates(labels, N, 20, pixels=synth.synthetic_images(N, labels, seed=1))
    ts2.set_mps(synth.random_mps(N, 20, seed=2)); ts2.init()
rng = np.random.default_rng(5)
sv0 = np.exp(-0.05 * np.arange(300))
U0, _ = np.linalg.qr(rng.standard_normal((300, 300)))
V0, _ = np.linalg.qr(rng.standard_normal((300, 300)))
M = (U0 * sv0) @ V0.T
b = 12 if variant == "b12" else 10
mg, te, sv = ts.svd_split(M.reshape(150, 2, 2, 150, order="F"), b, 1, 1e-10, 150, 75)
print(variant, "->", mg, te, sv[:3], ts.svd_stats()["fallbacks"])
