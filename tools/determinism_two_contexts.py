"""Two contexts with the same data run the same bond updates (24-site chain, m = 120, the kernels of BASELINE config 3): the site tensors must
agree bit for bit after every bond update (replicas of W on the ranks of a multi-GPU run rely on it).  Prints the first disagreement."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tnml_amd import lib, synth
from tnml_amd.fixedl import TrainStates
N, m, NT = 24, 120, 1024
labels = synth.synthetic_labels(NT); pixels = synth.synthetic_images(N, labels)
W0 = synth.random_mps(N, m, seed=1)
opts = [tuple(a.split("=")) for a in sys.argv[1:]]
def make():
    ts = TrainStates(labels, N, m, pixels=pixels, device=0, rank=0, nranks=1, NT_total=NT, dtype="f64")
    for k, v in opts:
        ts.set_option(k, int(v))
    ts.set_mps(W0); ts.init()
    for bb in range(1, 8):
        ts.shiftE(bb, True)
    return ts
A, B = make(), make()
b, ha = 8, 1
bad = 0
for step in range(10):
    ra = A.bond_update(b, ha, m, m, 1e-10, 4, 1e-3, 1e-10)
    rb = B.bond_update(b, ha, m, m, 1e-10, 4, 1e-3, 1e-10)
    Wa, Wb = A.get_mps(), B.get_mps()
    d = [float(np.max(np.abs(np.asarray(x) - np.asarray(y)))) for x, y in zip(Wa, Wb)]
    same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(Wa, Wb))
    print("bond %d ha %d: cost %.12g / %.12g  W identical: %s  max |dW| %.2e at site %d  stats %s" % (b, ha, ra["cost"], rb["cost"], same, max(d), int(np.argmax(d)) + 1, A.svd_stats()), flush=True)
    bad += not same
    b, ha = lib.sweepnext(b, ha, N)
print("options", opts, "->", "DETERMINISTIC" if not bad else "%d of 10 bond updates differ" % bad)
