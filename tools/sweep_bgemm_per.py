"""One process, one set of environments: images per slab of the gradient GEMM (option bgemm_per, units of 32 images) against the
time of its launches, two bond updates (eight launches) per point.  python tools/sweep_bgemm_per.py IMAGES p0 p1 p2 ..."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from tnml_amd import lib, synth
from tnml_amd.fixedl import TrainStates
NT = int(sys.argv[1]); pers = [int(x) for x in sys.argv[2:]]
N, maxm = 784, 120
labels = synth.synthetic_labels(NT); pixels = synth.synthetic_images(N, labels)
ts = TrainStates(labels, N, maxm, pixels=pixels, device=0, rank=0, nranks=1, NT_total=NT, dtype="f64")
ts.set_mps(synth.random_mps(N, maxm, seed=1)); ts.init(); ts.synchronize()
b0 = N // 2 + 8 + 80
for bb in range(1, b0):
    ts.shiftE(bb, True)
b, ha = b0, 1
def steps(k):
    global b, ha
    for _ in range(k):
        ts.bond_update(b, ha, maxm, maxm, 1e-10, 4, 1e-3, 1e-10)
        b, ha = lib.sweepnext(b, ha, N)
steps(3)
ts.profile(True)
for p in pers:
    ts.set_option("bgemm_per", p)
    steps(1)
    ts.profile_reset()
    steps(2)
    ts.synchronize()
    pr = ts.profile_read()
    n, ms = pr.get("bgemm", (0, 0.0))
    chunks = (NT + 255) // 256 * 8
    slabs = "default" if p == 0 else str(-(-chunks // p))
    print("per %5d images (%s slabs): %d launches, %.1f us each" % (p * 32, slabs, n, 1e3 * ms / max(n, 1)), flush=True)
