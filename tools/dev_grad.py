"""A/B of the gradient GEMM kernels on one bond of the BASELINE config 3 shape (m = 120, Label on RE): k_bgemm64 (grad_quad = 0) against
k_grad_quad (grad_quad = 2) -- same inputs, max relative difference of the two gradients, bit-identical repeats, and the event-timed
mean launch time of class 'bgemm' (+ 'slab_reduce') over `reps` evaluations.
  python tools/dev_grad.py [images=60000] [reps=20] [m=120]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    NT = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    N, m = 20, int(sys.argv[3]) if len(sys.argv) > 3 else 120
    labels = synth.synthetic_labels(NT)
    pixels = synth.synthetic_images(N, labels)
    ts = TrainStates(labels, N, m, pixels=pixels, device=0, dtype="f64")
    ts.set_mps(synth.random_mps(N, m, seed=1))
    ts.init()
    for bb in range(1, 8):
        ts.shiftE(bb, True)
    ts.setBond(8)
    B = ts.bond_tensor(8)
    rng = np.random.default_rng(3)
    B = B + 0.05 * rng.standard_normal(B.shape)
    out = {}
    modes = (0, 2) + ((3, 5) if os.environ.get("TNML_DEV_ABL") else ())
    if os.environ.get("TNML_DEV_WGS"):
        ts.set_option("bgemm_wgs", int(os.environ["TNML_DEV_WGS"]))      # workgroups the gradient GEMM aims at (slab count)
    for mode in modes:
        ts.set_option("grad_quad", mode)
        G = ts.gradient(B)
        G2 = ts.gradient(B)
        ts.profile(True, only="bgemm,grad_quad,slab_reduce")
        ts.profile_reset()
        t0 = time.time()
        for _ in range(reps):
            ts.gradient(B)
        ts.synchronize()
        dt = time.time() - t0
        ts.profile(False)
        pr = ts.profile_read()
        out[mode] = (G, np.array_equal(G, G2), pr.get("grad_quad") if pr.get("grad_quad", (0, 0))[0] else pr.get("bgemm"), pr.get("slab_reduce"), dt / reps)
    G0, G1 = out[0][0], out[2][0]
    print("m %d, images %d: max |G_quad - G_bgemm64| / max |G| = %.3e   (repeats bit-identical: bgemm64 %s, quad %s)" % (
        m, NT, np.abs(G1 - G0).max() / np.abs(G0).max(), out[0][1], out[2][1]))
    for mode, name in ((0, "k_bgemm64"), (2, "k_grad_quad"), (3, "quad, no EL loads"), (5, "quad, no staging")):
        if mode not in out:
            continue
        bg, sr = out[mode][2], out[mode][3]
        us = 1e3 * bg[1] / bg[0] if bg and bg[0] else float("nan")
        us_sr = 1e3 * sr[1] / sr[0] if sr and sr[0] else 0.0
        fl = 2.0 * (NT + (-NT) % 256) * (2 * m) * (2 * m)
        print("  %-12s %8.1f us per launch (+ slab reduce %.1f us) = %.1f TF = %.3f of the fp64 MFMA peak; host loop %.1f us per evaluation" % (
            name, us, us_sr, fl / us / 1e6, fl / us / 1e6 / 78.6, 1e6 * out[mode][4]))
    ts.close()


if __name__ == "__main__":
    main()
