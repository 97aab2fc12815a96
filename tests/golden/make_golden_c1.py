"""Generates tests/golden/fixedl_c1.npz -- BASELINE config 1 at its stated shape: fixedL on 14x14 images (N = 196),
100 training images per label (1000 images), maxm = 10, Nsweep = 2, Npass = 4, lambda = 1e-3, cutoff = 1e-10,
minm = max(10, maxm/2) = 10 (fixedL.cc:593), features [1, x/4] with x in [0, 1] (README.md:74; feature_scale = 255,
SURVEY.md 9-Q1 -- with the reference's double normalisation the sweep is too ill-conditioned to pin to more than 1e-5).

Produced by the CPU oracle (oracle/fixedl_oracle.c; the reference itself needs ITensor v2, absent offline).  Inputs
(pixels, labels, initial W) and expected outputs (per-bond cost / bond dimension / #correct / truncation error over the
780 bond updates, the CG cost trace of the first bond, and after the two sweeps the outputs W_l(x_n) of toverlap
(util.h:19-40) with the predicted labels argmax_l |W_l| of every training image).  Regenerate with
    python tests/golden/make_golden_c1.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import pyoracle  # noqa: E402
from tnml_amd import synth  # noqa: E402

N, PER_LABEL, M, SEED, FEATURE_SCALE = 196, 100, 10, 20160519, 255.0
PARAMS = dict(nsweep=2, maxm=10, minm=10, cutoff=1e-10, npass=4, lam=1e-3, cconv=1e-10)


def features(pixels):
    phi = synth.features_series(pixels)
    phi[..., 1] *= FEATURE_SCALE
    return phi


def main():
    NT = 10 * PER_LABEL
    labels = synth.synthetic_labels(NT, seed=SEED, per_label=PER_LABEL)
    pixels = synth.synthetic_images(N, labels, seed=SEED)
    phi = features(pixels)
    W = synth.random_mps(N, M, seed=1)
    o = pyoracle.Oracle(phi, labels, W, nthread=min(8, os.cpu_count() or 1))
    o.init()
    t0 = time.time()
    reps = o.mldmrg(PARAMS["nsweep"], PARAMS["maxm"], PARAMS["minm"], PARAMS["cutoff"], PARAMS["npass"], PARAMS["lam"], PARAMS["cconv"])
    print("oracle: %d bond updates in %.1f s" % (len(reps), time.time() - t0))
    w = np.stack([o.toverlap(i) for i in range(NT)])
    pred = np.abs(w).argmax(axis=1).astype(np.int32)            # first maximum, util.h:42-57
    srt = np.sort(np.abs(w), axis=1)
    out = dict(
        N=N, NT=NT, M=M, feature_scale=FEATURE_SCALE, pixels=pixels, labels=labels,
        **{"W%03d" % j: A for j, A in enumerate(W, start=1)},
        bond=np.array([r["bond"] for r in reps]), half=np.array([r["half"] for r in reps]),
        newm=np.array([r["newm"] for r in reps]), cost=np.array([r["cost"] for r in reps]),
        ncorrect=np.array([r["ncorrect"] for r in reps]), truncerr=np.array([r["truncerr"] for r in reps]),
        cg_cost0=np.array(reps[0]["cg"]["cost"][:PARAMS["npass"] - 1]),
        weights=w, pred=pred, margin=(srt[:, -1] - srt[:, -2]),
        **{"param_" + k: v for k, v in PARAMS.items()},
    )
    np.savez_compressed(os.path.join(HERE, "fixedl_c1.npz"), **out)
    print("cost/NT first, end of sweep 1, last:", reps[0]["cost"] / NT, reps[389]["cost"] / NT, reps[-1]["cost"] / NT)
    print("train accuracy after 2 sweeps: %.2f %%, smallest decision margin %.3e" % (100.0 * (pred == labels).mean(), out["margin"].min()))


if __name__ == "__main__":
    main()
