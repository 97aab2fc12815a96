"""Time of one split (tnml_svd_split: Gram matrix, tridiagonalisation, tridiagonal eigenproblem, back transformation, orthogonalisation,
the two new site tensors) by bond dimension, in-house path against stock rocsolver_dsyevd, on a bond tensor of numerical rank ~ m + 10
(a site-tensor product plus a small full-rank correction: what a CG-updated bond tensor looks like) and on a full-rank one.
  python tools/time_split.py m [m ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tnml_amd import synth
    from tnml_amd.fixedl import TrainStates
    for m in [int(a) for a in sys.argv[1:]] or [400]:
        N, NT, b = 24, 8, 10
        labels = synth.synthetic_labels(NT, seed=1)
        pixels = synth.synthetic_images(N, labels, seed=1)
        rng = np.random.default_rng(m)
        U, _ = np.linalg.qr(rng.standard_normal((2 * m, m)))
        V, _ = np.linalg.qr(rng.standard_normal((2 * m, m)))
        low = (U * np.logspace(0, -3, m)) @ V.T + 1e-9 * rng.standard_normal((2 * m, 2 * m))
        full = rng.standard_normal((2 * m, 2 * m)) / np.sqrt(2 * m)
        for backend, name in ((0, "in-house"), (1, "rocsolver_dsyevd")):
            if os.environ.get("TNML_SPLIT_BACKEND") not in (None, str(backend)):
                continue
            ts = TrainStates(labels, N, m, pixels=pixels, svd_backend=backend)
            ts.set_mps(synth.random_mps(N, m, seed=2))
            out = []
            for tag, M in (("rank m + noise", low), ("full rank", full)):
                B = M.reshape(m, 2, 2, m, order="F")
                for _ in range(3):
                    ts.svd_split(B, b, 1, 0.0, m, m)
                ts.synchronize()
                t0 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    mg, te, sv = ts.svd_split(B, b, 1, 0.0, m, m)
                ts.synchronize()
                dt = (time.perf_counter() - t0) / reps
                ref = np.linalg.svd(M, compute_uv=False)
                out.append("%s %.2f ms (max rel dev of sigma^2 %.1e)" % (tag, 1e3 * dt, np.abs(sv[:m] ** 2 - ref[:m] ** 2).max() / ref[0] ** 2))
            print("m %d (n = %d) %-18s %s; fallbacks %d" % (m, 2 * m, name, "; ".join(out), ts.svd_stats()["fallbacks"]))
            ts.close()


if __name__ == "__main__":
    main()
