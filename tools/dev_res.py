"""A/B of the resident-operand kernels of kernels_res.hip against the generic pair they replace, by bond dimension: the forward pass B*t.v
(k_fwd_res + k_pfinish against k_fgemm64 + k_labeldot) and the Label-carrying environment shift (k_shift_res against k_fgemm64's shift form)
on one bond of a 20-site chain -- same inputs, max relative difference, event-timed mean launch times.
  python tools/dev_res.py [images=60000] [reps=20] [m=120]"""
import os
import sys

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
    for bb in range(1, 12):
        ts.shiftE(bb, True)
    ts.setBond(12)
    B = ts.bond_tensor(12)
    rng = np.random.default_rng(3)
    B = B + 0.05 * rng.standard_normal(B.shape)
    NTp = NT + (-NT) % 256
    fl = 2.0 * NTp * (2 * m) * (2 * m)
    by = 8.0 * NTp * (11 * m + 4)
    res = {}
    if os.environ.get("TNML_DEV_PACE"):
        ts.set_option("res_pace", int(os.environ["TNML_DEV_PACE"]))
    for mode in (0, 2):
        ts.set_option("fwd_res", mode)
        ts.set_option("fused_fwd", 0)
        P = ts.forward(B)
        ts.profile(True, only="fwd_res,fgemm_fwd,labeldot,p_update")
        ts.profile_reset()
        for _ in range(reps):
            ts.forward(B)
        ts.synchronize()
        ts.profile(False)
        pr = ts.profile_read()
        us = sum(1e3 * v[1] / reps for v in pr.values())
        res[mode] = (P, us, {k: (v[0] // reps, round(1e3 * v[1] / max(v[0], 1), 1)) for k, v in pr.items() if v[0]})
    print("m %d, images %d: forward max |P_res - P_generic| / max |P| = %.3e" % (m, NT, np.abs(res[2][0] - res[0][0]).max() / np.abs(res[0][0]).max()))
    for mode, name in ((0, "k_fgemm64 + k_labeldot"), (2, "k_fwd_res + k_pfinish")):
        us = res[mode][1]
        print("  %-24s %8.1f us per forward pass = %.3f of the fp64 MFMA peak, %.2f TB/s of its %.0f MB   %s" % (
            name, us, fl / us / 1e6 / 78.6, by / us / 1e6, by / 1e6, res[mode][2]))
    # Label-carrying shift: the left environment of site 12 from the one of site 11 (m x m)
    fls = 2.0 * NTp * 10 * (2 * m) * m
    bys = 8.0 * NTp * (20 * m + 2)
    for mode, name in ((0, "k_fgemm64 (shift form)"), (2, "k_shift_res")):
        ts.set_option("shift_res", mode)
        ts.shiftE(12, True)
        E = ts.env(12) if NT <= 4096 else None
        ts.profile(True, only="fgemm_shift")
        ts.profile_reset()
        for _ in range(reps):
            ts.shiftE(12, True)
        ts.synchronize()
        ts.profile(False)
        pr = ts.profile_read()["fgemm_shift"]
        us = 1e3 * pr[1] / pr[0]
        print("  %-24s %8.1f us per Label-carrying shift = %.3f of the fp64 MFMA peak, %.2f TB/s of its %.0f MB" % (name, us, fls / us / 1e6 / 78.6, bys / us / 1e6, bys / 1e6))
        res["s%d" % mode] = E
    if res["s0"] is not None:
        print("  shift max |E_res - E_generic| / max |E| = %.3e" % (np.abs(res["s2"] - res["s0"]).max() / np.abs(res["s0"]).max()))
    ts.close()


if __name__ == "__main__":
    main()
