#!/usr/bin/env python3
"""BASELINE config 5 ("N=784, maxm=300, bf16 MFMA bond contraction vs fp32, high-bond-dim tolerance study").

One bond evaluation of the fixedL sweep -- forward map B*t.v (fixedL.cc:318), cost and number correct
(quadcost, :280-344), and the first gradient dP*dag(t.v) (:379) -- in the factorised form of this repo
 at bond dimension m, with the operand storage / matrix-pipe arithmetic varied:

  f64          fp64 storage, fp64 MFMA                      (TNML_F64, the library default)
  f64_e32      fp32-stored environments and features, fp64 MFMA (TNML_F64_E32)
  f32          fp32 storage, fp32 MFMA                       (TNML_F32)
  bf16         bf16 storage, bf16 MFMA with fp32 accumulation
  bf16x2       hi + lo bf16 pairs (16 mantissa bits), three bf16 MFMA passes per GEMM
  bf16+mean    fp32 common mode (mean over images) + bf16 deviation (SURVEY.md hard part H3)

The bf16 variants round the operands to bf16 and multiply in fp32: bf16 x bf16 products are exact in fp32, so
this is the arithmetic of v_mfma_f32_*_bf16 up to the order of the fp32 accumulation.  This is a numerics
study on the device through torch (plumbing), not a kernel of the library; nothing here touches oracle/.

Inputs: synthetic MNIST-shaped images (tnml_amd.synth), the reference feature map [1, byte/260100],
a random weight MPS whose s=0 part is isometric (what bench.py starts from), environments built by the
exact fp64 chain products.  Errors are relative to the f64 column.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[120, 300])
    ap.add_argument("--images", type=int, default=15360)
    ap.add_argument("--sites", type=int, default=40)
    ap.add_argument("--lam", type=float, default=1e-3)
    ap.add_argument("--npass", type=int, default=4)
    ap.add_argument("--device", default="cuda:0", help="cpu runs the same arithmetic (slowly) for a dry run")
    args = ap.parse_args()

    import torch
    from tnml_amd import synth
    if args.device != "cpu" and not torch.cuda.is_available():
        raise SystemExit("study_precision.py needs a HIP device (or --device cpu for a dry run)")
    dev = torch.device(args.device)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    torch.backends.cuda.matmul.allow_tf32 = False
    f64, f32, b16 = torch.float64, torch.float32, torch.bfloat16

    N, NT = args.sites, args.images
    labels = synth.synthetic_labels(NT)
    pix = synth.synthetic_images(784, labels)[:, 372:372 + N]          # N consecutive pixels of the two middle rows
    phi1 = torch.tensor(pix.astype(np.float64) / 255.0 / 255.0 / 4.0, device=dev)      # phi_1 = x/4, phi_0 = 1
    y = torch.zeros(NT, 10, dtype=f64, device=dev)
    y[torch.arange(NT), torch.tensor(labels.astype(np.int64))] = 1.0
    lab = torch.tensor(labels.astype(np.int64), device=dev)

    for m, on_B in [(m, on_B) for m in args.m for on_B in (False, True)]:
        W = [torch.tensor(a, device=dev) for a in synth.random_mps(N, m, seed=1)]
        c0 = N // 2
        b = c0 - 1 if on_B else c0 - 8                                   # bond (b, b+1): Label on the bond tensor / on the right environment
        # exact environments (fixedL.cc:122-157, factorised: E' = (E x phi) * A)
        E = torch.ones(NT, 1, dtype=f64, device=dev)
        for j in range(1, b):
            A = W[j - 1]
            E = E @ A[:, 0, :] + phi1[:, j - 1, None] * (E @ A[:, 1, :])
        R = torch.ones(NT, 1, dtype=f64, device=dev)                    # right chain; gains the Label index at c0
        for j in range(N, b + 1, -1):
            A = W[j - 1]
            if j == c0:
                R = torch.einsum("arl,nr->nal", A[:, 0], R) + phi1[:, j - 1, None, None] * torch.einsum("arl,nr->nal", A[:, 1], R)
            elif R.dim() == 3:
                R = torch.einsum("ar,nrl->nal", A[:, 0], R) + phi1[:, j - 1, None, None] * torch.einsum("ar,nrl->nal", A[:, 1], R)
            else:
                R = R @ A[:, 0, :].T + phi1[:, j - 1, None] * (R @ A[:, 1, :].T)
        if on_B:
            Bt = torch.einsum("asr,rtql->lastq", W[b - 1], W[b])        # bond tensor (10, ml, 2, 2, mr)
            ml, mr = Bt.shape[1], Bt.shape[4]
            M = Bt.permute(0, 1, 2, 4, 3).reshape(10, 2 * ml, 2 * mr).contiguous()
        else:
            Bt = torch.einsum("asr,rtq->astq", W[b - 1], W[b])        # bond tensor (ml, 2, 2, mr)
            ml, mr = Bt.shape[0], Bt.shape[3]
            M = Bt.permute(0, 1, 3, 2).reshape(2 * ml, 2 * mr).contiguous()  # rows (a,s), columns (q,t)
        pI = torch.stack([torch.ones(NT, dtype=f64, device=dev), phi1[:, b - 1]], 1)    # site b
        pO = torch.stack([torch.ones(NT, dtype=f64, device=dev), phi1[:, b]], 1)        # site b+1

        def rnd(x, dt):
            return x.to(dt).to(f32 if dt == b16 else dt)

        def split(x32):                                                  # hi + lo bf16 pair of an fp32 tensor
            hi = x32.to(b16).to(f32)
            return hi, (x32 - hi).to(b16).to(f32)

        def ops(kind):
            """operand storage of `kind`: (E, pI, pO, R) as the GEMMs will see them, and the matmul of that arithmetic"""
            if kind == "f64":
                return E, pI, pO, R
            if kind == "f64_e32":
                return tuple(t.to(f32).to(f64) for t in (E, pI, pO, R))
            if kind == "f32":
                return tuple(t.to(f32) for t in (E, pI, pO, R))
            if kind == "bf16":
                return tuple(rnd(t, b16) for t in (E, pI, pO, R))
            return tuple(t.to(f32) for t in (E, pI, pO, R))           # bf16x2, bf16+mean: split below

        def mm(kind, A_, B_):
            """A_ @ B_ on the matrix pipe of `kind` (operands already in their storage type)"""
            if kind in ("f64", "f64_e32"):
                return A_.to(f64) @ B_.to(f64)
            if kind == "f32":
                return A_.to(f32) @ B_.to(f32)
            if kind == "bf16":
                return rnd(A_.to(f32), b16) @ rnd(B_.to(f32), b16)    # exact products, fp32 accumulation
            Ah, Al = split(A_.to(f32))
            Bh, Bl = split(B_.to(f32))
            return Ah @ Bh + (Ah @ Bl + Al @ Bh)                       # bf16x2

        def forward(kind, Mat):
            """P [NT,10] = Mat*t.v in the storage / arithmetic of `kind` (Mat: fp64 bond matrix, rows (a,s), columns (q,t))"""
            Es, pIs, pOs, Rs = ops(kind)
            if on_B:                                                     # ten GEMMs, one per label slice of the bond tensor
                return torch.stack([forward1(kind, Mat[l], Es, pIs, pOs, Rs[:, :, None])[:, 0] for l in range(10)], 1)
            return forward1(kind, Mat, Es, pIs, pOs, Rs)

        def forward1(kind, Mat, Es, pIs, pOs, Rs):
            if kind == "bf16+mean":
                Em, Rm = Es.mean(0, keepdim=True), Rs.mean(0, keepdim=True)          # fp32 common mode
                Ed, Rd = rnd(Es - Em, b16), rnd(Rs - Rm, b16)                          # bf16 deviation
                Xm = (Em[:, :, None] * pIs[:, None, :]).reshape(NT, 2 * ml)
                Xd = (Ed[:, :, None] * pIs[:, None, :]).reshape(NT, 2 * ml)
                T = (mm("f32", Xm, Mat) + mm("bf16", Xd, Mat)).reshape(NT, mr, 2)
                U = (T * pOs[:, None, :]).sum(2)
                return (torch.einsum("nq,nql->nl", U, Rm.expand(NT, -1, -1)) + torch.einsum("nq,nql->nl", U, Rd)).to(f64)
            X = (Es[:, :, None] * pIs[:, None, :]).reshape(NT, 2 * ml)
            T = mm(kind, X, Mat).reshape(NT, mr, 2)
            U = (T * pOs.to(T.dtype)[:, None, :]).sum(2)
            if kind == "bf16x2":
                Rh, Rl = split(Rs)
                return (torch.einsum("nq,nql->nl", U, Rh) + torch.einsum("nq,nql->nl", U, Rl)).to(f64)
            return torch.einsum("nq,nql->nl", U, Rs.to(U.dtype)).to(f64)

        def gradient(kind, dP):
            """sum_n dP_n * dag(v_n) as a (2ml x 2mr) matrix in the arithmetic of `kind` (the image sum runs on the matrix pipe)"""
            Es, pIs, pOs, Rs = ops(kind)
            if kind == "bf16+mean":
                kind = "bf16x2"                                          # no cheap common-mode form for the transposed product
            X = (Es[:, :, None] * pIs[:, None, :]).reshape(NT, 2 * ml)
            Xt = X.T.contiguous()
            if on_B:
                out = []
                for l in range(10):
                    Z = (Rs.to(f64) * dP[:, l, None]).to(Es.dtype)
                    out.append(mm(kind, Xt, (Z[:, :, None] * pOs[:, None, :]).reshape(NT, 2 * mr)).to(f64))
                return torch.stack(out, 0)
            Z = torch.einsum("nql,nl->nq", Rs.to(f64), dP).to(Es.dtype)
            Y = (Z[:, :, None] * pOs[:, None, :]).reshape(NT, 2 * mr)
            return mm(kind, Xt, Y).to(f64)

        def cgrad(kind, npass=args.npass):
            """the reference's CG on the bond tensor (fixedL.cc:349-445) with the GEMMs in `kind`; vectors and scalars in fp64"""
            Bm = M.clone()
            P = forward(kind, Bm)
            r = gradient(kind, y - P) - args.lam * Bm
            p_ = r.clone()
            rr = float((r * r).sum())
            alphas = []
            for _ in range(npass):
                Pp = forward(kind, p_)
                pAp = float((Pp * Pp).sum()) + args.lam * float((p_ * p_).sum())
                a_ = rr / pAp
                alphas.append(a_)
                Bm = Bm + a_ * p_
                P = forward(kind, Bm)
                nr = gradient(kind, y - P) - args.lam * Bm
                nn = float((nr * nr).sum())
                p_ = nr + (nn / rr) * p_
                rr = nn
            Pex = forward("f64", Bm)                                     # what the optimised bond tensor really achieves
            return alphas, float(((y - Pex) ** 2).sum()) + args.lam * float((Bm * Bm).sum()), Bm

        Pref = forward("f64", M)
        Gref = gradient("f64", y - Pref)
        Cref = float(((y - Pref) ** 2).sum())
        cor_ref = int((Pref.abs().argmax(1) == lab).sum())
        spread = float((Pref - Pref.mean(0, keepdim=True)).abs().max() / Pref.abs().max())
        a_ref, c_ref, B_ref = cgrad("f64")
        upd = float((B_ref - M).norm())
        print("m = %d, Label on %s: bond (%d,%d) of a %d-site chain, %d images, bond matrix %s%d x %d; cost/image %.6f, correct %d; "
              "image-to-image variation of P relative to its size: %.2e" % (m, "the bond tensor" if on_B else "the right environment", b, b + 1, N, NT,
                                                                            "10 x " if on_B else "", 2 * ml, 2 * mr, Cref / NT, cor_ref, spread))
        print("  single evaluation (errors relative to f64)                                  | %d CG passes on the bond tensor (cost before: %.8g)" % (args.npass, Cref + args.lam * float((M * M).sum())))
        print("  %-10s %12s %12s %13s %12s | %12s %12s %14s" % ("arithmetic", "max|dP|/|P|", "cost rel.err", "labels agree", "|dG|/|G|", "alpha_last", "|dB|/|update|", "true cost after"))
        print("  %-10s %12s %12s %13s %12s | %12.5g %12s %14.8g" % ("f64", "-", "-", "-", "-", a_ref[-1], "-", c_ref))
        for kind in ("f64_e32", "f32", "bf16x2", "bf16+mean", "bf16"):
            P = forward(kind, M)
            eP = float((P - Pref).abs().max() / Pref.abs().max())
            eC = abs(float(((y - P) ** 2).sum()) - Cref) / Cref
            agree = float((P.abs().argmax(1) == Pref.abs().argmax(1)).double().mean())
            G = gradient(kind, y - P)
            eG = float((G - Gref).norm() / Gref.norm())
            al, cc, Bv = cgrad(kind)
            print("  %-10s %12.2e %12.2e %12.2f%% %12.2e | %12.5g %12.2e %14.8g" % (kind, eP, eC, 100 * agree, eG, al[-1], float((Bv - B_ref).norm()) / upd, cc))

        # library GEMM rates of the forward shape (hipBLASLt/rocBLAS through torch; NOT this repo's kernels), for scale
        X64 = (E[:, :, None] * pI[:, None, :]).reshape(NT, 2 * ml).contiguous()
        for dt, name in ((f64, "fp64"), (f32, "fp32"), (b16, "bf16")):
            a, bm = X64.to(dt), (M[0] if on_B else M).to(dt)
            for _ in range(3):
                a @ bm
            sync()
            t0 = time.perf_counter()
            for _ in range(20):
                a @ bm
            sync()
            dt_s = (time.perf_counter() - t0) / 20
            print("  library GEMM %s: %d x %d x %d in %.1f us = %.1f TFLOP/s" % (name, NT, 2 * mr, 2 * ml, dt_s * 1e6, 2.0 * NT * 2 * mr * 2 * ml / dt_s / 1e12))
        del W, E, R, Bt, M, Pref, Gref
        if dev.type == "cuda":
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
