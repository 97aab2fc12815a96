// svd.hip -- truncated SVD that re-splits the optimised bond tensor (fixedL.cc:519-521) on rocSOLVER.
//
// ITensor v2 computes the SVD of the matricised bond tensor through the eigen-decomposition of
// M M^T; this back-end does the same on the device in fp64:
//   M (nl x nr): rows = (a,s[,l]) indices of site b, columns = (t,beta[,l]) indices of site b+1
//   rho = M M^T or M^T M on the smaller side  (rocBLAS dgemm)
//   rho = Q diag(lambda) Q^T                  (in-house eigensolver, eigh.hip / eigh_mc.hip; stock
//                                              rocSOLVER dsyevd measured 5.9 ms at n=240, dgesvdj 29 ms,
//                                              dgesvd 153 ms -- profiles/r01_probe_*)
//   sigma = sqrt(lambda), truncation rule on the host (tnml_truncate), kept factors by dgemm:
//   the site the sweep leaves gets the orthonormal factor, the site it moves to gets S*V
//   ("W.Aref(c+dc) *= S", fixedL.cc:521).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "tnml_internal.h"

static inline int nblk(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b ? b : 1)); }

// B_it [a][s][t][be][l] with Label on the LEFT site -> M[(a,s,l)][(t,be)]
__global__ void k_perm_labL_fwd(const double* __restrict__ B, double* __restrict__ M, int mL2, int nr) {
    const size_t total = (size_t)mL2 * TNML_NL * nr;
    const int nl = mL2 * TNML_NL;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int il = (int)(idx % nl), ir = (int)(idx / nl);
        const int as = il % mL2, l = il / mL2;
        M[idx] = B[as + (size_t)mL2 * ir + (size_t)mL2 * nr * l];
    }
}
// Lfac[(a,s,l)][g] -> A_b[a][s][g][l]
__global__ void k_perm_labL_back(const double* __restrict__ Lf, double* __restrict__ A, int mL2, int m) {
    const int nl = mL2 * TNML_NL;
    const size_t total = (size_t)nl * m;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int il = (int)(idx % nl), g = (int)(idx / nl);
        const int as = il % mL2, l = il / mL2;
        A[as + (size_t)mL2 * (g + (size_t)m * l)] = Lf[idx];
    }
}
// sigma_g (inv = 0) or 1/sigma_g (inv = 1) of the g-th largest eigenvalue, from the ascending eigenvalues on the device:
// the same IEEE sqrt / division the host applies to its copy, so no scale vector has to travel back
struct SigmaRef { const double* ev; int n; int inv; };
static __device__ __forceinline__ double sigma_of(const SigmaRef& s, int g) {
    double lam = s.ev[s.n - 1 - g];
    if (!(lam > 0.)) lam = 0.;
    const double sg = sqrt(lam);
    return s.inv ? (sg > 1e-300 ? 1.0 / sg : 0.0) : sg;
}
// Q[:, g] = G[:, n-1-g]  (dsyevd returns ascending eigenvalues), optional column scale
__global__ void k_take_top(const double* __restrict__ G, double* __restrict__ Q, int n, int m, const double* __restrict__ colscale) {
    const size_t total = (size_t)n * m;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % n), g = (int)(idx / n);
        double v = G[i + (size_t)n * (n - 1 - g)];
        if (colscale) v *= colscale[g];
        Q[idx] = v;
    }
}
// out[g + m*i] = scale[g] * Q[i + n*g]   (transpose of the kept columns, optional row scale)
__global__ void k_transpose_scale(const double* __restrict__ Q, double* __restrict__ out, int n, int m, SigmaRef rowscale) {
    const size_t total = (size_t)n * m;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % m), i = (int)(idx / m);
        double v = Q[i + (size_t)n * g];
        if (rowscale.ev) v *= sigma_of(rowscale, g);
        out[idx] = v;
    }
}
__global__ void k_scale_rows(double* __restrict__ X, int m, size_t cols, SigmaRef rowscale) {
    const size_t total = (size_t)m * cols;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) X[idx] *= sigma_of(rowscale, (int)(idx % m));
}
__global__ void k_scale_cols(double* __restrict__ X, size_t rows, int m, SigmaRef colscale) {
    const size_t total = rows * m;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) X[idx] *= sigma_of(colscale, (int)(idx / rows));
}

// rocsolver_dpotrf's info -> flags of the Cholesky QR: flag[0] = the factorisation failed (a pivot was not positive), flag[1] = a factorisation ran
__global__ void k_potrf_flags(const int* __restrict__ info, double* __restrict__ flag) {
    if (threadIdx.x == 0) { flag[0] = info[0] != 0 ? 1. : 0.; flag[1] = 1.; flag[-1] = 0.; }      // flag[-1]: the polish step's deviation slot (an atomic max)
}

// flags of the blocks of the block Gram-Schmidt -> the two flags of the Cholesky QR (any block failed / any block factored)
__global__ void k_flags_any(const double* __restrict__ fl, int nb, double* __restrict__ flag) {
    if (threadIdx.x == 0) {
        double f0 = 0., f1 = 0.;
        for (int k = 0; k < nb; ++k) { if (fl[2 * k] != 0.) f0 = 1.; if (fl[2 * k + 1] != 0.) f1 = 1.; }
        flag[0] = f0; flag[1] = f1; flag[-1] = 0.;              // flag[-1]: the polish step's deviation slot (an atomic max)
    }
}

// the workgroup cluster's status word -> a double that can be summed over the ranks (1 = this rank's cluster gave up)
__global__ void k_mc_flag(const unsigned long long* __restrict__ status, double* __restrict__ out) {
    if (threadIdx.x == 0) out[0] = status[0] != 0ull ? 1. : 0.;
}

// C = op(A) op(B) at the sizes of the split: k_dgemm_small (kernels_sgemm.hip) up to 4e7 multiply-adds, rocBLAS as
// `strips` column strips beyond (the Label-on-B bonds reduce over 2400: 133 us as one call, 17 us as 8 strips) or with option small_gemm = 0
int split_gemm(tnml_ctx* c, bool ta, bool tb, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc, int strips, const SmallGemmArgs* chk) {
    if (c->small_gemm && K <= 1024 && (double)M * N * K <= 4.0e7) {   // (tools/probe/probe_sgemm.hip: 8.7-9.9 us against 19 at 240^3, 17 against 25 at 300 x 300 x 600; loses from ~6e7 on)
        SmallGemmArgs g{A, lda, B, ldb, C, ldc, M, N, K, ta ? 1 : 0, tb ? 1 : 0};
        if (chk) { g.chk_src = chk->chk_src; g.chk_host = chk->chk_host; g.chk_bad = chk->chk_bad; }
        return launch_dgemm_small(c, g);
    }
    const rocblas_status st = dgemm_strips(c->blas, ta ? rocblas_operation_transpose : rocblas_operation_none, tb ? rocblas_operation_transpose : rocblas_operation_none,
                                           M, N, K, A, lda, B, ldb, C, ldc, strips);
    if (st != rocblas_status_success) return tnml_fail(c, "split_gemm: rocblas dgemm failed (%d)", (int)st);
    if (chk) return launch_split_check_mirror(c, chk->chk_src, chk->chk_host, chk->chk_bad);
    return 0;
}

#define RBCK(c, expr) do { rocblas_status s_ = (expr); if (s_ != rocblas_status_success) return tnml_fail((c), "%s failed: rocblas status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); } while (0)

// ---- density-matrix split with a noise term (per-label variant, single.h:648-672) ------------------------------------------------
// rho = B B^dag over the indices of site c is the Gram matrix the split forms anyway; the images add
//   drho = sum_n dr_n dr_n^dag,  dr_n = (B * E_n) (x) E_n,  E_n = environment on the outer link of site c,
// i.e. drho[(e1,s1),(e2,s2)] = sum_n w_{s1 s2}[n] E_n[e1] E_n[e2] with the 2 x 2 weights w[n] = T_n T_n^T, T_n = B * E_n as a
// (site index of c) x (indices of the other site) matrix: one GEMM for T over all images, a streaming kernel for the three weights,
// three weighted Gram matrices of the environment (dgemm over the images), summed over the ranks, added to rho.
__global__ void k_noise_weights(const double* __restrict__ T, int NTp, int NT, int ny, size_t stride_s, size_t stride_y, double* __restrict__ w) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= NTp) return;
    double w00 = 0., w01 = 0., w11 = 0.;
    if (n < NT)
        for (int y = 0; y < ny; ++y) {
            const double t0 = T[n + (size_t)NTp * (y * stride_y)], t1 = T[n + (size_t)NTp * (stride_s + y * stride_y)];
            w00 = fma(t0, t0, w00); w01 = fma(t0, t1, w01); w11 = fma(t1, t1, w11);
        }
    w[n] = w00; w[NTp + n] = w01; w[2 * (size_t)NTp + n] = w11;      // images beyond NT (padding) weigh nothing
}
__global__ void k_noise_scale_rows(const double* __restrict__ E, const double* __restrict__ w, double* __restrict__ Es, int mE, int NTp) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)mE * NTp) return;
    Es[i] = E[i] * w[i % NTp];
}
// rho[(e1,s1),(e2,s2)] += noise * blk_{s1 s2}[e1][e2]; row index e + mE s on the left site (ha = 1), s + 2 e on the right site (ha = 2)
__global__ void k_noise_add(double* __restrict__ rho, int n, const double* __restrict__ blk, int mE, int ha, double noise) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)n * n) return;
    const int r1 = (int)(i % n), r2 = (int)(i / n);
    const int e1 = ha == 1 ? r1 % mE : r1 / 2, s1 = ha == 1 ? r1 / mE : r1 % 2;
    const int e2 = ha == 1 ? r2 % mE : r2 / 2, s2 = ha == 1 ? r2 / mE : r2 % 2;
    const int k = s1 + s2;                                           // 0: w00, 1: w01 (= w10), 2: w11
    rho[i] += noise * blk[(size_t)k * mE * mE + e1 + (size_t)mE * e2];
}
__global__ void k_scale_all(double* __restrict__ x, size_t n, double f) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) x[i] *= f;
}
// G (n x n, the Gram matrix over the indices of site c) += noise * drho.  B_it: the bond tensor in ITensor order [a][s][t][be].
static int noise_add(tnml_ctx* c, const double* B_it, int b, int ha, int mL, int mR, double* G, int n) {
    hipStream_t st = c->stream;
    const int cs = ha == 1 ? b : b + 1, envsite = ha == 1 ? cs - 1 : cs + 1;
    const bool have_env = ha == 1 ? cs > 1 : cs < c->N - 1;          // single.h:650,655 as written: "ha == 2 && c < N-1" -- at c = N-1 the reference leaves the environment out
    if (!have_env) {                                                 // chain end: dr_n = B for every image, drho = NT rho
        hipLaunchKernelGGL(k_scale_all, dim3(nblk((size_t)n * n)), dim3(256), 0, st, G, (size_t)n * n, 1. + c->noise * (double)c->cfg.NT_total);
        HIPCK(c, hipGetLastError());
        return 0;
    }
    const int mE = ha == 1 ? mL : mR, mOth = ha == 1 ? mR : mL, NTp = c->NTp;
    if (!c->env[envsite].ptr || c->env[envsite].m != mE || c->env[envsite].L != 1) return tnml_fail(c, "noise split: environment of site %d missing", envsite);
    const double* E = (const double*)c->env[envsite].ptr;             // [mE][NTp], fp64 (checked when the option was set)
    if (!c->noise_ws) TCK(ctx_alloc_doubles(c, &c->noise_ws, (size_t)5 * c->maxm * NTp + (size_t)3 * NTp + (size_t)3 * c->maxm * c->maxm));
    double* T = c->noise_ws;                                         // [NTp][4 mOth]
    double* Es = T + (size_t)4 * c->maxm * NTp;                      // [mE][NTp]
    double* w = Es + (size_t)c->maxm * NTp;                          // [3][NTp]
    double* blk = w + (size_t)3 * NTp;                               // [3][mE][mE]
    const double one = 1.0, zero = 0.0;
    if (ha == 1) RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, NTp, 4 * mOth, mE, &one, E, NTp, B_it, mL, &zero, T, NTp));
    else         RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, NTp, 4 * mOth, mE, &one, E, NTp, B_it, 4 * mL, &zero, T, NTp));
    // columns of T: (s, t, be) = s + 2 (t + 2 be) on the left site; (a, s, t) = (a + mL s) + 2 mL t on the right site
    hipLaunchKernelGGL(k_noise_weights, dim3((NTp + 255) / 256), dim3(256), 0, st, (const double*)T, NTp, c->NT, 2 * mOth,
                       ha == 1 ? (size_t)1 : (size_t)2 * mL, ha == 1 ? (size_t)2 : (size_t)1, w);
    for (int k = 0; k < 3; ++k) {
        hipLaunchKernelGGL(k_noise_scale_rows, dim3(nblk((size_t)mE * NTp)), dim3(256), 0, st, E, (const double*)(w + (size_t)k * NTp), Es, mE, NTp);
        RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, mE, mE, NTp, &one, E, NTp, Es, NTp, &zero, blk + (size_t)k * mE * mE, mE));
    }
    TCK(allreduce_sum(c, blk, (size_t)3 * mE * mE));                 // the sum over the images of all ranks
    hipLaunchKernelGGL(k_noise_add, dim3(nblk((size_t)n * n)), dim3(256), 0, st, G, n, (const double*)blk, mE, ha, c->noise);
    HIPCK(c, hipGetLastError());
    return 0;
}

int svd_split_device(tnml_ctx* c, const double* B_it, int b, int ha, double cutoff, int maxm, int minm,
                     double* truncerr, int* newm, double* sv_host, int* nsv, int spec_slot) {
    ProfScope ps(c, KC_SVD);
    SiteT& Sl = c->W[b];
    SiteT& Sr = c->W[b + 1];
    const int mL = Sl.ml, mR = Sr.mr;
    const bool labL = (c->c0 == b), labR = (c->c0 == b + 1);
    const int nl = 2 * mL * (labL ? TNML_NL : 1), nr = 2 * mR * (labR ? TNML_NL : 1);
    const bool dm = c->single() && c->noise >= 1e-14;    // density-matrix split with a noise term (single.h:648-672): rho lives on site c, whatever its size
    const bool left = dm ? ha == 1 : (nl < nr) || (nl == nr && ha == 1);
    const int n = dm ? (left ? nl : nr) : (nl < nr ? nl : nr);
    if (n > c->svd_n) return tnml_fail(c, "svd_split: matrix side %d exceeds workspace %d (raise maxm)", n, c->svd_n);
    if (maxm < 1 || minm < 0) return tnml_fail(c, "svd_split: maxm must be >= 1 and minm >= 0");
    if (maxm > c->maxm) {                                // the workspaces (sS, sCm, sQ1, sF) are sized by the context's maxm
        char wb[256];
        snprintf(wb, sizeof wb, "svd_split: maxm = %d exceeds the context's maxm = %d (tnml_plan_maxm / tnml_create): the truncation keeps at most %d", maxm, c->maxm, c->maxm);
        c->warn = wb;
        maxm = c->maxm;
    }
    if (minm > maxm) minm = maxm;
    hipStream_t st = c->stream;

    const double* M = B_it;
    if (labL) {
        hipLaunchKernelGGL(k_perm_labL_fwd, dim3(nblk((size_t)nl * nr)), dim3(256), 0, st, B_it, c->sM, 2 * mL, nr);
        M = c->sM;
    }
    const double one = 1.0, zero = 0.0;
    const int gstrips = (left ? nr : nl) >= 1024 ? 8 : 4;     // the Label-on-B bonds reduce over 2400: 133 us as one call, 17 us as 8 strips
    if (left) TCK(split_gemm(c, false, true, nl, nl, nr, M, nl, M, nl, c->sG, nl, gstrips));
    else      TCK(split_gemm(c, true, false, nr, nr, nl, M, nl, M, nl, c->sG, nr, gstrips));
    if (dm) TCK(noise_add(c, B_it, b, ha, mL, mR, c->sG, n));           // rho += noise * drho (single.h:654-665)
    // eigen-decomposition of rho.  backend 0 (default): in-house Householder tridiagonalisation (one workgroup up to n = 240,
    // eigh.hip; a cluster of workgroups up to n = 640, eigh_mc.hip) + in-house bisection / inverse iteration + back
    // transformation, verified, with rocSOLVER as the fallback; 2: in-house tridiagonalisation + rocSOLVER dstedc; 1: stock
    // rocSOLVER dsyevd (5.9 ms at n = 240, 11.7 ms at n = 600: ~9 000 launches of 3-4 us, profiles/r03_prof_m300_before.txt)
    const bool tri = (c->cfg.svd_backend != TNML_SVD_ROCSOLVER) && n >= 3 && (n <= 240 || (n <= eigh_mc_max_n() && c->mc_xbuf));
    bool own_eig = tri && c->cfg.svd_backend == TNML_SVD_SYEVD;
    const bool mc = tri && n > 240;
    const int mk = maxm < n ? maxm : n;                   // the truncation never keeps more than maxm
    const double* evals = c->sD;                          // ascending eigenvalues of rho
    // Speculative form (tnml_bond_update_begin, option spec_split): minm >= mk means the truncation keeps exactly mk columns whatever
    // the spectrum (tnml_truncate stops at its minm test), so nothing the host would read decides anything but the FALLBACK -- and that
    // decision can wait for tnml_bond_update_end.  No eigenvalue broadcast, no copy, no stream synchronisation: eigenvalues and check
    // values reach the host through pinned mirrors written by the kernels themselves, the two site tensors go to spare buffers, and a
    // failed check rolls the bond update back (tnml_abi.hip).  One-workgroup sizes only (the cluster's give-up flag is a collective decision).
    bool spec = spec_slot >= 0 && c->spec_split && !c->force_safe && own_eig && !mc && minm >= mk && !sv_host && c->hrep != nullptr;
    double* hmir = spec ? c->hrep + (size_t)spec_slot * c->hrep_stride : nullptr;     // [n eigenvalues | 4 check values]
    if (spec_slot >= 0) {
        // inside a bond update in flight the two new site tensors ALWAYS go to spare buffers (whichever form the split takes): a later
        // roll-back -- of this bond update, or of the one before it whose check is still pending -- can then restore both sites
        PendingReport& pr = c->pend[spec_slot];
        pr.nundo = 0; pr.spec = false;
        auto& pl = (b == c->c0) ? c->spare_big : c->spare_small;
        auto& pr_ = (b + 1 == c->c0) ? c->spare_big : c->spare_small;
        const bool have = !pl.empty() && !pr_.empty() && !(&pl == &pr_ && pl.size() < 2);     // (always, with two bond updates in flight at most)
        if (have) {
            for (int j = b; j <= b + 1; ++j) {
                SiteT& S = c->W[j];
                auto& pool = (j == c->c0) ? c->spare_big : c->spare_small;
                pr.undo[pr.nundo++] = SiteUndo{j, S.a, S.ml, S.mr};
                S.a = pool.back(); pool.pop_back();
            }
        } else spec = false;
        if (spec) { pr.spec = true; pr.split_n = n; pr.split_mk = mk; c->spec_splits += 1; c->spec_splits_total += 1; }
        else hmir = nullptr;
    }
    if (tri) {
        // sG is a Gram matrix: rank-adaptive exit once the trailing block weighs less than the error G = B^T B carries anyway -- ~4 eps of
        // the trace in the fp64 / fp32 modes; in the bf16 study modes the bond tensor itself carries the relative noise e of the bf16 operands of
        // its gradient (2^-8 plain, ~2^-16 with hi + lo operands), eigenvalues of G below (e / 8)^2 trace(G) are that noise, and chasing them
        // costs the whole chain (config 5, m = 300: 598 steps instead of ~330 -- the bf16 bond update was SLOWER than the fp32 one for it)
        const double exit_tol = c->cfg.dtype == TNML_BF16 ? 2.4e-7 : (c->cfg.dtype == TNML_BF16X3 ? 3.6e-12 : 1e-15);
        TCK(eigh_tridiagonalize(c, c->sG, n, c->sD, c->sE2, c->sTau, c->sV, c->sytrd_exit ? exit_tol : 0.));
        if (own_eig) { TCK(eigh_tridiag_eig(c, c->sD, c->sE2, n, c->sW, mk, c->sC, n, c->sScr, hmir)); evals = c->sW; }
        else RBCK(c, rocsolver_dstedc(c->blas, rocblas_evect_tridiagonal, n, c->sD, c->sE2, c->sC, n, c->sInfo));
    } else {
        RBCK(c, rocsolver_dsyevd(c->blas, rocblas_evect_original, rocblas_fill_upper, n, c->sG, n, c->sD, c->sE, c->sInfo));
    }
    double* Q = c->sF;                                   // kept eigenvectors, n x m
    bool direct_left = false;
    double* Lf = c->sF + (size_t)c->svd_n * c->maxm;     // left factor when a permutation is still needed
    double* Q0 = c->sQ1;
    double* hd = c->h_scal + 2 * c->svd_n + 32;
    double* dv = own_eig ? c->sW + n : c->sDev;           // [0] max|Q^T Q - I| into the polish step, [1] Cholesky failed, [2] a factorisation was needed
    if (own_eig) {
        // Z (already "largest first") -> U = H_0 H_1 ... Z for all mk candidate columns, queued BEFORE the eigenvalues go to
        // the host, so that the truncation decision costs no idle gap on the device; the kept columns are the first m.
        TCK(eigh_backtransform(c, c->sV, c->sTau, n, c->sC, n, Q0, n, mk));
        // The Gram matrix of a bond tensor spans 12+ decades (the common mode of the images dominates), so most of the kept
        // eigenvalues sit within a few 100 eps*|T| of each other: inverse iteration returns the right invariant subspace for
        // them but not orthogonal vectors (80 % of the bond updates of a sweep).  Any orthonormal basis of that subspace is
        // an equally valid set of singular vectors, so the basis ALWAYS goes through a Cholesky QR (Q1 = Q0 R^-1,
        // Q0^T Q0 = R^T R) and one Newton-Schulz polish step whose input deviation is the check -- no host decision, no
        // second synchronisation.
        const double* Qin;
        if (mk <= TNML_CHOL_MAXM) {                       // one-workgroup kernel; returns R = I straight away when Q0 is orthonormal to 5e-7
            TCK(split_gemm(c, true, false, mk, mk, n, Q0, n, Q0, n, c->sS, mk, 1));
            TCK(eigh_chol_rinv(c, c->sS, mk, c->sCm, dv + 1, 1));           // writes both flags, clears dv[0] for the polish step's atomic max
            TCK(split_gemm(c, false, false, n, mk, mk, Q0, n, c->sCm, mk, c->sG, n, 1));   // the Gram matrix is consumed by now
            Qin = c->sG;
        } else if (c->bgs_chol && mk <= 3 * TNML_CHOL_MAXM) {
            // larger bases (128 < mk <= 384; BASELINE config 5 keeps 300 columns): block Gram-Schmidt over column blocks of <= 128 with the
            // one-workgroup Cholesky kernel inside a block.  Block k is projected against the finished blocks before it -- twice
            // (classical Gram-Schmidt loses what the first pass leaves of a cluster that straddles a block boundary; the second pass
            // removes it) -- then Q_k <- Q_k R_k^-1 with Q_k^T Q_k = R_k^T R_k.  Everything is a small GEMM or the 26-110 us kernel:
            // 0.69 ms of stock dpotrf (three potf2 panels) + dtrsm become ~0.4 ms at mk = 300.  The polish step and its check below are
            // the same for every path, so a basis this does not fix still ends in the rocSOLVER fallback.
            const int nb = (mk + TNML_CHOL_MAXM - 1) / TNML_CHOL_MAXM, wb = (mk + nb - 1) / nb;
            double* Cw = c->sCm;                            // coefficients against the finished columns (c0 x w)
            double* Rk = c->sCm + (size_t)(mk - wb) * wb;   // R_k^-1 (w x w); (mk - wb) wb + wb^2 <= maxm^2
            double* fl = c->sScr;                           // two flags per block (the eigen stages are done with the scratch)
            const double mone = -1.0;
            HIPCK(c, hipMemsetAsync(fl, 0, sizeof(double) * 2 * nb, st));        // (a failing block writes its first flag only)
            for (int k = 0, c0 = 0; k < nb; ++k) {
                const int w = std::min(wb, mk - c0);
                double* Qk = Q0 + (size_t)c0 * n;
                if (c0 > 0)
                    for (int pass = 0; pass < 2; ++pass) {
                        RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, c0, w, n, &one, c->sG, n, Qk, n, &zero, Cw, c0));
                        RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, n, w, c0, &mone, c->sG, n, Cw, c0, &one, Qk, n));
                    }
                RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, w, w, n, &one, Qk, n, Qk, n, &zero, c->sS, w));
                TCK(eigh_chol_rinv(c, c->sS, w, Rk, fl + 2 * k));
                RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, n, w, w, &one, Qk, n, Rk, w, &zero, c->sG + (size_t)c0 * n, n));
                c0 += w;
            }
            hipLaunchKernelGGL(k_flags_any, dim3(1), dim3(64), 0, st, (const double*)fl, nb, dv + 1);
            Qin = c->sG;
        } else {                                          // rocSOLVER dpotrf + rocBLAS dtrsm, in place (option bgs_chol = 0, or more than 384 columns)
            RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, mk, mk, n, &one, Q0, n, Q0, n, &zero, c->sS, mk));
            RBCK(c, rocsolver_dpotrf(c->blas, rocblas_fill_upper, mk, c->sS, mk, c->sInfo));
            hipLaunchKernelGGL(k_potrf_flags, dim3(1), dim3(64), 0, st, (const int*)c->sInfo, dv + 1);
            RBCK(c, rocblas_dtrsm(c->blas, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, n, mk, &one, c->sS, mk, Q0, n));
            Qin = Q0;
        }
        TCK(split_gemm(c, true, false, mk, mk, n, Qin, n, Qin, n, c->sS, mk, 1));
        // Newton-Schulz step Q <- Q (1.5 I - 0.5 Q^T Q); d = max|Q^T Q - I| before the step is checked on the host (after it: ~0.75 d^2)
        // the kept basis lands where it is needed: straight in the site tensor when that is its final place
        direct_left = left && !labL && mk <= c->maxm;
        if (direct_left) Q = Sl.a;
        if (c->small_gemm) {                              // the factor 1.5 I - 0.5 S is formed while the product loads S; d by an atomic max into dv[0]
            SmallGemmArgs g{Qin, n, c->sS, mk, Q, n, n, mk, mk, 0, 0, 1, dv};
            TCK(launch_dgemm_small(c, g));
        } else {
            TCK(eigh_ns_matrix(c, c->sS, c->sCm, mk, dv));
            RBCK(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, n, mk, mk, &one, Qin, n, c->sCm, mk, &zero, Q, n));
        }
    }
    int m = mk;
    bool stock = !tri;                                   // eigenvectors of rho itself in sG (dsyevd)
    double* h = c->h_scal;          // pinned, capacity >= 2*svd_n + 64
    SmallGemmArgs chk{};                                   // the check-value side job of the speculative form (rides in the factor product below)
    const SmallGemmArgs* chkp = nullptr;
    if (spec) {
        // m = mk; the eigenvalues are on their way to hmir[0..n) (k_teig_vectors), the check values follow with the factor product
        chk.chk_src = dv; chk.chk_host = hmir + n; chk.chk_bad = c->tail + TNML_SPECSLOT;
        // test hook (option debug_fail_split): the k-th speculative split reports a failed check -- by spoiling the check VALUE in stream
        // order before the product that mirrors it, so that no product kernel carries a test switch
        if (c->debug_fail_split >= 0 && c->spec_splits - 1 == c->debug_fail_split) TCK(launch_fill_f64(c, dv + 1, 1.0, 1));
        chkp = &chk;
        if (truncerr) *truncerr = 0.;                      // tnml_bond_update_end computes it from the mirrored eigenvalues
        if (newm) *newm = m;
        if (nsv) *nsv = n;
    } else {
    // eigenvalues -> host: the truncation decision (ITensor truncate()) fixes the new bond dimension.  With more than
    // one rank the decision is made collective: every rank decides on rank 0's eigenvalues (and rank 0's orthogonality
    // check), so that a last-bit difference between replicas can never produce different bond dimensions.
    const int nev = own_eig ? n + 4 : n;                 // the check values ride behind the eigenvalues: one broadcast, one copy
    TCK(bcast_rank0(c, const_cast<double*>(evals), nev));
    // The workgroup cluster of ANY rank may have given up (a workgroup that never got a CU): the fallback below contains
    // collectives, so every rank has to take it together -- the status words are summed over the ranks (one 8-byte all-reduce,
    // splits above n = 240 with a communicator only) and every rank decides on the sum.
    double* mcflag = c->sDev + 3;
    if (mc) {
        hipLaunchKernelGGL(k_mc_flag, dim3(1), dim3(64), 0, st, static_cast<const unsigned long long*>(eigh_mc_status_ptr(c->mc_xbuf)), mcflag);
        TCK(allreduce_sum(c, mcflag, 1));
    }
    HIPCK(c, hipMemcpyAsync(h, evals, sizeof(double) * nev, hipMemcpyDeviceToHost, st));
    double* h_mc = h + n + 8;
    *h_mc = 0.;
    if (mc) HIPCK(c, hipMemcpyAsync(h_mc, mcflag, 8, hipMemcpyDeviceToHost, st));
    SYNCK(c, st);
    if (mc && *h_mc != 0.) {
        // the workgroup cluster gave up waiting for a peer (a workgroup that never got a CU): nothing it wrote is used.  Redo this
        // split with the stock solver: Gram matrix again (sG may have served as workspace), dsyevd, eigenvalues to the host.
        c->svd_fallbacks += 1;
        HIPCK(c, hipMemsetAsync(const_cast<void*>(static_cast<const void*>(static_cast<const char*>(eigh_mc_status_ptr(c->mc_xbuf)) - 8)), 0, 16, st));
        if (left) TCK(split_gemm(c, false, true, nl, nl, nr, M, nl, M, nl, c->sG, nl, gstrips));
        else      TCK(split_gemm(c, true, false, nr, nr, nl, M, nl, M, nl, c->sG, nr, gstrips));
        if (dm) TCK(noise_add(c, B_it, b, ha, mL, mR, c->sG, n));
        RBCK(c, rocsolver_dsyevd(c->blas, rocblas_evect_original, rocblas_fill_upper, n, c->sG, n, c->sD, c->sE, c->sInfo));
        evals = c->sD; own_eig = false; stock = true; Q = c->sF; direct_left = false;
        TCK(bcast_rank0(c, const_cast<double*>(evals), n));
        HIPCK(c, hipMemcpyAsync(h, evals, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        SYNCK(c, st);
    }
    if (own_eig) { hd[0] = h[n]; hd[1] = h[n + 1]; hd[2] = h[n + 2]; }
    if (c->svd_print >= 0) {                                                   // debugging aid (option svd_print = k): the spectrum of the k-th split of this context
        if (c->svd_calls++ == c->svd_print) { fprintf(stderr, "svd_spectrum n=%d:", n); for (int g = 0; g < n; ++g) fprintf(stderr, " %.3e", h[n - 1 - g]); fprintf(stderr, "\n"); }
    }
    std::vector<double> p(n), sig(n);
    for (int g = 0; g < n; ++g) { double lam = h[n - 1 - g]; if (!(lam > 0.)) lam = 0.; p[g] = lam; sig[g] = std::sqrt(lam); }
    double te = 0.;
    m = tnml_truncate(p.data(), n, maxm, minm, cutoff, &te);
    if (truncerr) *truncerr = te;
    if (newm) *newm = m;
    if (nsv) *nsv = n;
    if (sv_host) for (int g = 0; g < n; ++g) sv_host[g] = sig[g];
    if (m > c->maxm) return tnml_fail(c, "svd_split: new bond dimension %d exceeds maxm %d of the context", m, c->maxm);

    }
    const SigmaRef d_sig{evals, n, 0}, d_isig{evals, n, 1}, no_scale{nullptr, 0, 0};   // sigma_g / 1/sigma_g of the kept columns, computed where they are used

    if (own_eig && !spec) {
        if (c->svd_print == -1) {                                                  // debugging aid (option svd_print = -1): the check values of every split
            double nref = -1.;
            (void)hipMemcpy(&nref, c->sTau + (n - 1), sizeof(double), hipMemcpyDeviceToHost);
            fprintf(stderr, "svd_check n=%d mk=%d dev=%.2e cholfail=%g factored=%g dev_in=%.2e reflectors=%g\n", n, mk, hd[0], hd[1], hd[2], h[n + 3], nref);
        }
        c->svd_last_dev0 = hd[0]; c->svd_last_dev1 = 0.75 * hd[0] * hd[0];      // Newton-Schulz: error -> 3/4 error^2
        const bool ok = hd[0] < 1e-6 && hd[1] == 0.;                             // the polish step leaves 3/4 d^2 < 1e-12
        if (hd[2] != 0.) c->svd_cholqr += 1;
        if (!ok && !c->svd_dump.empty()) {                                       // debugging aid (TNML_SVD_DUMP at tnml_create): the offending tridiagonal problem
            const char* dump = c->svd_dump.c_str();
            int& dumped = c->svd_dumped;
            if (dumped < 4) {
                std::vector<double> hb((size_t)3 * n + (size_t)n * mk + 2);
                hb[0] = n; hb[1] = mk;
                (void)hipMemcpy(hb.data() + 2, c->sD, sizeof(double) * n, hipMemcpyDeviceToHost);
                (void)hipMemcpy(hb.data() + 2 + n, c->sE2, sizeof(double) * n, hipMemcpyDeviceToHost);
                (void)hipMemcpy(hb.data() + 2 + 2 * n, c->sW, sizeof(double) * n, hipMemcpyDeviceToHost);
                (void)hipMemcpy(hb.data() + 2 + 3 * n, c->sC, sizeof(double) * (size_t)n * mk, hipMemcpyDeviceToHost);
                char fn[512]; snprintf(fn, sizeof fn, "%s.%d.bin", dump, dumped++);
                if (FILE* f = fopen(fn, "wb")) { fwrite(hb.data(), sizeof(double), hb.size(), f); fclose(f); }
            }
        }
        if (!ok) {
            // dependent vectors even after re-orthonormalisation: redo the tridiagonal stage with rocSOLVER's
            // divide and conquer (D, E, V, tau are still intact)
            c->svd_fallbacks += 1;
            own_eig = false;
            Q = c->sF; direct_left = false;
            RBCK(c, rocsolver_dstedc(c->blas, rocblas_evect_tridiagonal, n, c->sD, c->sE2, c->sC, n, c->sInfo));
        }
    }
    if (tri && !own_eig && !stock) {
        // kept eigenvectors of the tridiagonal matrix (largest first), then U = H_0 H_1 ... Z
        hipLaunchKernelGGL(k_take_top, dim3(nblk((size_t)n * m)), dim3(256), 0, st, c->sC, c->sG, n, m, (const double*)nullptr);
        TCK(eigh_backtransform(c, c->sV, c->sTau, n, c->sG, n, Q, n, m));
    } else if (stock) {
        hipLaunchKernelGGL(k_take_top, dim3(nblk((size_t)n * m)), dim3(256), 0, st, c->sG, Q, n, m, (const double*)nullptr);
    }
    double* Aleft = labL ? Lf : Sl.a;      // left factor target, (nl x m), ld = nl
    double* Aright = Sr.a;                 // right factor (m x nr), ld = m  == A_{b+1}[g][t][be](,[l])
    if (left) {
        // Q = U_m
        TCK(split_gemm(c, true, false, m, nr, nl, Q, nl, M, nl, Aright, m, 2, chkp));   // U^T M = S V^T
        if (Q != Aleft) HIPCK(c, hipMemcpyAsync(Aleft, Q, sizeof(double) * (size_t)nl * m, hipMemcpyDeviceToDevice, st));
        if (ha == 2) {   // orthonormal factor goes right: V^T = S^-1 U^T M ; left gets U S
            hipLaunchKernelGGL(k_scale_rows, dim3(nblk((size_t)m * nr)), dim3(256), 0, st, Aright, m, (size_t)nr, d_isig);
            hipLaunchKernelGGL(k_scale_cols, dim3(nblk((size_t)nl * m)), dim3(256), 0, st, Aleft, (size_t)nl, m, d_sig);
        }
    } else {
        // Q = V_m
        TCK(split_gemm(c, false, false, nl, m, nr, M, nl, Q, nr, Aleft, nl, 2, chkp));        // M V = U S
        if (ha == 2) {
            hipLaunchKernelGGL(k_transpose_scale, dim3(nblk((size_t)nr * m)), dim3(256), 0, st, Q, Aright, nr, m, no_scale);
        } else {         // orthonormal factor goes left: U = M V S^-1 ; right gets S V^T
            hipLaunchKernelGGL(k_scale_cols, dim3(nblk((size_t)nl * m)), dim3(256), 0, st, Aleft, (size_t)nl, m, d_isig);
            hipLaunchKernelGGL(k_transpose_scale, dim3(nblk((size_t)nr * m)), dim3(256), 0, st, Q, Aright, nr, m, d_sig);
        }
    }
    if (labL) hipLaunchKernelGGL(k_perm_labL_back, dim3(nblk((size_t)nl * m)), dim3(256), 0, st, Lf, Sl.a, 2 * mL, m);
    HIPCK(c, hipGetLastError());
    Sl.mr = m; Sr.ml = m;
    return 0;
}
