// kernels_small.hip -- layout conversion, bond-tensor product and the replicated CG vector algebra.
//
// These touch only O(m^2) data (<= 576k elements) and are latency items, not roofline items.
// The CG scalar algebra of cgrad (fixedL.cc:386-388,403-407,422-428,442) runs on device scalars in
// fp64 so that no host round trip sits inside a CG pass.
#include "tnml_internal.h"

// ---- pack / unpack -------------------------------------------------------------------------
__global__ void k_pack(PackDesc d, const double* __restrict__ T, double* __restrict__ Md, float* __restrict__ Mf, double* __restrict__ zero, int nzero) {
    const size_t per = (size_t)d.Kp * d.Np, total = per * d.L;
    if (blockIdx.x == 0 && (int)threadIdx.x < nzero) zero[threadIdx.x] = 0.;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / per);
        const int k = (int)((idx % per) / d.Np), j = (int)(idx % d.Np);
        const int x = k >> 1, s = k & 1;
        const int y = d.TO == 2 ? (j >> 1) : j, t = d.TO == 2 ? (j & 1) : 0;
        double v = 0.;
        if (x < d.nx && y < d.ny) v = T[x * d.sx + s * d.ss + y * d.sy + t * d.st + l * d.sl];
        if (Md) Md[idx] = v;
        if (Mf) Mf[idx] = (float)v;
    }
}
__global__ void k_unpack(PackDesc d, const double* __restrict__ Md, double* __restrict__ T) {
    const size_t per = (size_t)d.Kp * d.Np, total = per * d.L;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / per);
        const int k = (int)((idx % per) / d.Np), j = (int)(idx % d.Np);
        const int x = k >> 1, s = k & 1;
        const int y = d.TO == 2 ? (j >> 1) : j, t = d.TO == 2 ? (j & 1) : 0;
        if (x < d.nx && y < d.ny) T[x * d.sx + s * d.ss + y * d.sy + t * d.st + l * d.sl] = Md[idx];
    }
}
__global__ void k_cvt(const double* __restrict__ s, float* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
__global__ void k_fill_f32(float* p, float v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static inline int nblocks(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b ? b : 1)); }

int launch_pack(tnml_ctx* c, const PackDesc& d, const double* T, double* Md, float* Mf, double* zero, int nzero) {
    ProfScope ps(c, KC_PACK);
    hipLaunchKernelGGL(k_pack, dim3(nblocks((size_t)d.Kp * d.Np * d.L)), dim3(256), 0, c->stream, d, T, Md, Mf, zero, zero ? nzero : 0);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_unpack(tnml_ctx* c, const PackDesc& d, const double* Md, double* T) {
    ProfScope ps(c, KC_PACK);
    hipLaunchKernelGGL(k_unpack, dim3(nblocks((size_t)d.Kp * d.Np * d.L)), dim3(256), 0, c->stream, d, Md, T);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_cvt(tnml_ctx* c, const double* src, float* dst, size_t n) {
    ProfScope ps(c, KC_PACK);
    hipLaunchKernelGGL(k_cvt, dim3(nblocks(n)), dim3(256), 0, c->stream, src, dst, n);
    HIPCK(c, hipGetLastError());
    return 0;
}
__global__ void k_fill_f64(double* p, double v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill_f64(tnml_ctx* c, double* p, double v, size_t n) {
    hipLaunchKernelGGL(k_fill_f64, dim3(nblocks(n)), dim3(256), 0, c->stream, p, v, n);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_fill_f32(tnml_ctx* c, float* p, float v, size_t n) {
    hipLaunchKernelGGL(k_fill_f32, dim3(nblocks(n)), dim3(256), 0, c->stream, p, v, n);
    HIPCK(c, hipGetLastError());
    return 0;
}

// ---- oB = W.A(b)*W.A(b+1)  (fixedL.cc:494,527) ----------------------------------------------
__global__ void k_bond_form(const double* __restrict__ A1, const double* __restrict__ A2, double* __restrict__ B,
                            int mL, int k, int mR, int L1, int L2) {
    const int LB = L1 > L2 ? L1 : L2;
    const size_t total = (size_t)mL * 4 * mR * LB;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int a = (int)(r % mL); r /= mL;
        const int s = (int)(r % 2); r /= 2;
        const int t = (int)(r % 2); r /= 2;
        const int be = (int)(r % mR); r /= mR;
        const int l = (int)r;
        const int l1 = L1 == 1 ? 0 : l, l2 = L2 == 1 ? 0 : l;
        const double* p1 = A1 + a + (size_t)mL * (s + 2 * ((size_t)k * l1));      // + mL*2*g
        const double* p2 = A2 + (size_t)k * (t + 2 * (be + (size_t)mR * l2));      // + g
        double acc = 0.;
        for (int g = 0; g < k; ++g) acc += p1[(size_t)mL * 2 * g] * p2[g];
        B[idx] = acc;
    }
}
int launch_bond_form(tnml_ctx* c, const SiteT& A1, const SiteT& A2, double* B) {
    ProfScope ps(c, KC_SMALLGEMM);
    const int LB = A1.L > A2.L ? A1.L : A2.L;
    const int nl = 2 * A1.ml, k = A1.mr, nr = 2 * A2.mr;
    // B[(a,s)][(t,be)](,l) = A1[(a,s)][g] * A2[g][(t,be)]: a plain column-major GEMM in ITensor index order.  The Label
    // index rides as extra columns when it sits on the right site and as a strided batch when it sits on the left one.
    if (k >= 16 && (A1.L == 1 || A2.L == 1)) {
        const double one = 1.0, zero = 0.0;
        rocblas_status st;
        if (A1.L == 1) return split_gemm(c, false, false, nl, nr * A2.L, k, A1.a, nl, A2.a, k, B, nl, 2);      // k_dgemm_small (kernels_sgemm.hip) unless option small_gemm = 0
        else           st = rocblas_dgemm_strided_batched(c->blas, rocblas_operation_none, rocblas_operation_none, nl, nr, k, &one, A1.a, nl, (rocblas_stride)nl * k,
                                                         A2.a, k, 0, &zero, B, nl, (rocblas_stride)nl * nr, A1.L);
        if (st != rocblas_status_success) return tnml_fail(c, "bond_form: rocblas dgemm failed (%d)", (int)st);
        return 0;
    }
    const size_t total = (size_t)A1.ml * 4 * A2.mr * LB;
    hipLaunchKernelGGL(k_bond_form, dim3(nblocks(total)), dim3(256), 0, c->stream, A1.a, A2.a, B, A1.ml, A1.mr, A2.mr, A1.L, A2.L);
    HIPCK(c, hipGetLastError());
    return 0;
}

// ---- CG vector algebra: two-phase multi-workgroup kernels (fp64, fixed summation order) ---------
// phase 1: every workgroup reduces its contiguous slice to partial sums part[blk][2];
// phase 2: every workgroup re-reduces the partials in the same order (identical scalars everywhere),
//          derives the CG scalars and updates its slice; workgroup 0 publishes the scalars.
// A single workgroup streams at ~25-50 GB/s (one CU), which made the 1-block versions 40-90 us each.
#define VB 256
#define VNB_MAX 256
// workgroup sum (VB lanes, every lane active): DPP tree inside each wave, then the waves in order
static __device__ __forceinline__ double block_sum(double v, double* sh) {
    const double w = wave_sum(v);
    const int tid = threadIdx.x;
    __syncthreads();                                        // readers of a previous call are done with sh
    if ((tid & 63) == 0) sh[tid >> 6] = w;
    __syncthreads();
    double r = 0.;
#pragma unroll
    for (int k = 0; k < VB / 64; ++k) r += sh[k];
    return r;
}
static __device__ __forceinline__ void slice(size_t n, size_t* lo, size_t* hi) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    *lo = per * blockIdx.x; *hi = *lo + per; if (*hi > n) *hi = n; if (*lo > n) *lo = n;
}
// sum of the nb <= VB phase-1 partial pairs, by the whole workgroup: lane k takes pair k, then the fixed
// block_sum tree -> the same value in every workgroup (a serial walk by every lane cost ~6 us per kernel)
static __device__ __forceinline__ void sum_partials(const double* part, int nb, double* s0, double* s1, double* sh) {
    const int tid = threadIdx.x;
    const double a = tid < nb ? part[2 * tid] : 0., b = tid < nb ? part[2 * tid + 1] : 0.;
    *s0 = block_sum(a, sh);
    *s1 = block_sum(b, sh);
}

// column `col` of the per-image-block partial sums partials[nblk][12] of a forward pass / output update, by the whole workgroup:
// lane k takes rows k, k + VB, ... in order, then the fixed block_sum tree -> the same value in every workgroup.  Lets the CG step
// kernels consume the partial sums themselves instead of waiting for a k_reduce_partials launch of their own.
static __device__ __forceinline__ double sum_column(const double* __restrict__ partials, int nblk, int col, double* sh) {
    // the summation order of k_reduce_partials (lane i takes rows i, i + 64, ... in order, then its shuffle tree), so that a
    // folded reduction gives the bits the separate launch gave: wave 0 sums, the workgroup reads the result
    __syncthreads();                                        // readers of a previous call are done with sh
    if (threadIdx.x < 64) {
        double a = 0.;
        for (int r = threadIdx.x; r < nblk; r += 64) a += partials[(size_t)r * 12 + col];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
        if (threadIdx.x == 0) sh[0] = a;
    }
    __syncthreads();
    return sh[0];
}

// r = G - lambda*B (fixedL.cc:385-386); p = r (:388); partial |r|^2
// slab != nullptr (one rank): G = the sum of the nsplit split-K slabs the gradient GEMM left, in k_slab_reduce64's order -- the reduction
// launch of its own is folded in here.  set_flags: workgroup 0 also does what k_cg_init2 did besides the sum (clears the trace and the
// convergence flags); |r|^2 itself is then summed from `part` by the first k_cg_step2.
__global__ __launch_bounds__(VB) void k_cg_init1(double* __restrict__ G, const double* __restrict__ B, double* __restrict__ R,
                                                double* __restrict__ Pv, size_t n, double lambda, double* __restrict__ part,
                                                const double* __restrict__ slab, int nsplit, double* __restrict__ scal, int set_flags) {
    __shared__ double sh[VB];
    if (set_flags && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 4 * TNML_MAX_PASS; i += VB) scal[SC_N + i] = 0.;
        if (threadIdx.x == 0) { scal[SC_CONV] = 0.; scal[SC_CONVP] = 0.; scal[SC_NPASS] = 0.; }
    }
    size_t lo, hi; slice(n, &lo, &hi);
    double acc = 0.;
    for (size_t i = lo + threadIdx.x; i < hi; i += VB) {
        double r;
        if (slab) {
            double g = 0.;
#pragma unroll 8
            for (int k = 0; k < nsplit; ++k) g += slab[(size_t)k * n + i];
            G[i] = g; r = g;
        } else r = G[i];
        if (lambda != 0.) r = r - lambda * B[i];
        R[i] = r; Pv[i] = r;
        acc += r * r;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = 0.; }
}
// cconv0 >= 0 (TNML_MODE_SINGLE): |r| < cconv at entry -> "not optimizing" (single.h:202-206): flag 2 freezes every later kernel
__global__ __launch_bounds__(VB) void k_cg_init2(const double* __restrict__ part, int nb, double* __restrict__ scal, int rr_out, double cconv0) {
    __shared__ double sh[VB / 64];
    for (int i = threadIdx.x; i < 4 * TNML_MAX_PASS; i += VB) scal[SC_N + i] = 0.;      // the per-pass trace lives behind the scalars
    double a, b; sum_partials(part, nb, &a, &b, sh);
    if (threadIdx.x == 0) {
        const double cv = (cconv0 >= 0. && sqrt(a) < cconv0) ? 2. : 0.;
        scal[rr_out] = a; scal[SC_CONV] = cv; scal[SC_CONVP] = cv; scal[SC_NPASS] = 0.;      // slot 0: the state before pass 1
    }
}
// partial |x|^2 (and |y|^2)
__global__ __launch_bounds__(VB) void k_norm1(const double* __restrict__ x, const double* __restrict__ y, size_t n, double* __restrict__ part) {
    __shared__ double sh[VB];
    size_t lo, hi; slice(n, &lo, &hi);
    double a = 0., b = 0.;
    for (size_t i = lo + threadIdx.x; i < hi; i += VB) { a += x[i] * x[i]; if (y) b += y[i] * y[i]; }
    const double s0 = block_sum(a, sh);
    const double s1 = block_sum(b, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s0; part[2 * blockIdx.x + 1] = s1; }
}
// pAp = sum_n|p v_n|^2 + lambda|p|^2 (:402-403); a = |r|^2/pAp (:405); B = B + a p (:406)
// Workgroups [0, nbv) update B.  With U.P != nullptr the launch carries nbu more workgroups that apply the SAME step to the model
// outputs, P <- P + a (p*t.v) (the fast CG's k_pupdate, a launch of its own before round 5): every workgroup derives a from the same
// partial sums in the same order, so both halves use identical bits.
struct StepUpd { double* P; const double* Pp; double* dP; const int* label; int NTp; double* partials; int nl, target; };
__global__ __launch_bounds__(VB) void k_cg_step2(double* __restrict__ B, const double* __restrict__ Pv, size_t n, double lambda,
                                                const double* __restrict__ tail, const double* __restrict__ part, int nb,
                                                double* __restrict__ scal, int rr_in, double* __restrict__ trace, int pass, int merged,
                                                const double* __restrict__ pp_part, int npp,
                                                const double* __restrict__ rr_part, int nbv, StepUpd U, double* __restrict__ hmir) {
    // merged CG: the cost partials of the PREVIOUS pass's update (fixedL.cc:419,427-428) came with this pass's all-reduce
    if (merged && pass > 1 && blockIdx.x == 0 && threadIdx.x == 0 && scal[SC_CONVP + (pass & 1)] == 0.) {   // (slot of pass - 2: not converged before the previous pass)
        double cs = 0.;
        for (int l = 0; l < TNML_NL; ++l) cs += tail[SC_COST0 + l];
        const double cst = cs + lambda * scal[SC_BNORM2];
        scal[SC_COST] = cst; trace[4 * (pass - 2) + 2] = cst;
        if (hmir) hmir[SC_N + 4 * (pass - 2) + 2] = cst;
    }
    if (scal[SC_CONVP + ((pass - 1) & 1)] != 0.) return;   // |r| < cconv was hit in an earlier pass (fixedL.cc:432-436)
    __shared__ double sh[VB];
    // |r|^2: left in scal by k_cg_resid2 / k_cg_init2, or (pass 1 without k_cg_init2) still as k_cg_init1's partial sums
    double rr, unused;
    if (rr_part) sum_partials(rr_part, nb, &rr, &unused, sh); else rr = scal[rr_in];
    // |p|^2: p = r in pass 1 (fixedL.cc:388), afterwards the partial sums left by k_cg_resid2 when it formed p = r + beta p
    double pn2;
    if (pass == 1) pn2 = rr; else sum_partials(part, nb, &pn2, &unused, sh);
    // sum_n |p.v_n|^2: reduced already (tail), or still as the per-block partial sums of the pAp pass (column 11)
    const double pp = pp_part ? sum_column(pp_part, npp, 11, sh) : tail[SC_PP];
    const double pAp = pp + lambda * pn2;
    const double a = rr / pAp;
    if ((int)blockIdx.x >= nbv) {                          // output update: two units of LD_IMGS images per workgroup
        __shared__ double s_part[(VB / 64) * 12];
        const int half = threadIdx.x / LD_IMGS;
        const int unit = ((int)blockIdx.x - nbv) * (VB / LD_IMGS) + half;
        pupdate_unit<double>(U.P, U.Pp, U.dP, U.label, U.NTp, a, U.partials, U.nl, U.target, s_part + half * (LD_IMGS / 64) * 12, (int)threadIdx.x % LD_IMGS, unit);
        return;
    }
    // slice over the nbv vector workgroups (gridDim.x may be larger)
    {
        const size_t per = (n + nbv - 1) / nbv;
        size_t lo = per * blockIdx.x, hi = lo + per; if (hi > n) hi = n; if (lo > n) lo = n;
        for (size_t i = lo + threadIdx.x; i < hi; i += VB) B[i] = B[i] + a * Pv[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal[SC_PNORM2] = pn2; scal[SC_PAP] = pAp; scal[SC_ALPHA] = a; scal[SC_NPASS] = (double)pass;
        if (rr_part) scal[rr_in] = rr;                     // k_cg_resid2 reads it
        trace[4 * (pass - 1) + 0] = pAp; trace[4 * (pass - 1) + 1] = a;
        // hmir: the pinned host mirror of [scal | trace] of the bond update in flight -- what the host report needs lands there without a copy
        if (hmir) { hmir[SC_NPASS] = (double)pass; hmir[SC_N + 4 * (pass - 1) + 0] = pAp; hmir[SC_N + 4 * (pass - 1) + 1] = a; }
    }
}
// partial |nr|^2 with nr = G - lambda B, and |B|^2
// merged (R, Pv given): G holds A p = sum_n (p.v_n) v_n and nr = r - a (A p + lambda p) with a = scal[SC_ALPHA] (single.h:378-379 structure)
// slab != nullptr (one rank, literal order): G is first formed as the ordered sum of the gradient GEMM's split-K slabs (the
// k_slab_reduce64 launch folded in)
__global__ __launch_bounds__(VB) void k_cg_resid1(double* __restrict__ G, const double* __restrict__ B, size_t n, double lambda,
                                                 double* __restrict__ part, const double* __restrict__ R, const double* __restrict__ Pv, const double* __restrict__ scal,
                                                 const double* __restrict__ slab, int nsplit, int pass,
                                                 const double* __restrict__ cost_part, int ncp, double* __restrict__ cost_sum, int nbv) {
    if (slab && scal[SC_CONVP + ((pass - 1) & 1)] != 0.) return;   // converged: k_cg_resid2 will not read G or the partials (the separate reduction launch used to run regardless)
    __shared__ double sh[VB];
    if ((int)blockIdx.x >= nbv) {
        // the one workgroup beyond the vector's: the cost partials of the output update (per label, in k_reduce_partials' order, then the
        // labels in label order) -> cost_sum[0], beside the others' streaming instead of on k_cg_resid2's critical path (6 us there)
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int l = w; l < TNML_NL; l += VB / 64) {
            double a = 0.;
            for (int r = lane; r < ncp; r += 64) a += cost_part[(size_t)r * 12 + l];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if (lane == 0) sh[l] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) { double cs = 0.; for (int l = 0; l < TNML_NL; ++l) cs += sh[l]; cost_sum[0] = cs; }
        return;
    }
    size_t lo, hi;
    { const size_t per = (n + nbv - 1) / nbv; lo = per * blockIdx.x; hi = lo + per; if (hi > n) hi = n; if (lo > n) lo = n; }
    double an = 0., ab = 0.;
    const double a = R ? scal[SC_ALPHA] : 0.;
    for (size_t i = lo + threadIdx.x; i < hi; i += VB) {
        double nr;
        if (slab) {
            double g = 0.;
#pragma unroll 8
            for (int k = 0; k < nsplit; ++k) g += slab[(size_t)k * n + i];
            G[i] = g;
        }
        if (R) { double t = G[i]; if (lambda != 0.) t = t + lambda * Pv[i]; nr = R[i] - a * t; }
        else { nr = G[i]; if (lambda != 0.) nr = nr - lambda * B[i]; }
        an += nr * nr; ab += B[i] * B[i];
    }
    const double s0 = block_sum(an, sh);
    const double s1 = block_sum(ab, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s0; part[2 * blockIdx.x + 1] = s1; }
}
// beta = sqr(norm(nr)/norm(r)) (:423); r = nr (:424); C = sum dP^2 + lambda|B|^2 (:427-428);
// conv = |r| < cconv (:432); p = r + beta p (:442)
__global__ __launch_bounds__(VB) void k_cg_resid2(const double* __restrict__ G, const double* __restrict__ B, double* __restrict__ R,
                                                 double* __restrict__ Pv, size_t n, double lambda, double cconv,
                                                 const double* __restrict__ tail, const double* __restrict__ part, int nb,
                                                 double* __restrict__ scal, int rr_in, int rr_out,
                                                 double* __restrict__ trace, int pass, double* __restrict__ part_p, int merged,
                                                 const double* __restrict__ cost_part, int ncp, double* __restrict__ hmir) {
    const double was = scal[SC_CONVP + ((pass - 1) & 1)];
    if (was != 0.) {                                       // already converged: hand the flag on to the next pass's slot
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_CONVP + (pass & 1)] = was;
        return;
    }
    __shared__ double sh[VB];
    double nn, bn2; sum_partials(part, nb, &nn, &bn2, sh);
    const double q = sqrt(nn) / sqrt(scal[rr_in]);
    const double beta = q * q;
    const double rn = sqrt(nn);
    const int conv = rn < cconv;
    size_t lo, hi; slice(n, &lo, &hi);
    double pacc = 0.;
    const double a = merged ? scal[SC_ALPHA] : 0.;
    for (size_t i = lo + threadIdx.x; i < hi; i += VB) {
        double nr;
        if (merged) { double t = G[i]; if (lambda != 0.) t = t + lambda * Pv[i]; nr = R[i] - a * t; }
        else { nr = G[i]; if (lambda != 0.) nr = nr - lambda * B[i]; }
        R[i] = nr;
        if (!conv) { const double pv = nr + beta * Pv[i]; Pv[i] = pv; pacc += pv * pv; }
    }
    const double ps = block_sum(pacc, sh);                 // partial |p|^2 of the next pass (k_cg_step2 sums them in block order)
    if (threadIdx.x == 0) { part_p[2 * blockIdx.x] = ps; part_p[2 * blockIdx.x + 1] = 0.; }
    double csp = 0.;                                       // workgroup 0: the cost partials of the output update, when they have not been reduced yet
    if (blockIdx.x == 0 && cost_part && !merged) {
        if (ncp < 0) csp = cost_part[0];                   // already summed by the extra workgroup of k_cg_resid1
        else {
        // wave w sums the columns of labels w, w + 4, w + 8 in the order of k_reduce_partials; the labels are then added in label order,
        // as the reduced tail would be (one pass over the partial sums instead of ten workgroup-wide ones: 19 -> 6 us)
        __syncthreads();
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int l = w; l < TNML_NL; l += VB / 64) {
            double a = 0.;
            for (int r = lane; r < ncp; r += 64) a += cost_part[(size_t)r * 12 + l];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if (lane == 0) sh[l] = a;
        }
        __syncthreads();
        for (int l = 0; l < TNML_NL; ++l) csp += sh[l];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!merged) {                                    // (merged: this pass's cost partials are summed over the ranks by the next all-reduce)
            double cs = 0.;
            if (cost_part) cs = csp;
            else for (int l = 0; l < TNML_NL; ++l) cs += tail[SC_COST0 + l];
            scal[SC_COST] = cs + lambda * bn2;
            trace[4 * (pass - 1) + 2] = cs + lambda * bn2;
            if (hmir) hmir[SC_N + 4 * (pass - 1) + 2] = cs + lambda * bn2;
        }
        scal[SC_BNORM2] = bn2; scal[SC_BETA] = beta; scal[SC_RNORM] = rn;
        scal[rr_out] = nn;
        trace[4 * (pass - 1) + 3] = rn;
        scal[SC_CONVP + (pass & 1)] = (double)conv;        // read by the kernels of the next pass; this pass's readers use the other slot
        scal[SC_CONV] = (double)conv;                      // host copy
        if (hmir) { hmir[SC_N + 4 * (pass - 1) + 3] = rn; hmir[SC_CONV] = (double)conv; }
    }
}
// method = fast_conj (single.h:290-398): the image sum of this pass is A p = sum_n (p.v_n) v_n, and the residual follows the
// recurrence nr = r - a*Ap (:378); writing G <- r - a*Ap lets k_cg_resid1/2 finish the pass exactly as the reference writes it
// (":379 nr = nr - lambda*B").  No cost is evaluated on this path (the reference prints none): the cost partials are cleared.
__global__ __launch_bounds__(VB) void k_cg_fast_resid0(double* __restrict__ G, const double* __restrict__ R, size_t n, const double* __restrict__ scal,
                                                      double* __restrict__ tail, int pass) {
    if (scal[SC_CONVP + ((pass - 1) & 1)] != 0.) return;
    const double a = scal[SC_ALPHA];
    size_t lo, hi; slice(n, &lo, &hi);
    for (size_t i = lo + threadIdx.x; i < hi; i += VB) G[i] = R[i] - a * G[i];
    if (blockIdx.x == 0 && threadIdx.x < TNML_NL) tail[SC_COST0 + threadIdx.x] = 0.;
}
__global__ __launch_bounds__(VB) void k_norm2(const double* __restrict__ part, int nb, double* __restrict__ out, int nout) {
    __shared__ double sh[VB / 64];
    double a, b; sum_partials(part, nb, &a, &b, sh);
    if (threadIdx.x == 0) { out[0] = a; if (nout == 2) out[1] = b; if (nout == 3) { out[1] = a; out[2] = b; } }
}
__global__ __launch_bounds__(VB) void k_diffnorm1(const double* __restrict__ x, const double* __restrict__ y, size_t n, double* __restrict__ part) {   // part may be pinned host memory
    __shared__ double sh[VB];
    size_t lo, hi; slice(n, &lo, &hi);
    double a = 0., d = 0.;
    for (size_t i = lo + threadIdx.x; i < hi; i += VB) { a += x[i] * x[i]; const double t = x[i] - y[i]; d += t * t; }
    const double s0 = block_sum(a, sh);
    const double s1 = block_sum(d, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s0; part[2 * blockIdx.x + 1] = s1; }
}

// Workgroups of the CG vector kernels.  Large vectors (config 3: 57 600 elements): one element per lane, up to VNB_MAX workgroups --
// the kernels that fold the split-K slab reduction read nsplit values per element and need the width.  Small vectors keep the
// partition of rounds 1-4 (1024 elements per workgroup): nothing to gain there, and the partition IS the summation order of
// |r|^2, |p|^2, |B|^2 -- the free-running CLI test on the reference's badly conditioned feature map amplifies a change of it
// (tests/test_gpu_parity.py::test_fixedl_cli_driver_end_to_end) exactly as it amplifies the oracle's own thread count.
static inline int vec_blocks(size_t n) { const size_t per = n >= 16384 ? VB : 1024; size_t b = (n + per - 1) / per; if (b > VNB_MAX) b = VNB_MAX; if (b < 1) b = 1; return (int)b; }

// the |r|^2 of the previous evaluation lives in scal[SC_RR + (c->rr_slot)], alternating between two
// slots so that phase-2 workgroups never read a slot another workgroup is writing
int launch_cg_init(tnml_ctx* c, size_t n, double lambda, double cconv0) {
    ProfScope ps(c, KC_VEC);
    const int nb = vec_blocks(n);
    c->rr_slot = 0;
    // one rank (the slabs of the gradient GEMM are still unreduced: cgrad_device's fold path): the slab sum and, without the entry
    // check of the per-label variant, everything k_cg_init2 did are folded into k_cg_init1; |r|^2 is summed by the first k_cg_step2
    const double* slab = c->slab_pending > 0 ? (const double*)c->slab : nullptr;
    const int nsplit = c->slab_pending;
    c->slab_pending = 0;
    c->rr_from_part = cconv0 < 0. && slab != nullptr;
    hipLaunchKernelGGL(k_cg_init1, dim3(nb), dim3(VB), 0, c->stream, c->vG, c->vB, c->vR, c->vP, n, lambda, c->vpart, slab, nsplit, c->scal, c->rr_from_part ? 1 : 0);
    if (!c->rr_from_part) hipLaunchKernelGGL(k_cg_init2, dim3(1), dim3(VB), 0, c->stream, c->vpart, nb, c->scal, (int)SC_RR, cconv0);
    HIPCK(c, hipGetLastError());
    return 0;
}
// with_update (one rank, f64, literal pass order): the launch also applies the step to the model outputs (P, dP and the cost partials of
// the new B -> c->partials2), what launch_pupdate did in a launch of its own
int launch_cg_step(tnml_ctx* c, size_t n, double lambda, int pass, bool merged, const double* pp_part, int npp, bool with_update) {
    ProfScope ps(c, KC_VEC);
    const int nb = vec_blocks(n);
    StepUpd U{nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, 0};
    int nbu = 0;
    if (with_update) {
        U = StepUpd{(double*)c->P, (const double*)c->Pp, (double*)c->dP, c->label, c->NTp, c->partials2, c->nl(), c->target()};
        nbu = c->NTp / VB;
        c->part_n = c->NTp / LD_IMGS;
    }
    const double* rr_part = (pass == 1 && c->rr_from_part) ? (const double*)c->vpart : nullptr;
    hipLaunchKernelGGL(k_cg_step2, dim3(nb + nbu), dim3(VB), 0, c->stream, c->vB, c->vP, n, lambda, c->tail, c->vpart + 512, nb, c->scal, SC_RR + c->rr_slot, c->cgtrace, pass, merged ? 1 : 0, pp_part, npp, rr_part, nb, U, c->hmir);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_cg_resid(tnml_ctx* c, size_t n, double lambda, double cconv, int pass, bool merged, const double* cost_part, int ncp) {
    ProfScope ps(c, KC_VEC);
    const int nb = vec_blocks(n);
    const int in = SC_RR + c->rr_slot, out = SC_RR + (c->rr_slot ^ 1);
    const double* slab = c->slab_pending > 0 ? (const double*)c->slab : nullptr;
    const int nsplit = c->slab_pending;
    c->slab_pending = 0;
    // the cost partials of the output update (one rank: not reduced yet) are summed by one extra workgroup of k_cg_resid1
    const bool early = cost_part != nullptr && !merged;
    double* csum = c->vpart + 1024;
    hipLaunchKernelGGL(k_cg_resid1, dim3(nb + (early ? 1 : 0)), dim3(VB), 0, c->stream, c->vG, c->vB, n, lambda, c->vpart, merged ? (const double*)c->vR : (const double*)nullptr, (const double*)c->vP, (const double*)c->scal, slab, nsplit, pass,
                       cost_part, ncp, csum, nb);
    hipLaunchKernelGGL(k_cg_resid2, dim3(nb), dim3(VB), 0, c->stream, c->vG, c->vB, c->vR, c->vP, n, lambda, cconv, c->tail, c->vpart, nb, c->scal, in, out, c->cgtrace, pass, c->vpart + 512, merged ? 1 : 0,
                       early ? (const double*)csum : cost_part, early ? -1 : ncp, c->hmir);
    c->rr_slot ^= 1;
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_cg_fast_resid0(tnml_ctx* c, size_t n, int pass) {
    ProfScope ps(c, KC_VEC);
    hipLaunchKernelGGL(k_cg_fast_resid0, dim3(vec_blocks(n)), dim3(VB), 0, c->stream, c->vG, c->vR, n, c->scal, c->tail, pass);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_sqnorm(tnml_ctx* c, const double* x, size_t n, double* out) {
    ProfScope ps(c, KC_VEC);
    const int nb = vec_blocks(n);
    hipLaunchKernelGGL(k_norm1, dim3(nb), dim3(VB), 0, c->stream, x, (const double*)nullptr, n, c->vpart);
    hipLaunchKernelGGL(k_norm2, dim3(1), dim3(VB), 0, c->stream, c->vpart, nb, out, 1);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_diffnorm(tnml_ctx* c, const double* x, const double* y, size_t n, double* out2, int nout) {      // nout = 3: out = |x|^2, |x|^2, |x - y|^2
    ProfScope ps(c, KC_VEC);
    const int nb = vec_blocks(n);
    hipLaunchKernelGGL(k_diffnorm1, dim3(nb), dim3(VB), 0, c->stream, x, y, n, c->vpart);
    hipLaunchKernelGGL(k_norm2, dim3(1), dim3(VB), 0, c->stream, c->vpart, nb, out2, nout);
    HIPCK(c, hipGetLastError());
    return 0;
}
// the same sums for the HOST: the per-workgroup partial pairs go straight to pinned memory (no second launch, no copy); the host adds
// them in workgroup order (host_diffnorm_sum) once the stream has passed this point.  Returns the number of pairs.
int launch_diffnorm_host(tnml_ctx* c, const double* x, const double* y, size_t n, double* part_host, int cap_pairs) {
    ProfScope ps(c, KC_VEC);
    const int nb = vec_blocks(n);
    if (nb > cap_pairs) return tnml_fail(c, "diffnorm: %d partial pairs exceed the host block (%d)", nb, cap_pairs);
    hipLaunchKernelGGL(k_diffnorm1, dim3(nb), dim3(VB), 0, c->stream, x, y, n, part_host);
    HIPCK(c, hipGetLastError());
    c->last_dn_pairs = nb;
    return 0;
}

// ---- replica fingerprint (multi-rank runs) ---------------------------------------------------
// acc[0] += salt * sum_i bits(x_i) * (2 i + 1)  (mod 2^64): order-independent inside the kernel, position- and
// bit-sensitive, so two replicas of a tensor agree on it iff they are bit-identical (up to 2^-64 collisions).
// acc[1] is kept as ~acc[0]: max-all-reducing the pair over ranks gives [max h, ~min h].
__global__ __launch_bounds__(1024) void k_fingerprint(const double* __restrict__ x, size_t n, unsigned long long salt,
                                                      unsigned long long* __restrict__ acc, int reset) {
    __shared__ unsigned long long sh[16];
    unsigned long long h = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) h += (unsigned long long)__double_as_longlong(x[i]) * (2ull * i + 1ull);
    for (int o = 32; o >= 1; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
        const unsigned long long v = (reset ? 0ull : acc[0]) + t * salt;
        acc[0] = v; acc[1] = ~v;
    }
}
int launch_fingerprint(tnml_ctx* c, const double* x, size_t n, unsigned long long salt, unsigned long long* acc, bool reset) {
    hipLaunchKernelGGL(k_fingerprint, dim3(1), dim3(1024), 0, c->stream, x, n, salt | 1ull, acc, reset ? 1 : 0);
    HIPCK(c, hipGetLastError());
    return 0;
}

// The 64-bit fingerprint h as four 16-bit pieces p_i and their squares: a SUM all-reduce over R ranks gives S_i = sum p_i and Q_i = sum p_i^2
// exactly (S_i <= R 65535, Q_i <= R 2^32: integers far below 2^53), and all ranks hold the same h iff R Q_i == S_i^2 for every piece
// (Cauchy-Schwarz with equality) -- so the check rides in the packed sum all-reduce instead of a max-reduce of its own.
__global__ void k_fingerprint_pieces(const unsigned long long* __restrict__ acc, double* __restrict__ out8) {
    const int i = threadIdx.x;
    if (i < 4) { const double p = (double)((acc[0] >> (16 * i)) & 0xffffull); out8[i] = p; out8[4 + i] = p * p; }
}
int launch_fingerprint_pieces(tnml_ctx* c, const unsigned long long* acc, double* out8) {
    hipLaunchKernelGGL(k_fingerprint_pieces, dim3(1), dim3(64), 0, c->stream, acc, out8);
    HIPCK(c, hipGetLastError());
    return 0;
}

// test hook (tnml_set_option "debug_nudge_rank"): the first element of a replicated tensor moves by one ulp
__global__ void k_nudge(double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = __longlong_as_double(__double_as_longlong(p[0]) + 1); }
int launch_nudge(tnml_ctx* c, double* p) {
    hipLaunchKernelGGL(k_nudge, dim3(1), dim3(64), 0, c->stream, p);
    HIPCK(c, hipGetLastError());
    return 0;
}
