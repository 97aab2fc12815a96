// kernels_small.hip -- layout conversion, bond-tensor product and the replicated CG vector algebra.
//
// These touch only O(m^2) data (<= 576k elements) and are latency items, not roofline items.
// The CG scalar algebra of cgrad (fixedL.cc:386-388,403-407,422-428,442) runs on device scalars in
// fp64 so that no host round trip sits inside a CG pass.
#include "tnml_internal.h"

// ---- pack / unpack -------------------------------------------------------------------------
__global__ void k_pack(PackDesc d, const double* __restrict__ T, double* __restrict__ Md, float* __restrict__ Mf) {
    const size_t per = (size_t)d.Kp * d.Np, total = per * d.L;
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

int launch_pack(tnml_ctx* c, const PackDesc& d, const double* T, double* Md, float* Mf) {
    ProfScope ps(c, KC_PACK);
    hipLaunchKernelGGL(k_pack, dim3(nblocks((size_t)d.Kp * d.Np * d.L)), dim3(256), 0, c->stream, d, T, Md, Mf);
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
    const size_t total = (size_t)A1.ml * 4 * A2.mr * LB;
    hipLaunchKernelGGL(k_bond_form, dim3(nblocks(total)), dim3(256), 0, c->stream, A1.a, A2.a, B, A1.ml, A1.mr, A2.mr, A1.L, A2.L);
    HIPCK(c, hipGetLastError());
    return 0;
}

// ---- block reductions (fp64, fixed order) -----------------------------------------------------
#define VB 1024
static __device__ __forceinline__ double block_sum(double v, double* sh) {
    // deterministic: every thread writes its partial, thread 0.. tree over fixed layout
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = VB / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// r = G - lambda*B (fixedL.cc:385-386); p = r (:388); RR = |r|^2
__global__ __launch_bounds__(VB) void k_cg_init(const double* __restrict__ G, const double* __restrict__ B, double* __restrict__ R,
                                               double* __restrict__ Pv, size_t n, double lambda, double* __restrict__ scal) {
    __shared__ double sh[VB];
    double acc = 0.;
    for (size_t i = threadIdx.x; i < n; i += VB) {
        double r = G[i];
        if (lambda != 0.) r = r - lambda * B[i];
        R[i] = r; Pv[i] = r;
        acc += r * r;
    }
    const double rr = block_sum(acc, sh);
    if (threadIdx.x == 0) scal[SC_RR] = rr;
}
// pAp = sum_n|p v_n|^2 + lambda|p|^2 (:402-403); a = |r|^2/pAp (:405); B = B + a p (:406)
__global__ __launch_bounds__(VB) void k_cg_step(double* __restrict__ B, const double* __restrict__ Pv, size_t n, double lambda,
                                               const double* __restrict__ tail, double* __restrict__ scal) {
    __shared__ double sh[VB];
    double acc = 0.;
    for (size_t i = threadIdx.x; i < n; i += VB) acc += Pv[i] * Pv[i];
    const double pn2 = block_sum(acc, sh);
    const double pAp = tail[SC_PP] + lambda * pn2;
    const double a = scal[SC_RR] / pAp;
    for (size_t i = threadIdx.x; i < n; i += VB) B[i] = B[i] + a * Pv[i];
    if (threadIdx.x == 0) { scal[SC_PNORM2] = pn2; scal[SC_PAP] = pAp; scal[SC_ALPHA] = a; }
}
// nr = G - lambda B (:421-422); beta = sqr(norm(nr)/norm(r)) (:423); r = nr (:424);
// C = sum dP^2 + lambda|B|^2 (:427-428); conv = |r| < cconv (:432); p = r + beta p (:442)
__global__ __launch_bounds__(VB) void k_cg_resid(const double* __restrict__ G, const double* __restrict__ B, double* __restrict__ R,
                                                double* __restrict__ Pv, size_t n, double lambda, double cconv,
                                                const double* __restrict__ tail, double* __restrict__ scal) {
    __shared__ double sh[VB];
    double an = 0., ab = 0.;
    for (size_t i = threadIdx.x; i < n; i += VB) {
        double nr = G[i];
        if (lambda != 0.) nr = nr - lambda * B[i];
        an += nr * nr;
        ab += B[i] * B[i];
    }
    const double nn = block_sum(an, sh);
    const double bn2 = block_sum(ab, sh);
    const double q = sqrt(nn) / sqrt(scal[SC_RR]);
    const double beta = q * q;
    const double rn = sqrt(nn);
    const int conv = rn < cconv;
    for (size_t i = threadIdx.x; i < n; i += VB) {
        double nr = G[i];
        if (lambda != 0.) nr = nr - lambda * B[i];
        R[i] = nr;
        if (!conv) Pv[i] = nr + beta * Pv[i];
    }
    if (threadIdx.x == 0) {
        double cs = 0.;
        for (int l = 0; l < TNML_NL; ++l) cs += tail[SC_COST0 + l];
        scal[SC_COST] = cs + lambda * bn2;
        scal[SC_BNORM2] = bn2; scal[SC_BETA] = beta; scal[SC_RNORM] = rn; scal[SC_CONV] = (double)conv;
    }
    __syncthreads();
    if (threadIdx.x == 0) scal[SC_RR] = nn;
}
__global__ __launch_bounds__(VB) void k_sqnorm(const double* __restrict__ x, size_t n, double* __restrict__ out) {
    __shared__ double sh[VB];
    double acc = 0.;
    for (size_t i = threadIdx.x; i < n; i += VB) acc += x[i] * x[i];
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = s;
}
__global__ __launch_bounds__(VB) void k_diffnorm(const double* __restrict__ x, const double* __restrict__ y, size_t n, double* __restrict__ out) {
    __shared__ double sh[VB];
    double a = 0., d = 0.;
    for (size_t i = threadIdx.x; i < n; i += VB) { a += x[i] * x[i]; const double t = x[i] - y[i]; d += t * t; }
    const double s1 = block_sum(a, sh);
    const double s2 = block_sum(d, sh);
    if (threadIdx.x == 0) { out[0] = s1; out[1] = s2; }
}

int launch_cg_init(tnml_ctx* c, size_t n, double lambda) {
    ProfScope ps(c, KC_VEC);
    hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(VB), 0, c->stream, c->vG, c->vB, c->vR, c->vP, n, lambda, c->scal);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_cg_step(tnml_ctx* c, size_t n, double lambda) {
    ProfScope ps(c, KC_VEC);
    hipLaunchKernelGGL(k_cg_step, dim3(1), dim3(VB), 0, c->stream, c->vB, c->vP, n, lambda, c->vG + n, c->scal);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_cg_resid(tnml_ctx* c, size_t n, double lambda, double cconv) {
    ProfScope ps(c, KC_VEC);
    hipLaunchKernelGGL(k_cg_resid, dim3(1), dim3(VB), 0, c->stream, c->vG, c->vB, c->vR, c->vP, n, lambda, cconv, c->vG + n, c->scal);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_sqnorm(tnml_ctx* c, const double* x, size_t n, double* out) {
    ProfScope ps(c, KC_VEC);
    hipLaunchKernelGGL(k_sqnorm, dim3(1), dim3(VB), 0, c->stream, x, n, out);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_diffnorm(tnml_ctx* c, const double* x, const double* y, size_t n, double* out2) {
    ProfScope ps(c, KC_VEC);
    hipLaunchKernelGGL(k_diffnorm, dim3(1), dim3(VB), 0, c->stream, x, y, n, out2);
    HIPCK(c, hipGetLastError());
    return 0;
}
