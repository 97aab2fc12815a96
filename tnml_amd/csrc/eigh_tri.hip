// eigh_tri.hip -- the tridiagonal eigenproblem of the bond-tensor split (fixedL.cc:519-521), fp64: all eigenvalues by multisection
// on Sturm counts, the eigenvectors of the `mk` largest by inverse iteration inside the unreduced block that owns them.  Two
// launches (round 5; four before: split, eigenvalues, rank, inverse iteration):
//
//   k_teig_values   one wave per row i: finds the unreduced block of row i itself (the cut rule |e_k| <= eps*||T|| needs a
//                   maximum over n row sums: cheaper recomputed by every wave than fetched from a launch of its own), then the
//                   (i - lo)-th smallest eigenvalue of that block by 65-section.  The Sturm recurrence is a pure latency chain
//                   of one wave, so its length in INSTRUCTIONS is the run time: the three-term recurrence is kept to one FMA on
//                   the dependent path (exact zeros need no special case: a zero minor counts as positive and its successor has
//                   the sign opposite to its predecessor either way), the signs are shifted into a bit string (one v_alignbit
//                   per row) and counted with a popcount per 8 rows, the magnitudes are renormalised through the exponent
//                   (v_frexp_exp / v_ldexp) once per 8 rows.  ~6 instructions per row instead of ~13.
//   k_teig_vectors  global rank of every eigenvalue by the whole workgroup (was a launch of its own), then inverse iteration with
//                   one lane per vector (LAPACK dlagtf / dlagts with partial pivoting).  Round 4's kernel spent ~100
//                   instructions and an LDS round trip per row; here a row of the factorisation is ~35 instructions, rows are
//                   processed four at a time from registers (all LDS loads of a group issued before its first use, all stores
//                   after its last), the interchange flag of a row travels in the last mantissa bit of its multiplier (one
//                   ulp of a multiplier is the size of the rounding the factorisation carries anyway), the second super-diagonal is re-derived from that flag, the start vector is generated inside the
//                   first forward sweep, and the result leaves through a coalesced write of the whole workgroup.
//
// The vectors are NOT re-orthogonalised against each other (that is what makes LAPACK's dstein sequential): svd.hip puts the kept
// basis through a Cholesky QR and a Newton-Schulz step, verifies it and falls back to rocSOLVER when the check fails.
#include "tnml_internal.h"

#ifndef TEIG_MAXN
#define TEIG_MAXN 1024
#endif

struct Teig2Args {
    const double* D; const double* E; int n;      // tridiagonal (E has n-1 entries)
    double* W;                                     // out: eigenvalues, ascending
    int mk; double* Z; int ldz;                    // out: eigenvectors of the mk largest (column g = g-th largest)
    // workspace: Es = E with negligible couplings zeroed, mu[i] = eigenvalue owned by row i (the (i - lo[i])-th smallest of its
    // unreduced block), tnb[i] = norm of that block, blo/bhi = block [lo, hi) of row i, src[g] = row owning the g-th largest eigenvalue
    double* Es; double* mu; double* tnb; int* blo; int* bhi; int* src;
    double* Wh;                                    // pinned host mirror of W (may be null): a speculative split hands its eigenvalues over without a copy
};

static __device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
static __device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// Sturm count over rows [lo, hi): number of eigenvalues of the block < x.  s_de[j] = {d_j, e_{j-1}^2} of the block scaled to unit
// norm.  p_j = (d_j - x) p_{j-1} - e_{j-1}^2 p_{j-2}; growth per row <= ~2.5, decay per two rows >= ~1e-32 (couplings below
// eps*||T|| were cut), so a renormalisation every 8 rows keeps everything far inside the exponent range.
static __device__ __forceinline__ int sturm_count8(const double2* __restrict__ s_de, int lo, int hi, double x) {
    double pm2 = 1., pm1 = s_de[lo].x - x;
    unsigned sg = (unsigned)__double2hiint(pm1) >> 31;            // sign string, newest row in bit 0
    int cnt = (int)sg;                                           // p_{-1} = 1 is positive
    int j = lo + 1;
    for (; j + 8 <= hi; j += 8) {
        double2 de[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) de[u] = s_de[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double p = fma(de[u].x - x, pm1, -(de[u].y * pm2));
            sg = __builtin_amdgcn_alignbit(sg, (unsigned)__double2hiint(p), 31);      // (sg << 1) | sign(p)
            pm2 = pm1; pm1 = p;
        }
        cnt += __popc((sg ^ (sg >> 1)) & 0xffu);                 // sign changes among the 9 newest rows
        const int e1 = __builtin_amdgcn_frexp_exp(pm1), e2 = __builtin_amdgcn_frexp_exp(pm2);
        const int e = e1 > e2 ? e1 : e2;
        pm1 = __builtin_amdgcn_ldexp(pm1, -e); pm2 = __builtin_amdgcn_ldexp(pm2, -e);
    }
    int rem = 0;
    for (; j < hi; ++j, ++rem) {
        const double2 de = s_de[j];
        const double p = fma(de.x - x, pm1, -(de.y * pm2));
        sg = __builtin_amdgcn_alignbit(sg, (unsigned)__double2hiint(p), 31);
        pm2 = pm1; pm1 = p;
    }
    cnt += __popc((sg ^ (sg >> 1)) & ((1u << rem) - 1u));
    return cnt;
}

__global__ __launch_bounds__(64) void k_teig_values(Teig2Args T) {
    __shared__ __attribute__((aligned(16))) double2 s_de[TEIG_MAXN];
    __shared__ double s_e[TEIG_MAXN];
    const int lane = threadIdx.x, n = T.n, i = blockIdx.x;
    // ---- the unreduced block of row i.  T splits at couplings |e_k| <= eps*||T|| (a normwise backward-stable perturbation, the size
    // of the error the Gram matrix carries anyway): a trained bond tensor is numerically rank deficient, the tail of its spectrum sits
    // below eps*lambda_max and the tridiagonal form decouples there; eigenvectors of different blocks have disjoint supports.
    for (int k = lane; k < n; k += 64) s_e[k] = k < n - 1 ? T.E[k] : 0.;
    __syncthreads();
    double rs = 0.;
    for (int k = lane; k < n; k += 64) rs = fmax(rs, fabs(T.D[k]) + (k > 0 ? fabs(s_e[k - 1]) : 0.) + fabs(s_e[k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) rs = fmax(rs, __shfl_xor(rs, o));
    const double thr = 2.220446049250313e-16 * rs;
    int plo = 0, phi = n;                                         // lo = 1 + last cut before row i, hi = 1 + first cut at or after row i
    for (int k = lane; k < n; k += 64) {
        const bool cut = (k == n - 1) || !(fabs(s_e[k]) > thr);   // the block ends after row k
        if (cut && k < i) plo = k + 1 > plo ? k + 1 : plo;
        if (cut && k >= i) phi = k + 1 < phi ? k + 1 : phi;
    }
    const int lo_r = __builtin_amdgcn_readfirstlane(wave_max_i(plo)), hi_r = __builtin_amdgcn_readfirstlane(wave_min_i(phi));   // (uniform: scalar loop control below)
    if (lane == 0) {
        T.blo[i] = lo_r; T.bhi[i] = hi_r;
        T.Es[i] = (i == hi_r - 1) ? 0. : s_e[i];
    }
    if (hi_r - lo_r == 1) { if (lane == 0) { T.mu[i] = T.D[i]; T.tnb[i] = fabs(T.D[i]); } return; }
    // ---- Gershgorin interval and norm of the block (the coupling out of the block's last row is cut)
    double gl = 1e300, gu = -1e300, tn = 0.;
    for (int j = lo_r + lane; j < hi_r; j += 64) {
        const double r = (j > lo_r ? fabs(s_e[j - 1]) : 0.) + (j < hi_r - 1 ? fabs(s_e[j]) : 0.);
        const double d = T.D[j];
        gl = fmin(gl, d - r); gu = fmax(gu, d + r);
        tn = fmax(tn, fabs(d) + r);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { gl = fmin(gl, __shfl_xor(gl, o)); gu = fmax(gu, __shfl_xor(gu, o)); tn = fmax(tn, __shfl_xor(tn, o)); }
    if (!(tn > 0.)) tn = 1.;
    const double pad = 2.2e-16 * tn * (hi_r - lo_r) + 1e-300;
    const double itn = 1. / tn;
    double lo = (gl - pad) / tn, hi = (gu + pad) / tn;
    for (int k = lo_r + lane; k < hi_r; k += 64) {
        const double e = k > lo_r ? s_e[k - 1] * itn : 0.;
        s_de[k] = make_double2(T.D[k] * itn, e * e);
    }
    __syncthreads();
    // ---- the (i - lo_r)-th smallest eigenvalue of the block by 65-section: every lane probes one interior point of the bracket per
    // round, the bracket shrinks 65 x per round (9-10 rounds to fp64 resolution instead of 53 bisections)
    const int li = i - lo_r;
    for (int it = 0; it < 16; ++it) {
        const double w = hi - lo;
        if (!(w > 4.4e-16 * (1. + fmax(fabs(lo), fabs(hi))))) break;
        const double step = w * (1. / 65.);
        const double x = lo + step * (lane + 1);
        const bool below = sturm_count8(s_de, lo_r, hi_r, x) > li;                 // eigenvalue li is below x
        const unsigned long long mask = __ballot(below);
        const int p = mask ? __ffsll((long long)mask) - 1 : 64;                    // first probe above the eigenvalue
        const double nlo = p > 0 ? lo + step * p : lo;
        const double nhi = p < 64 ? lo + step * (p + 1) : hi;
        if (!(nhi > nlo)) break;
        lo = nlo; hi = nhi;
    }
    if (lane == 0) { T.mu[i] = 0.5 * (lo + hi) * tn; T.tnb[i] = tn; }
}

// ---- inverse iteration -----------------------------------------------------------------------------------------------------------
// IV_L vectors per workgroup (16 up to n = 248; 8 / 4 for larger blocks, whose LU factors would not fit in LDS otherwise), one lane
// of wave 0 per vector; the other three waves take part in the ranking before and the write-out after.  LDS: [array][row][lane].
static __device__ __forceinline__ double teig_frcp(double x) {
    double y = __builtin_amdgcn_rcp(x);                       // ~2^-23 relative; two Newton steps
    double e = fma(-x, y, 1.0); y = fma(y, e, y);
    e = fma(-x, y, 1.0); y = fma(y, e, y);
    if (__builtin_expect(!(fabs(x) > 1e-290 && fabs(x) < 1e290), 0)) y = 1. / x;      // outside the fast path's range: IEEE division
    return y;
}
static __device__ __forceinline__ double teig_setflag(double m, bool f) {
    return __hiloint2double(__double2hiint(m), (__double2loint(m) & ~1) | (f ? 1 : 0));
}
static __device__ __forceinline__ bool teig_flag(double m) { return (__double2loint(m) & 1) != 0; }

template <int IV_L>
__global__ __launch_bounds__(256) void k_teig_vectors(Teig2Args T) {
    extern __shared__ __attribute__((aligned(16))) double iv_lds[];
    __shared__ int s_src[16], s_lo[16], s_hi[16];
    __shared__ double s_f[16];
    const int n = T.n, tid = threadIdx.x;
    const int ns = (n + 63) & ~63;
    double* s_d = iv_lds;                 // [ns]
    double* s_e = s_d + ns;               // [ns]  couplings, zero at block ends
    double* s_mu = s_e + ns;              // [ns]
    double* a = s_mu + ns;                // reciprocal pivots          [row][IV_L]
    double* b = a + (size_t)n * IV_L;     // U first superdiagonal
    double* c = b + (size_t)n * IV_L;     // L multipliers, interchange flag in the last mantissa bit
    double* x = c + (size_t)n * IV_L;     // iterate
    for (int k = tid; k < n; k += 256) { s_d[k] = T.D[k]; s_e[k] = T.Es[k]; s_mu[k] = T.mu[k]; }
    if (tid < 16) { s_src[tid] = -1; s_lo[tid] = 0; s_hi[tid] = 0; s_f[tid] = 0.; }
    __syncthreads();
    // ---- global order: row i owns the g-th largest eigenvalue, g = #{j : mu_j > mu_i or (mu_j == mu_i and j < i)}
    const int g0 = blockIdx.x * IV_L;
    for (int i = tid; i < n; i += 256) {
        const double mi = s_mu[i];
        int g = 0;
        for (int j = 0; j < n; ++j) { const double mj = s_mu[j]; g += (mj > mi || (mj == mi && j < i)) ? 1 : 0; }
        if (blockIdx.x == 0) { T.W[n - 1 - g] = mi; T.src[g] = i; if (T.Wh) T.Wh[n - 1 - g] = mi; }
        if (g >= g0 && g < g0 + IV_L) s_src[g - g0] = i;
    }
    __syncthreads();
#define IX(k) ((k) * IV_L + lane)
    if (tid < IV_L && g0 + tid < T.mk) {
        const int lane = tid;
        const int row = s_src[lane];
        const int lo = T.blo[row], hi = T.bhi[row];       // the vector is supported on rows [lo, hi)
        s_lo[lane] = lo; s_hi[lane] = hi;
        if (hi - lo == 1) { x[IX(lo)] = 1.; s_f[lane] = 1.; }
        else {
            const double lam = s_mu[row];
            const double rtiny = 1. / (2.2e-16 * T.tnb[row] + 1e-300);      // reciprocal of the smallest pivot the solves accept
            // ---- LAPACK dlagtf: (T - lam I) = P L U with partial pivoting, rows generated on the fly.  Row k: pivot candidates
            // a_k (running) and c_k = e_k; dlagtf's test |c_k|/scale2 <= |a_k|/scale1 cross-multiplied (no divisions).
            double ak = s_d[lo] - lam, bk = s_e[lo];
            double scale1 = fabs(ak) + fabs(bk);
#define TEIG_FSTEP(CK, DK1, BK1, OA, OB, OC) {                                                                        \
                const double ck_ = (CK), ak1_ = (DK1) - lam, bk1_ = (BK1);   /* c_k != 0 inside an unreduced block */       \
                const double scale2_ = fabs(ck_) + fabs(ak1_) + fabs(bk1_);                                                 \
                const bool sw_ = !(fabs(ck_) * scale1 <= fabs(ak) * scale2_);             /* interchange rows k, k+1 */      \
                const double den_ = sw_ ? ck_ : ak, num_ = sw_ ? ak : ck_;                                                  \
                const double r_ = teig_frcp(den_);                                                                          \
                const double mult_ = num_ * r_;                                                                             \
                (OB) = sw_ ? ak1_ : bk; (OC) = teig_setflag(mult_, sw_);     /* the flag costs the stored copy one ulp: off the dependent path */ \
                (OA) = copysign(fmin(fabs(r_), rtiny), r_);   /* |pivot| < tiny counts as tiny (dlagts job = -1 in spirit) */         \
                const double nak_ = (sw_ ? bk : ak1_) - mult_ * (sw_ ? ak1_ : bk);                                          \
                bk = sw_ ? -mult_ * bk1_ : bk1_;                                                                            \
                ak = nak_; scale1 = scale2_; }
            int k = lo;
            for (; k + 4 <= hi - 1; k += 4) {
                double dd[4], ee[5], oa[4], ob[4], oc[4];
#pragma unroll
                for (int u = 0; u < 5; ++u) ee[u] = s_e[k + u];                   // (s_e[hi - 1] = 0: the block's last coupling is cut)
#pragma unroll
                for (int u = 0; u < 4; ++u) dd[u] = s_d[k + 1 + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) TEIG_FSTEP(ee[u], dd[u], ee[u + 1], oa[u], ob[u], oc[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[IX(k + u)] = oa[u]; b[IX(k + u)] = ob[u]; c[IX(k + u)] = oc[u]; }
            }
            for (; k < hi - 1; ++k) {
                double oa, ob, oc;
                TEIG_FSTEP(s_e[k], s_d[k + 1], s_e[k + 1], oa, ob, oc);
                a[IX(k)] = oa; b[IX(k)] = ob; c[IX(k)] = oc;
            }
#undef TEIG_FSTEP
            { const double r = teig_frcp(ak); a[IX(hi - 1)] = copysign(fmin(fabs(r), rtiny), r); }
            b[IX(hi - 1)] = 0.; c[IX(hi - 1)] = 0.;
            // ---- two sweeps from a deterministic pseudo-random start in (-1, 1), different for every vector (the shift is exact to
            // round-off: two suffice)
            unsigned int seed = 12345u + 7919u * (unsigned)(g0 + lane);
            auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) * (1.0 / 8388608.0)) - 1.0; };
            double xscale = 1.;                                     // max-norm scaling of the iterate, applied when the next pass reads it
            double nrm2 = 0.;
            for (int iter = 0; iter < 2; ++iter) {
                // forward: apply (P L)^-1 (the interchange is a select: the lanes of a wave pivot differently)
                double xk = iter == 0 ? rnd() : x[IX(lo)] * xscale;
                k = lo;
                for (; k + 4 <= hi - 1; k += 4) {
                    double xs[4], ms[4], out[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { xs[u] = iter == 0 ? rnd() : x[IX(k + 1 + u)] * xscale; ms[u] = c[IX(k + u)]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool sw = teig_flag(ms[u]);
                        const double keep = sw ? xs[u] : xk, go = sw ? xk : xs[u];
                        out[u] = keep;
                        xk = go - ms[u] * keep;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[IX(k + u)] = out[u];
                }
                for (; k < hi - 1; ++k) {
                    const double xk1 = iter == 0 ? rnd() : x[IX(k + 1)] * xscale, m = c[IX(k)];
                    const bool sw = teig_flag(m);
                    const double keep = sw ? xk1 : xk, go = sw ? xk : xk1;
                    x[IX(k)] = keep;
                    xk = go - m * keep;
                }
                x[IX(hi - 1)] = xk;
                // back substitution with U (second super-diagonal: e_{k+1} where rows k, k+1 were interchanged)
                double xn1 = 0., xn2 = 0., vmax = 0.;
                nrm2 = 0.;
                k = hi - 1;
                for (; k - 3 >= lo; k -= 4) {
                    double xv[4], bv[4], av[4], dv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xv[u] = x[IX(k - u)]; bv[u] = b[IX(k - u)]; av[u] = a[IX(k - u)];
                        dv[u] = teig_flag(c[IX(k - u)]) ? s_e[k - u + 1] : 0.;     // (row hi - 1: c = 0, no flag; s_e is padded to ns >= n)
                    }
                    double tv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double t = (xv[u] - bv[u] * xn1 - dv[u] * xn2) * av[u];
                        tv[u] = t; xn2 = xn1; xn1 = t;
                        vmax = fmax(vmax, fabs(t)); nrm2 = fma(t, t, nrm2);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[IX(k - u)] = tv[u];
                }
                for (; k >= lo; --k) {
                    const double d2 = teig_flag(c[IX(k)]) ? s_e[k + 1] : 0.;
                    const double t = (x[IX(k)] - b[IX(k)] * xn1 - d2 * xn2) * a[IX(k)];
                    x[IX(k)] = t; xn2 = xn1; xn1 = t;
                    vmax = fmax(vmax, fabs(t)); nrm2 = fma(t, t, nrm2);
                }
                xscale = vmax > 0. ? 1. / vmax : 1.;               // keeps the iterates in range
            }
            // |x * xscale|^2 = nrm2 * xscale^2 may leave the range when taken apart: scale first
            const double nr = sqrt(nrm2) * xscale;                 // (sqrt(nrm2) <= sqrt(n) vmax: in range whenever vmax is)
            s_f[lane] = nr > 0. && nr < 1e300 ? xscale / nr : 0.;
        }
    }
#undef IX
    __syncthreads();
    // ---- Z[:, g0 + v] = f_v x[:, v] inside the block, 0 outside: the whole workgroup, IV_L columns x 256 / IV_L rows per pass
    {
        const int v = tid % IV_L, kk = tid / IV_L;
        const int g = g0 + v;
        if (g < T.mk) {
            const int lo = s_lo[v], hi = s_hi[v];
            const double f = s_f[v];
            double* zc = T.Z + (size_t)T.ldz * g;
            for (int k = kk; k < n; k += 256 / IV_L) zc[k] = (k >= lo && k < hi) ? x[k * IV_L + v] * f : 0.;
        }
    }
}

// scratch: TEIG_SCRATCH_DOUBLES doubles
int eigh_tridiag_eig(tnml_ctx* c, const double* D, const double* E, int n, double* W, int mk, double* Z, int ldz, double* scratch, double* W_host) {
    if (n > TEIG_MAXN || mk > n) return tnml_fail(c, "eigh_tridiag_eig: n=%d mk=%d exceed %d", n, mk, TEIG_MAXN);
    int* is = (int*)(scratch + 3 * TEIG_MAXN);
    Teig2Args t{D, E, n, W, mk, Z, ldz, scratch, scratch + TEIG_MAXN, scratch + 2 * TEIG_MAXN, is, is + TEIG_MAXN, is + 2 * TEIG_MAXN, W_host};
    static_assert(3 * TEIG_MAXN + (3 * TEIG_MAXN + 1) / 2 <= TEIG_SCRATCH_DOUBLES, "scratch of the tridiagonal eigensolver");
    hipLaunchKernelGGL(k_teig_values, dim3(n), dim3(64), 0, c->stream, t);
    const int ns = (n + 63) & ~63;
    const int ivl = n <= 248 ? 16 : (n <= 448 ? 8 : 4);        // (16 vectors of 249..256 rows would need more than 160 KB)
    const size_t lds = sizeof(double) * (3 * (size_t)ns + (size_t)4 * n * ivl);
    if (lds > 158 * 1024) return tnml_fail(c, "eigh_tridiag_eig: %zu bytes of LDS for n=%d", lds, n);
    if (!c->attr_invit) {
        (void)hipFuncSetAttribute((const void*)k_teig_vectors<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        (void)hipFuncSetAttribute((const void*)k_teig_vectors<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        (void)hipFuncSetAttribute((const void*)k_teig_vectors<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        c->attr_invit = true;
    }
    const dim3 grid((mk + ivl - 1) / ivl);
    if (ivl == 16)     hipLaunchKernelGGL((k_teig_vectors<16>), grid, dim3(256), lds, c->stream, t);
    else if (ivl == 8) hipLaunchKernelGGL((k_teig_vectors<8>), grid, dim3(256), lds, c->stream, t);
    else               hipLaunchKernelGGL((k_teig_vectors<4>), grid, dim3(256), lds, c->stream, t);
    HIPCK(c, hipGetLastError());
    return 0;
}
