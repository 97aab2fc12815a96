// eigh.hip -- eigen-decomposition stages of the bond-tensor split (fixedL.cc:519-521), fp64, n <= 240 in one workgroup
// (larger n: eigh_mc.hip): Householder tridiagonalisation with the matrix resident in registers, the tridiagonal
// eigenproblem (bisection + inverse iteration), back transformation, Cholesky QR and Newton-Schulz helpers.
//
// Why: the SVD of fixedL.cc:519-521 sits on the critical path of every bond update and rocSOLVER's
// dsyevd spends 4.0 of its 5.6 ms at n=240 in ~1700 tiny sytrd kernels (profiles/r01_bench_c3_kernel_stats.csv:
// hemvn, sytd2, dot, syr2, latrd, larfg, set_tau).  The reduction is inherently sequential in n, so it is
// done here by ONE workgroup with the matrix resident in registers.  Two earlier generations of that kernel (16 x 16 blocks
// with four lanes each; 8 x 8 blocks with the column in LDS) were measured against k_sytrd_v3 in rounds 1-2
// (profiles/r02_probe_eigh.txt: 1.20 / 0.70 / 0.62 ms for all 238 steps at n = 240) and are gone.
#include "tnml_internal.h"

#define TRI_MAXN 240

struct TriArgs {
    const double* A; int n; int lda;     // symmetric input (both triangles valid)
    double* D; double* E; double* tau;   // outputs: diagonal n, subdiagonal n-1, tau n-1
    double* V; int ldv;                  // Householder vectors: column k holds v_k (v_k[k+1] = 1, zeros above)
    long long* dbg;                      // TNML_EIGH_PROF builds: per-phase cycle counters
    double* nref;                        // out (may be null): number of reflectors formed (n-1 unless k_sytrd_v3 stopped early)
    double psd_tol;                      // k_sytrd_v3, positive semidefinite input only: stop once trace(trailing block) <= psd_tol * trace(A); 0 = never
};

#define T8 8
#define T8_MAXNB (TRI_MAXN / T8)
static __device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ==========================================================================================
// k_sytrd_v3 -- 8x8 blocks of the lower triangle, one lane per block (64 doubles), enumerated column-block major so that waves
// retire as the active window shrinks; the serial part of a Householder step cut down (profiles/r02_probe_eigh.txt):
//   * the current column lives in REGISTERS of every wave (rows lane + 64 e): the look-ahead column formed at the end of
//     step k is the input of step k+1 without an LDS round trip, alpha comes from a readlane;
//   * Householder scalars from v_rsq_f64 / v_rcp_f64 with explicit Newton steps (one dependent chain of ~25 fp64 ops
//     instead of the ~45 of IEEE sqrt + division), tau = 1 + |alpha|/norm;
//   * the partial sums of y = A v go to a table Y[c][i] (c = block column that produced the partial, i = row; row stride
//     242 doubles): every block writes its 8 row sums and its 8 column sums as 16-byte stores that are bank-conflict
//     free across the lanes of a wave (a [block][8] layout is 4-way conflicted), and the partials of a row are
//     a strided run that TWO lanes per row read with all loads in flight before the first add;
//   * the global stores of v are taken by a different wave every step.
// Two barriers per step; Householder convention of LAPACK dlarfg.
// ==========================================================================================
#define V3_LD 242
// s_v / s_w are read by every lane at the eight rows of its block (row 8 R + r).  V3_PAD = 1 puts one pad double after every eight (stride 72
// bytes: 30 blocks on 30 different bank pairs instead of every fourth bank group) -- measured SLOWER (2.90 against 2.78 us per step,
// profiles/r05_probe_sytrd_padding.txt: the padded rows lose their 16-byte alignment and with it ds_read_b128), so it is off
#ifndef V3_PAD
#define V3_PAD 0
#endif
#define SVI(i) (V3_PAD ? (i) + ((i) >> 3) : (i))
#define V3_VLEN (240 + 32)
#define V3_SMEM_DOUBLES ((T8_MAXNB + 1) * V3_LD + 2 * V3_VLEN + 2 * 240 + 32)
static __device__ __forceinline__ double lane_bcast(double x, int l) {          // l uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(512) void k_sytrd_v3(TriArgs T) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Y = smem;                                   // [nb][V3_LD]
    double* s_v = Y + (T8_MAXNB + 1) * V3_LD;           // [240] at index SVI(i); (row nb of Y stays zero: phase C reads it instead of branching on a short half)
    double* s_w = s_v + V3_VLEN;                        // [240], padded like s_v
    double* s_xo = s_w + V3_VLEN;                           // [2][240]
    double* s_red = s_xo + 480;                         // [32]: 0..7 v^T A v partials, 8 tau, 9 exact-trace flag, 16..23 trailing-trace partials, 24..31 trace(A) partials
    const int n = T.n, nb = (n + T8 - 1) / T8;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nblocks = nb * (nb + 1) / 2;
    const bool owner = tid < nblocks;
    int R = 0, C = 0;
    if (owner) { int cc = 0; while ((cc + 1) * nb - (cc + 1) * cc / 2 <= tid) ++cc; C = cc; R = cc + (tid - (cc * nb - cc * (cc - 1) / 2)); }
    const int i0 = T8 * R, j0 = T8 * C;
    // the last block column held by this wave: once kb passes it the wave has no block left ("retired") and only helps
    // with the row sums of phase C
    int cmax = owner ? C : -1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(cmax, o); cmax = t > cmax ? t : cmax; }
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    double a[T8][T8];
#pragma unroll
    for (int r = 0; r < T8; ++r)
#pragma unroll
        for (int cc = 0; cc < T8; ++cc) {
            const int i = i0 + r, j = j0 + cc;
            a[r][cc] = (owner && i < n && j < n) ? T.A[i + (size_t)T.lda * j] : 0.;
        }
    // the current column lives in the REGISTERS of every live wave as xt = its rows k+2.. (rows lane + 64 e, zero above), its two leading
    // entries d_k and alpha are broadcast once, by the look-ahead that forms the column (dk_n, alpha_n: uniform values carried to the next
    // step), so that phase A starts from them instead of two register selects by uniform branches, two broadcasts and eight masked moves
    double x[4], v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int i = lane + 64 * e; x[e] = (i >= 2 && i < n) ? T.A[i] : 0.; v[e] = 0.; }
    double dk_n = lane_bcast(T.A[0], 0), alpha_n = lane_bcast(n > 1 ? T.A[1] : 0., 0);   // (through readlane: the compiler keeps them in scalar registers)
    //   // d_k and alpha of the coming step (uniform): set by the look-ahead of the step before
    for (int i = tid; i < V3_LD; i += 512) Y[nb * V3_LD + i] = 0.;      // the row behind the last block column: phase C reads it where a lane's half is one short
    if (tid < V3_VLEN) { s_v[tid] = 0.; s_w[tid] = 0.; }
    if (tid < 240) { s_xo[tid] = 0.; s_xo[240 + tid] = 0.; }
    {   // trace(A) (rank-adaptive early exit below)
        double dg = 0.;
        if (owner && R == C) {
#pragma unroll
            for (int r = 0; r < T8; ++r) dg += a[r][r];
        }
        dg = wave_sum(dg);
        if (lane == 0) s_red[24 + wid] = dg;
    }
    // every load of the prologue has landed before the chain starts: on gfx9 stores count in vmcnt too, and a wait the compiler places
    // in the loop for a register of the prologue would, from the second step on, wait for the reflector stores of the step before
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
    __syncthreads();
    double t0 = 0.;
#pragma unroll
    for (int w = 0; w < 8; ++w) t0 += s_red[24 + w];
    // Rank-adaptive exit (psd_tol > 0, positive semidefinite A): the trailing block A_k[k+1:, k+1:] of a PSD matrix is PSD, so
    // every entry of it is bounded by its trace.  The trace follows the recursion t_{k+1} = t_k - d_k (a similarity leaves the
    // trace alone); once that estimate is within 100x of the threshold the diagonal is summed exactly each step, and when the
    // exact trace is <= psd_tol * trace(A) the block is dropped: D, E, tau of the remaining rows are zero and T.nref says how many
    // reflectors exist.  A bond tensor B_old + (a few CG corrections) has numerical rank ~ m + 10 of n = 2m, so its Gram matrix
    // stops after ~half of the n-2 steps of this latency-bound chain (profiles/r02_sytrd_early_exit.txt).
    const double t_exit = T.psd_tol * t0, t_screen = 100. * t_exit;
    double trem = t0;                                   // trace(A_k[k:, k:]), tracked by the live waves
    int kexit = -1;
#ifdef TNML_EIGH_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64();
    const int pw = T.dbg ? (int)T.dbg[7] : 0;
#define TP3(i) do { if (wid == pw && lane == 0) { long long t_ = clock64(); prof[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define TP3(i) do {} while (0)
#endif
    // row-sum duty of phase C: two lanes per row, lane h takes the first / second half of the block columns kb..nb-1
    const int ci = tid >> 1, ch = tid & 1;
    const int cic = ci < 240 ? ci : 239;
    // d_{k+1} and alpha_{k+1} out of the freshly formed column (row i in register i >> 6 of lane i & 63): uniform selects, two broadcasts
    auto next_scalars = [&](double v0, double v1, double v2, double v3, int k) {
        const int i1 = k + 1, i2 = k + 2;
        const int ea = i1 >> 6, eb = (i2 >> 6) & 3;
        const double va = ea == 0 ? v0 : (ea == 1 ? v1 : (ea == 2 ? v2 : v3));
        const double vb = eb == 0 ? v0 : (eb == 1 ? v1 : (eb == 2 ? v2 : v3));
        dk_n = lane_bcast(va, i1 & 63);
        alpha_n = i2 < n ? lane_bcast(vb, i2 & 63) : 0.;
    };
    for (int k = 0; k < n - 1; ++k) {
        const int par = k & 1;
        const int kb = (k + 1) / T8;
        const bool live = cmax >= kb || wid == 7;       // uniform per wave; wave 7 (holds the last block) is live to the end
        double tau = 0.;
        bool need_exact = false;
        if (live) {
            // ---- A: Householder scalars and v, redundantly per live wave, from the column in registers.  x is zero above
            //      row k; element k is the diagonal entry, element k+1 is alpha.
            const double alpha = alpha_n, dk = dk_n;                 // left by the look-ahead of the last step (or the prologue)
            double xt[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[e] = x[e];                // rows k+2.. of the column
            double sig = fma(xt[0], xt[0], fma(xt[1], xt[1], fma(xt[2], xt[2], xt[3] * xt[3])));
            sig = wave_sum(sig);
            double beta = alpha, scale = 0.;
            if (sig > 0.) {
                const double n2 = fma(alpha, alpha, sig);
                double g, ih, s0;
                const double aa = fabs(alpha);
                if (n2 > 1e-280 && n2 < 1e280) {
                    // sqrt and 1/sqrt by Goldschmidt from v_rsq_f64, reciprocal from v_rcp_f64 + two Newton steps
                    const double y0 = __builtin_amdgcn_rsq(n2);
                    g = n2 * y0; double h = 0.5 * y0;
                    double r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h);
                    r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h);
                    const double d = fma(-g, g, n2); g = fma(d, h, g);
                    ih = h + h;
                    const double ee = fma(-g, ih, 1.0); ih = fma(ih, ee, ih);
                    const double den = aa + g;
                    s0 = __builtin_amdgcn_rcp(den);
                    double e2 = fma(-den, s0, 1.0); s0 = fma(s0, e2, s0);
                    e2 = fma(-den, s0, 1.0); s0 = fma(s0, e2, s0);
                } else { g = sqrt(n2); ih = 1. / g; s0 = 1. / (aa + g); }
                beta = alpha >= 0. ? -g : g;
                tau = fma(aa, ih, 1.0);                               // (beta - alpha)/beta = 1 + |alpha|/norm
                scale = alpha >= 0. ? s0 : -s0;                       // 1/(alpha - beta)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (lane + 64 * e == k + 1) ? 1. : xt[e] * scale;
                if (e < 3 || lane < 48) s_v[SVI(lane + 64 * e)] = v[e];    // identical values from every live wave
            }
            trem -= dk;                                               // trace of rows k+1.. (before and after this step's update)
            need_exact = T.psd_tol > 0. && trem <= t_screen;
            if (lane == 0) { s_red[8] = tau; s_red[9] = need_exact ? 1. : 0.; }   // for the retired waves
            if (wid == 7) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int i = lane + 64 * e; if (i < n) T.V[i + (size_t)T.ldv * k] = v[e]; }
                if (lane == 0) { T.D[k] = dk; T.E[k] = beta; T.tau[k] = tau; }
            }
            wave_lds_fence();
        }
        TP3(0);
        const bool active = owner && C >= kb;
        if (owner && C == kb) {                                       // column k+1 as it is before this step's update
            double* dst = s_xo + par * 240 + i0;
#define V3_STAGE(J) { _Pragma("unroll") for (int r = 0; r < T8; ++r) dst[r] = a[r][J]; } break
            switch ((k + 1) & 7) {
                case 0: V3_STAGE(0); case 1: V3_STAGE(1); case 2: V3_STAGE(2); case 3: V3_STAGE(3);
                case 4: V3_STAGE(4); case 5: V3_STAGE(5); case 6: V3_STAGE(6); default: V3_STAGE(7);
            }
#undef V3_STAGE
        }
        double vI[T8], vJ[T8];
        if (live) {
            double q = 0.;
            if (active && tau != 0.) {
#pragma unroll
                for (int r = 0; r < T8; ++r) vI[r] = s_v[SVI(i0) + r];
#pragma unroll
                for (int cc = 0; cc < T8; ++cc) vJ[cc] = s_v[SVI(j0) + cc];
                double c1[T8];
#pragma unroll
                for (int r = 0; r < T8; ++r) {
                    double t = 0.;
#pragma unroll
                    for (int cc = 0; cc < T8; ++cc) t = fma(a[r][cc], vJ[cc], t);
                    c1[r] = t;
                    q = fma(vI[r], t, q);
                }
                double* y1 = Y + C * V3_LD + i0;                      // partial of rows i0.. from block column C
#pragma unroll
                for (int r = 0; r < T8; r += 2) *reinterpret_cast<double2*>(y1 + r) = make_double2(c1[r], c1[r + 1]);
                if (R != C) {
                    double c2[T8];
#pragma unroll
                    for (int cc = 0; cc < T8; ++cc) {
                        double t = 0.;
#pragma unroll
                        for (int r = 0; r < T8; ++r) t = fma(a[r][cc], vI[r], t);
                        c2[cc] = t;
                    }
                    double* y2 = Y + R * V3_LD + j0;                  // partial of rows j0.. from the transposed block
#pragma unroll
                    for (int cc = 0; cc < T8; cc += 2) *reinterpret_cast<double2*>(y2 + cc) = make_double2(c2[cc], c2[cc + 1]);
                    q *= 2.;
                }
            }
            q = wave_sum(q);
            if (lane == 0) s_red[wid] = q;
            if (need_exact) {                                         // uniform over the live waves
                double dg = 0.;
                if (owner && R == C && C >= kb) {
#pragma unroll
                    for (int r = 0; r < T8; ++r) dg += (i0 + r >= k + 1) ? a[r][r] : 0.;
                }
                dg = wave_sum(dg);
                if (lane == 0) s_red[16 + wid] = dg;
            }
        } else if (lane == 0) { s_red[wid] = 0.; s_red[16 + wid] = 0.; }
        TP3(1);
        __syncthreads();
        TP3(2);
        tau = s_red[8];                                               // uniform over the workgroup
        if (s_red[9] != 0.) {
            double te = 0.;
#pragma unroll
            for (int w = 0; w < 8; ++w) te += s_red[16 + w];
            trem = te;                                                // the exact value replaces the recursion
            if (te <= t_exit) { kexit = k; break; }                   // uniform: reflector k exists, rows k+1.. are dropped
        }
        if (tau != 0.) {
            // ---- C: y_i = sum_c Y[c][i] by two lanes per row, w_i = tau y_i - tau^2/2 (v^T A v) v_i
            {
                double vAv = 0.;
#pragma unroll
                for (int w = 0; w < 8; ++w) vAv += s_red[w];
                const double K = -0.5 * tau * tau * vAv;
                const int nc = nb - kb, half = (nc + 1) >> 1;         // uniform
                const int cbeg = kb + ch * half;                      // (lane 1 of a pair: block columns kb + half .. nb - 1, then the zero row)
                const double* yp = Y + cbeg * V3_LD + cic;
                double y0 = 0., y1 = 0., y2 = 0.;
                int u = 0;
                for (; u + 3 <= half; u += 3) {                       // uniform trip count: a lane whose half is one short reads the zero row nb
                    const double t0 = yp[u * V3_LD], t1 = yp[(u + 1) * V3_LD], t2 = yp[(u + 2) * V3_LD];
                    y0 += t0; y1 += t1; y2 += t2;
                }
                for (; u < half; ++u) y0 += yp[u * V3_LD];
                const double yh = (y0 + y1) + y2;
                const double yo = dpp_quad<0xB1>(yh);                 // the other half of the row (lane ^ 1)
                const double y = ch == 0 ? yh + yo : yo + yh;         // first-half part + second-half part on both lanes
                if (ch == 0 && ci < 240) s_w[SVI(ci)] = (ci > k && ci < n) ? fma(K, s_v[SVI(cic)], tau * y) : 0.;
            }
            TP3(3);
            __syncthreads();
            TP3(4);
            // ---- D: A <- A - v w^T - w v^T
            if (active) {
                double wI[T8], wJ[T8];
#pragma unroll
                for (int r = 0; r < T8; ++r) wI[r] = s_w[SVI(i0) + r];
#pragma unroll
                for (int cc = 0; cc < T8; ++cc) wJ[cc] = s_w[SVI(j0) + cc];
#pragma unroll
                for (int r = 0; r < T8; ++r)
#pragma unroll
                    for (int cc = 0; cc < T8; ++cc) a[r][cc] = fma(-vI[r], wJ[cc], fma(-wI[r], vJ[cc], a[r][cc]));
            }
            if (live) {
                // look-ahead: column k+1 of the updated matrix, into the registers of every live wave
                const double wk1 = s_w[SVI(k + 1)];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = lane + 64 * e;
                    const int ic = i < 240 ? i : 239;
                    const double xo = s_xo[par * 240 + ic], wi = s_w[SVI(ic)];
                    x[e] = (i < n) ? (xo - v[e] * wk1) - wi : 0.;
                }
                next_scalars(x[0], x[1], x[2], x[3], k);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (lane + 64 * e >= k + 3) ? x[e] : 0.;
            }
            TP3(5);
        } else {                                                      // no reflector: the matrix is unchanged
            if (live) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = lane + 64 * e; const int ic = i < 240 ? i : 239;
                    x[e] = i < n ? s_xo[par * 240 + ic] : 0.;
                }
                next_scalars(x[0], x[1], x[2], x[3], k);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (lane + 64 * e >= k + 3) ? x[e] : 0.;
            }
            __syncthreads();
        }
    }
#ifdef TNML_EIGH_PROF
    if (wid == pw && lane == 0 && T.dbg) for (int i = 0; i < 6; ++i) T.dbg[i] = prof[i];
#endif
    if (kexit >= 0) {
        for (int i = kexit + 1 + tid; i < n; i += 512) { T.D[i] = 0.; if (i < n - 1) { T.E[i] = 0.; T.tau[i] = 0.; } }
        if (tid == 0 && T.nref) T.nref[0] = (double)(kexit + 1);
    } else {
        const int kl = n - 1;
        const double dl = dk_n;                                        // d_{n-1}: the look-ahead of the last step (wave 7 is live to the end)
        if (tid == 7 * 64) { T.D[kl] = dl; if (T.nref) T.nref[0] = (double)(n - 1); }   // wave 7 is live to the end
    }
}

// U[:, c] = H_0 H_1 ... H_{n-2} Z[:, c]; one wave per column, NE rows per lane (n <= 64 NE).  The
// reflectors are fetched PF at a time so that the L2 latency of V is paid once per PF dependent updates.
template <int NE, int PF>
__global__ __launch_bounds__(64) void k_backtransform(const double* __restrict__ V, int ldv, const double* __restrict__ tau, int n,
                                                     const double* __restrict__ Z, int ldz, double* __restrict__ U, int ldu, const double* __restrict__ nrefp) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double z[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { const int i = lane + 64 * e; z[e] = i < n ? (Z ? Z[i + (size_t)ldz * c] : (i == c ? 1. : 0.)) : 0.; }   // Z == nullptr: the identity (forms H_0 ... H_{n-2} itself)
    const int nr = nrefp ? (int)nrefp[0] : n - 1;                      // reflectors 0 .. nr-1 exist (the tridiagonalisation may stop early)
    // (a software pipeline over the batches -- the loads of batch b + 1 in flight during batch b's updates -- measured SLOWER in round 5:
    // 36.5 us against 29.6; the second register set costs more than the L2 round trip it hides)
    for (int k0 = nr - 1; k0 >= 0; k0 -= PF) {
        double v[PF][NE], t[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int k = k0 - q;
            t[q] = k >= 0 ? tau[k] : 0.;
#pragma unroll
            for (int e = 0; e < NE; ++e) { const int i = lane + 64 * e; v[q][e] = (k >= 0 && i < n && i > k) ? V[i + (size_t)ldv * k] : 0.; }
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            double dot = 0.;
#pragma unroll
            for (int e = 0; e < NE; ++e) dot = fma(v[q][e], z[e], dot);
            dot = wave_sum(dot);                    // DPP reduction: six ds_bpermute round trips per reflector were 40 % of this kernel
            const double f = t[q] * dot;
#pragma unroll
            for (int e = 0; e < NE; ++e) z[e] -= f * v[q][e];
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) { const int i = lane + 64 * e; if (i < n) U[i + (size_t)ldu * c] = z[e]; }
}

// A (n x n symmetric, device) -> D, E, tau, V on the context's stream.  n <= TRI_MAXN.  `tau` has room for n doubles: tau[n-1]
// receives the number of reflectors formed, which eigh_backtransform reads back on the device.  psd_tol > 0 promises a positive
// semidefinite A (a Gram matrix) and lets k_sytrd_v3 stop once the trailing block's trace is <= psd_tol * trace(A).
int eigh_tridiagonalize(tnml_ctx* c, const double* A, int n, double* D, double* E, double* tau, double* V, double psd_tol) {
    if (n > TRI_MAXN) {                                // the multi-workgroup kernel (eigh_mc.hip)
        if (!c->mc_xbuf) return tnml_fail(c, "eigh_tridiagonalize: n=%d needs the multi-workgroup exchange buffer (context created with maxm <= %d)", n, TRI_MAXN / 2);
        return eigh_mc_tridiagonalize(c, c->stream, A, n, D, E, tau, V, psd_tol, c->mc_xbuf, &c->mc_epoch, nullptr, 0, c->mc_spin_max);
    }
    TriArgs t{A, n, n, D, E, tau, V, n, nullptr, tau + (n - 1), psd_tol};
    if (!c->attr_sytrd) {                              // function attributes are per device: remembered per context, not per process
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_v3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(V3_SMEM_DOUBLES * sizeof(double))) != hipSuccess)
            return tnml_fail(c, "eigh_tridiagonalize: cannot reserve %zu bytes of LDS", V3_SMEM_DOUBLES * sizeof(double));
        c->attr_sytrd = true;
    }
    hipLaunchKernelGGL(k_sytrd_v3, dim3(1), dim3(512), V3_SMEM_DOUBLES * sizeof(double), c->stream, t);
    HIPCK(c, hipGetLastError());
    return 0;
}
int eigh_backtransform(tnml_ctx* c, const double* V, const double* tau, int n, const double* Z, int ldz, double* U, int ldu, int ncols, hipStream_t st) {
    if (!st) st = c->stream;
    if (n <= 256)      hipLaunchKernelGGL((k_backtransform<4, 8>), dim3(ncols), dim3(64), 0, st, V, n, tau, n, Z, ldz, U, ldu, tau + (n - 1));
    else if (n <= 384) hipLaunchKernelGGL((k_backtransform<6, 6>), dim3(ncols), dim3(64), 0, st, V, n, tau, n, Z, ldz, U, ldu, tau + (n - 1));
    else if (n <= 640) hipLaunchKernelGGL((k_backtransform<10, 4>), dim3(ncols), dim3(64), 0, st, V, n, tau, n, Z, ldz, U, ldu, tau + (n - 1));
    else if (n <= 1024) hipLaunchKernelGGL((k_backtransform<16, 4>), dim3(ncols), dim3(64), 0, st, V, n, tau, n, Z, ldz, U, ldu, tau + (n - 1));
    else return tnml_fail(c, "eigh_backtransform: n=%d exceeds 1024", n);
    HIPCK(c, hipGetLastError());
    return 0;
}

// (the tridiagonal eigenproblem -- eigenvalues by multisection, eigenvectors by inverse iteration -- lives in eigh_tri.hip since round 5)

// C = 1.5 I - 0.5 S (Newton-Schulz polish of a nearly orthonormal basis), dev[0] = max |S - I|
// (one workgroup of 1024 lanes, a column per group of lanes: 14 independent loads per lane at m = 120 instead of a 57-deep chain)
__global__ __launch_bounds__(1024) void k_ns_matrix(const double* __restrict__ S, double* __restrict__ Cm, int m, double* __restrict__ dev) {
    __shared__ double sh[16];
    double mx = 0.;
    const int i = threadIdx.x & 127, j0 = threadIdx.x >> 7;               // row, first column (m <= 128: one lane per row; larger m: rows strided)
    for (int ii = i; ii < m; ii += 128)
#pragma unroll 4
        for (int j = j0; j < m; j += 8) {
            const size_t idx = ii + (size_t)m * j;
            const double s = S[idx], id = (ii == j) ? 1. : 0.;
            mx = fmax(mx, fabs(s - id));
            Cm[idx] = 1.5 * id - 0.5 * s;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) { double t = sh[0]; for (int w = 1; w < 16; ++w) t = fmax(t, sh[w]); dev[0] = t; }
}
int eigh_ns_matrix(tnml_ctx* c, const double* S, double* Cm, int m, double* dev) {
    hipLaunchKernelGGL(k_ns_matrix, dim3(1), dim3(1024), 0, c->stream, S, Cm, m, dev);
    HIPCK(c, hipGetLastError());
    return 0;
}

// ==========================================================================================
// Cholesky QR of a basis whose Gram matrix S = Q^T Q is given (m x m): S = L L^T, then Rinv = L^-T written dense
// (upper triangular), so that Q Rinv is orthonormal.  Used because inverse iteration returns, for an eigenvalue
// CLUSTER, vectors (random starts) that span the right invariant subspace but are not orthogonal to each other; any
// orthonormal basis of that subspace is an equally valid set of eigenvectors, and well separated vectors are left alone
// to round-off (their rows of L are ~ e_i).  flag[0] = 1 when a pivot is not safely positive (dependent vectors) --
// the caller then falls back.  (A column-by-column LDS kernel did this in 204 us at m = 120, profiles/r02_ab_chol_panels.txt.)
// ==========================================================================================
// k_chol_rinv_blocked -- S = L L^T, Rinv = L^-T for m <= 128 by a blocked right-looking Cholesky on
// 8 x 8 register tiles, one lane per tile of the lower triangle (16 x 16 lanes), with the inverse accumulated alongside:
// W starts as the identity and takes the same eliminations as the trailing matrix, so that the row block p of
// X = L^-1 is final right after panel p (X_p* = L_pp^-1 W_p*) and no separate triangular inversion pass is needed.
// A tile holds A_ij until its column panel j has been factored and W_ij from then on: 64 doubles per lane throughout.
// Per panel p: (1) lane (p,p) factors its tile and inverts the 8 x 8 triangle; (2) the lanes of column p form
// L_ip = A_ip L_pp^-T, the lanes of row p form X_pj = L_pp^-1 W_pj, both published in LDS; (3) every lane below row
// block p updates its tile with one 8 x 8 x 8 product.  Two barriers per panel, 16 panels: ~45 us at m = 120 against
// 204 us for a column-by-column kernel (two barriers per COLUMN and a separate inversion pass).
// ==========================================================================================
#define CQ_T 8
#define CQ_NT 16
__global__ __launch_bounds__(256) void k_chol_rinv_blocked(const double* __restrict__ S, int m, double* __restrict__ Rinv, double* __restrict__ flag, int all_panels, int zero_prev) {
    __shared__ __attribute__((aligned(16))) double s_li[CQ_T * CQ_T];            // L_pp^-1, row major, zeros above the diagonal
    __shared__ __attribute__((aligned(16))) double Pl[CQ_NT * CQ_T * CQ_T];      // L_ip rows: [row][k]
    __shared__ __attribute__((aligned(16))) double Px[CQ_T * CQ_NT * CQ_T];      // X_p* : [k][column]
    __shared__ int s_fail;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int nt = (m + CQ_T - 1) / CQ_T;
    if (tid == 0) { s_fail = 0; if (zero_prev) flag[-1] = 0.; }      // flag[-1]: the deviation slot of the polish step that follows (an atomic max)
    double a[CQ_T][CQ_T];
#pragma unroll
    for (int r = 0; r < CQ_T; ++r)
#pragma unroll
        for (int cc = 0; cc < CQ_T; ++cc) {
            const int i = CQ_T * ti + r, j = CQ_T * tj + cc;
            a[r][cc] = (i < m && j < m && tj <= ti) ? S[i + (size_t)m * j] : (i == j ? 1. : 0.);
        }
    int np = nt;                                                     // panels that have to be factored
    {   // orthonormal already (max |S - I| < 5e-7): R = I, nothing to factor
        __shared__ double s_dev[4];
        __shared__ int s_np[4];
        double dv = 0.;
#pragma unroll
        for (int r = 0; r < CQ_T; ++r)
#pragma unroll
            for (int cc = 0; cc < CQ_T; ++cc) dv = fmax(dv, fabs(a[r][cc] - ((CQ_T * ti + r == CQ_T * tj + cc) ? 1. : 0.)));
        // The kept basis is ordered "largest eigenvalue first": the vectors of the unreduced block of T come first, the unit vectors
        // of the rows the tridiagonalisation dropped (k_sytrd_v3's rank-adaptive exit) follow, and those are orthonormal and orthogonal to
        // everything else to round-off.  S = [S_main e; e^T I + e'] then, and only the leading panels need factoring: the rest of
        // R^-1 is the identity (the deviation left in is < 5e-7, which the caller's polish step squares away).
        int lp = (tj <= ti && !(dv < 5e-7)) ? ti + 1 : 0;           // NaN counts as non-trivial
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { dv = fmax(dv, __shfl_xor(dv, o)); const int t = __shfl_xor(lp, o); lp = t > lp ? t : lp; }
        if ((tid & 63) == 0) { s_dev[tid >> 6] = dv; s_np[tid >> 6] = lp; }
        __syncthreads();
        dv = fmax(fmax(s_dev[0], s_dev[1]), fmax(s_dev[2], s_dev[3]));
        np = max(max(s_np[0], s_np[1]), max(s_np[2], s_np[3]));
        if (np > nt || all_panels) np = nt;
        if (tid == 0) flag[2] = dv;                                  // max |S - I| of the incoming basis (diagnostic)
        if (dv < 5e-7) {                                             // the caller's Newton-Schulz step takes d to 3/4 d^2 < 2e-13
#pragma unroll
            for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                for (int cc = 0; cc < CQ_T; ++cc) {
                    const int i = CQ_T * ti + r, j = CQ_T * tj + cc;
                    if (i < m && j < m) Rinv[j + (size_t)m * i] = i == j ? 1. : 0.;
                }
            if (tid == 0) { flag[0] = 0.; flag[1] = 0.; }
            return;
        }
    }
    if (ti >= np && tj <= ti) {                                      // rows that need no factoring: R^-1 = I there
#pragma unroll
        for (int r = 0; r < CQ_T; ++r)
#pragma unroll
            for (int cc = 0; cc < CQ_T; ++cc) a[r][cc] = (ti == tj && r == cc) ? 1. : 0.;
    }
    for (int p = 0; p < np; ++p) {
        if (ti == p && tj == p) {
            // (1) Cholesky of the diagonal tile (lower triangle), by its one lane: the serial part of a panel.  Published: L_pp below the
            //     diagonal and the RECIPROCALS of its diagonal on it -- the lanes of phase (2) substitute with L_pp itself, so that the
            //     inversion of the 8 x 8 triangle (another ~200 dependent instructions of this lane) runs in phase (2), beside them
#pragma unroll
            for (int k = 0; k < CQ_T; ++k) {
                double d = a[k][k];
                if (!(d > 1e-13) || !(d < 1e280)) { s_fail = 1; d = 1.; }
                // sqrt and 1/sqrt from v_rsq_f64 by Goldschmidt steps: no IEEE sqrt / division on this single-lane chain
                const double y0 = __builtin_amdgcn_rsq(d);
                double g = d * y0, h = 0.5 * y0;
                double rr = fma(-h, g, 0.5); g = fma(g, rr, g); h = fma(h, rr, h);
                rr = fma(-h, g, 0.5); g = fma(g, rr, g); h = fma(h, rr, h);
                const double dd = fma(-g, g, d); g = fma(dd, h, g);
                double inv = h + h;
                const double e1 = fma(-g, inv, 1.0); inv = fma(inv, e1, inv);
                a[k][k] = inv;                                       // (the diagonal of L itself is not needed again)
#pragma unroll
                for (int r = k + 1; r < CQ_T; ++r) a[r][k] *= inv;
#pragma unroll
                for (int cc = k + 1; cc < CQ_T; ++cc)
#pragma unroll
                    for (int r = cc; r < CQ_T; ++r) a[r][cc] = fma(-a[r][k], a[cc][k], a[r][cc]);
            }
#pragma unroll
            for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                for (int cc = 0; cc < CQ_T; ++cc) { s_li[r * CQ_T + cc] = cc <= r ? a[r][cc] : 0.; a[r][cc] = r == cc ? 1. : 0.; }     // the tile is W_pp = I from here: (2b) turns it into X_pp = L_pp^-1
        }
        __syncthreads();
        if (tj <= ti && ti >= p && (tj == p || ti == p)) {
            double L[CQ_T][CQ_T];                                    // L_pp below the diagonal, 1 / diag(L_pp) on it
#pragma unroll
            for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                for (int cc = 0; cc < CQ_T; ++cc) L[r][cc] = s_li[r * CQ_T + cc];
            double t2[CQ_T][CQ_T];
            if (tj == p && ti > p) {
                // (2a) L_ip = A_ip L_pp^-T (L_ip L_pp^T = A_ip, column by column), published; the tile then becomes
                //      W_ip = -L_ip L_pp^-1 (W_ip L_pp = -L_ip, from the last column back)
#pragma unroll
                for (int cc = 0; cc < CQ_T; ++cc)
#pragma unroll
                    for (int r = 0; r < CQ_T; ++r) {
                        double t = a[r][cc];
#pragma unroll
                        for (int k = 0; k < cc; ++k) t = fma(-t2[r][k], L[cc][k], t);
                        t2[r][cc] = t * L[cc][cc];
                    }
                double* dst = Pl + (size_t)(CQ_T * ti) * CQ_T;
#pragma unroll
                for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                    for (int cc = 0; cc < CQ_T; cc += 2) *reinterpret_cast<double2*>(dst + r * CQ_T + cc) = make_double2(t2[r][cc], t2[r][cc + 1]);
                // ... and its transpose into column block ti of Px: phase (3) then reads L_jp^T (tiles right of panel p) and X_pj (tiles left
                // of it) from the same array in the same form -- one update loop instead of two that a wave with tiles of both kinds ran in turn
#pragma unroll
                for (int k = 0; k < CQ_T; ++k)
#pragma unroll
                    for (int r = 0; r < CQ_T; r += 2) *reinterpret_cast<double2*>(Px + (size_t)k * (CQ_NT * CQ_T) + CQ_T * ti + r) = make_double2(t2[r][k], t2[r + 1][k]);
#pragma unroll
                for (int cc = CQ_T - 1; cc >= 0; --cc)
#pragma unroll
                    for (int r = 0; r < CQ_T; ++r) {
                        double t = -t2[r][cc];
#pragma unroll
                        for (int k = cc + 1; k < CQ_T; ++k) t = fma(-a[r][k], L[k][cc], t);
                        a[r][cc] = t * L[cc][cc];
                    }
            } else {
                // (2b) X_pj = L_pp^-1 W_pj (forward substitution down the rows), published and final; the diagonal tile (W_pp = I) with them
#pragma unroll
                for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                    for (int cc = 0; cc < CQ_T; ++cc) {
                        double t = a[r][cc];
#pragma unroll
                        for (int k = 0; k < r; ++k) t = fma(-L[r][k], t2[k][cc], t);
                        t2[r][cc] = t * L[r][r];
                    }
#pragma unroll
                for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                    for (int cc = 0; cc < CQ_T; cc += 2) {
                        a[r][cc] = t2[r][cc]; a[r][cc + 1] = t2[r][cc + 1];
                        *reinterpret_cast<double2*>(Px + (size_t)r * (CQ_NT * CQ_T) + CQ_T * tj + cc) = make_double2(t2[r][cc], t2[r][cc + 1]);
                    }
            }
        }
        __syncthreads();
        if (ti > p && tj <= ti && tj != p) {
            // (3) A_ij -= L_ip L_jp^T (columns still to be factored) or W_ij -= L_ip X_pj (columns already factored): Px holds L_jp^T / X_pj
            double lrow[CQ_T][CQ_T];
            const double* src = Pl + (size_t)(CQ_T * ti) * CQ_T;
#pragma unroll
            for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                for (int k = 0; k < CQ_T; ++k) lrow[r][k] = src[r * CQ_T + k];
#pragma unroll
            for (int k = 0; k < CQ_T; ++k) {
                double xr[CQ_T];
#pragma unroll
                for (int cc = 0; cc < CQ_T; ++cc) xr[cc] = Px[(size_t)k * (CQ_NT * CQ_T) + CQ_T * tj + cc];
#pragma unroll
                for (int r = 0; r < CQ_T; ++r)
#pragma unroll
                    for (int cc = 0; cc < CQ_T; ++cc) a[r][cc] = fma(-lrow[r][k], xr[cc], a[r][cc]);
            }
        }
    }
    __syncthreads();
    if (s_fail) { if (tid == 0) flag[0] = 1.; return; }
    // Rinv = X^T (upper triangular, dense): lane (i,j), j <= i, holds X_ij; the lanes above the diagonal write the zeros
#pragma unroll
    for (int r = 0; r < CQ_T; ++r)
#pragma unroll
        for (int cc = 0; cc < CQ_T; ++cc) {
            const int i = CQ_T * ti + r, j = CQ_T * tj + cc;
            if (i < m && j < m) {
                if (tj <= ti) Rinv[j + (size_t)m * i] = (tj < ti || cc <= r) ? a[r][cc] : 0.;
                else Rinv[j + (size_t)m * i] = 0.;
            }
        }
    if (tid == 0) { flag[0] = 0.; flag[1] = 1.; }          // flag[1]: a factorisation was needed
}
int eigh_chol_rinv(tnml_ctx* c, const double* S, int m, double* Rinv, double* flag, int zero_prev) {
    if (m > CQ_T * CQ_NT) return tnml_fail(c, "eigh_chol_rinv: m=%d exceeds %d", m, CQ_T * CQ_NT);
    hipLaunchKernelGGL(k_chol_rinv_blocked, dim3(1), dim3(256), 0, c->stream, S, m, Rinv, flag, 0, zero_prev);
    HIPCK(c, hipGetLastError());
    return 0;
}
