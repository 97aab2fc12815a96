// eigh_mc.hip -- Householder tridiagonalisation of the split's Gram matrix for 240 < n <= 1024 (maxm up to 512) on a
// small CLUSTER of workgroups, the matrix resident in registers.
//
// Why: the re-split of a bond tensor at maxm = 300 (BASELINE config 5) needs the eigen-decomposition of a 600 x 600 Gram
// matrix; stock rocsolver_dsyevd takes 11.7 ms there (8 800 latrd launches of 3-4 us each, profiles/r03_prof_m300_before.txt),
// 77 % of a bond update.  The one-workgroup kernel of eigh.hip (k_sytrd_v3) keeps the lower triangle in the registers of
// ONE CU, which holds 240 x 240 and no more.
//
// Here the FULL symmetric matrix is cut into 8 x 8 blocks (one lane per block, as in k_sytrd_v3) and the block ROWS are dealt
// cyclically to P <= 32 workgroups of 512 lanes (13 at n = 600, 20 at n = 800, 32 at n = 1 024: a workgroup holds at most 512 blocks).  With whole rows at home y_i = (A v)_i is a local sum, so a
// Householder step costs ONE exchange between the workgroups, and there is no grid barrier (4-5 us each) in it: the payload is
// written as 16-byte granules {lo32, tag, hi32, tag} by one write-through store (global_store_dwordx4 sc0 sc1), the tag encodes
// (launch, step), and a consumer polls the granule itself with agent-scope loads until both tags are the current step's
// (MI355X_MICROARCH.md "handoff-1to1"; tools/probe/probe_xwg2.hip: 1.1-1.4 us per publish + poll for 13-16 workgroups).
// Per step: every workgroup publishes the y of its <= 64 rows and its partial of v^T y; every row thread polls the ONE y it
// needs, forms w_i and the entry of the next column, and the rank-2 update runs on the registers.  The column a step eliminates
// must be known everywhere: every workgroup keeps a copy of the current PANEL of 8 columns in LDS, applies the rank-2 updates
// to it itself (16 FMAs per lane and step), and the owners publish the next raw block column every 8th step.  What every wave
// of every workgroup reads (the scalars, y of row k+1) is published in 8 copies on separate 128-byte lines: one line polled by
// 104 waves at once is served one reader after the other.  Sums run in a fixed order and the Householder scalars are derived
// redundantly from the same bits: results are bit-identical on every workgroup and from run to run (replicas of W on different
// ranks must stay bit-identical).
//
// Measured (tools/probe/probe_mc.hip, profiles/r03_probe_mc_row_ownership.txt): 5.5 us per step at n = 600 (3.3 ms for the full
// chain, 1.9 ms for the 324 reflectors of a rank-320 Gram matrix), 5.0 us at n = 241.  The first version of this file dealt
// block COLUMNS of the lower triangle to 8 workgroups: half the storage, but y needed a reduce-scatter and an all-gather
// (two exchanges, 18 polled loads per lane in the first): 10.7-11 us per step (profiles/r03_probe_mc.txt).  Where a step goes
// now (per-phase cycle counters of the MC_PROF build): the exchange ~45 % (of which ~1 500 cycles are memory latency, the rest
// instruction issue of the polling code at 2 waves per SIMD and waiting for the slowest peer), block products 12 %, rank-2 update
// 13 %, Householder scalars 9 %, barriers the rest.  One XCD for all workgroups, plain stores + L1 invalidates instead of
// write-through, 8-byte atomics instead of 16-byte loads: no gain or slower (profiles/r03_probe_xwg2_one_hop.txt).
//
// Same Householder convention, outputs and rank-adaptive exit (positive semidefinite input) as k_sytrd_v3.
// A poll that does not complete (a workgroup that never got a CU) sets an abort word: every workgroup leaves, the host
// sees status != 0 and falls back to rocSOLVER -- the GPU is never left hanging.
#include "tnml_internal.h"

#define MC_T 8
#define MC_MAXN 1024
#define MC_MAXNB (MC_MAXN / MC_T)
#define MC_PMAX 32                        // workgroups (their scalars are polled by the lanes 0 .. MC_PMAX - 1 of every wave, y of row k + 1 by lane MC_PMAX)
#define MC_THREADS 512
#define MC_LD 66                          // stride of the partial table between block columns (64 local rows + 2)
#define MC_SMEM_DOUBLES (MC_MAXNB * MC_LD + 8 * MC_MAXN + 2 * MC_MAXN + 8 * 64 + 64)
#define MC_XB_U64 (2 * MC_MAXN * 2 + 2 * 8 * MC_PMAX * 16 + 2 * 8 * 16 + MC_MAXN * 16)   // exchange buffer, then [0] abort word, [1] status
#define MC_SPIN_MAX (1 << 19)

typedef unsigned long long mc_u64;
typedef unsigned int mc_u32x4 __attribute__((ext_vector_type(4)));
#ifdef MC_PROF
#define MCP(i) do { if (p == 0 && lane == 0 && wid == pw) { long long t_ = clock64(); prof[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define MCP(i) do {} while (0)
#endif

struct McArgs {
    const double* A; int n; int lda;
    double* D; double* E; double* tau; double* V; int ldv;
    double* nref;                          // out: number of reflectors formed
    double psd_tol;
    mc_u64* xb;                            // exchange buffer (MC_XB_U64), then the abort word and the status word
    unsigned tag0;                         // (launch epoch) * 1024
    int P;
    int xp;                                // MC_PROF builds: 99 no validation of the polled granules, 98 also no publishes, 97 also no polls (timing experiments)
    int spin_max;                          // polls before a waiting thread gives up and aborts the launch
    long long* dbg;                        // MC_PROF builds: per-phase cycle counters of workgroup 0, wave dbg[15]
};

static __device__ __forceinline__ double mc_bcast(double x, int l) {             // l uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
// one double as a self-validating 16-byte granule {lo32, tag, hi32, tag}, written by ONE write-through store: each 8-byte half carries
// its own tag, so the granule needs no atomicity beyond that of an aligned 8-byte word.
static __device__ __forceinline__ void mc_put(mc_u64* p, double v, unsigned tag) {
    const mc_u64 b = (mc_u64)__double_as_longlong(v);
    mc_u32x4 g; g.x = (unsigned)b; g.y = tag; g.z = (unsigned)(b >> 32); g.w = tag;
    // (s_nop: a store of more than 8 bytes followed by a write of its data registers needs one wait state; the compiler cannot see
    // into the asm to insert it)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(g) : "memory");
}
static __device__ __forceinline__ mc_u64 mc_ld(const mc_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(MC_THREADS) void k_sytrd_mc(McArgs T) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ytab = smem;                                  // [MC_MAXNB][MC_LD]: partial of local row lr from block column C
    double* panel = Ytab + MC_MAXNB * MC_LD;                    // [8][MC_MAXN]: columns 8 c .. 8 c + 7 of the current matrix, replicated
    double* s_v = panel + 8 * MC_MAXN;
    double* s_w = s_v + MC_MAXN;
    double* s_raw = s_w + MC_MAXN;                        // [8][64]: this workgroup's rows of the next raw block column
    double* s_red = s_raw + 8 * 64;                       // [64]: 0..7 v^T A v partials, 16..23 trailing-trace partials, 24..31 |x[k+2:]|^2 partials, 32..39 trace(A) partials
    const int n = T.n, nb = (n + MC_T - 1) / MC_T, P = T.P, p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    mc_u64* const ybuf = T.xb;                            // [2][MC_MAXN][2]  y_i, by the owner of row i
    // what EVERY wave of every workgroup reads (the workgroups' scalars, y of row k+1) is published in 8 copies, one per wave index, each
    // workgroup's pair in a 128-byte line of its own: a line polled by 104 waves at once is served one reader after the other
    mc_u64* const sbuf = ybuf + 2 * MC_MAXN * 2;          // [2][8][MC_PMAX][16]  {v^T y partial, trailing-trace partial} per workgroup
    mc_u64* const y1buf = sbuf + 2 * 8 * MC_PMAX * 16;    // [2][8][16]           y of row k+1
    mc_u64* const pbuf = y1buf + 2 * 8 * 16;              // [MC_MAXN][8][2]      raw block column, by rows
    mc_u64* const abortw = T.xb + MC_XB_U64;

    // ---- block ownership: local block rows rl = 0, 1, ... are the global block rows R = p + rl P; lane (rl, C) holds block (R, C)
    const int nbl = p < nb ? (nb - 1 - p) / P + 1 : 0;
    const bool owner = tid < nbl * nb;
    const int rl_ = owner ? tid / nb : 0;
    const int C = owner ? tid - rl_ * nb : 0, R = p + rl_ * P;
    const int i0_ = MC_T * R, j0_ = MC_T * C;
    const int Rmax_wg = nbl > 0 ? p + (nbl - 1) * P : -1;
    const int Rmax_lane = lane < P && lane < nb ? lane + ((nb - 1 - lane) / P) * P : -1;   // last block row of workgroup `lane`: it is active while this is >= kb
    int* const s_flag = reinterpret_cast<int*>(s_red + 48);
    if (tid == 0) *s_flag = 0;

    double a[MC_T][MC_T];
#pragma unroll
    for (int r = 0; r < MC_T; ++r)
#pragma unroll
        for (int cc = 0; cc < MC_T; ++cc) {
            const int i = i0_ + r, j = j0_ + cc;
            a[r][cc] = (owner && i < n && j < n) ? T.A[i + (size_t)T.lda * j] : 0.;
        }
    {   // panel 0, |x[2:]|^2 of column 0 and trace(A), redundantly in every workgroup (fixed order: identical everywhere)
        for (int idx = tid; idx < 8 * MC_MAXN; idx += MC_THREADS) {
            const int j = idx / MC_MAXN, i = idx - j * MC_MAXN;
            panel[idx] = (i < n && j < n) ? T.A[i + (size_t)T.lda * j] : 0.;
        }
        double dg = 0., sg = 0.;
        for (int i = tid; i < MC_MAXN; i += MC_THREADS) {
            const double xi = i < n ? T.A[i] : 0.;
            s_w[i] = 0.; s_v[i] = 0.;
            if (i >= 2) sg = fma(xi, xi, sg);
            if (i < n) dg += T.A[i + (size_t)T.lda * i];
        }
        dg = wave_sum(dg); sg = wave_sum(sg);
        if (lane == 0) { s_red[32 + wid] = dg; s_red[24 + wid] = sg; }
    }
    __syncthreads();
#ifdef MC_PROF
    const bool XP_PUB = T.xp != 98, XP_LD = T.xp != 97;
    const bool XP_NOVAL = T.xp >= 97 && T.xp <= 99;
#else
    const bool XP_PUB = true;
#endif
    double t0 = 0.;
#pragma unroll
    for (int w = 0; w < MC_THREADS / 64; ++w) t0 += s_red[32 + w];
    const double t_exit = T.psd_tol * t0, t_screen = 100. * t_exit;
    double trem = t0;
    int kexit = -1;
    bool aborted = false;
#ifdef MC_PROF
    long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64();
    const int pw = T.dbg ? (int)T.dbg[15] : 0;
#endif

    for (int k = 0; k < n - 1; ++k) {
        const int kb = (k + 1) / MC_T;
        if (Rmax_wg < kb) break;
        int i0 = i0_, j0 = j0_, rl = rl_;
        asm volatile("" : "+v"(i0), "+v"(j0), "+v"(rl));          // the lane's offsets are made opaque once per step: hoisted out of the loop they become ~20 loop-invariant address registers that the allocator spills to scratch
        const unsigned tag = T.tag0 + (unsigned)k + 1u;
        const int par = k & 1;
        const bool newpanel = ((k + 1) & 7) == 0;                  // column k+1 opens the next block column
        const double* const sx = panel + (k & 7) * MC_MAXN;        // column k: sx[k] = d_k, sx[k+1] = alpha, rows k+2.. = the part to eliminate
        double* const sxn = panel + ((k + 1) & 7) * MC_MAXN;
        // ---- A: Householder scalars, by every thread from the same LDS words
        double sig = 0.;
#pragma unroll
        for (int w = 0; w < MC_THREADS / 64; ++w) sig += s_red[24 + w];
        const double alpha = sx[k + 1], dk = sx[k];
        double beta = alpha, scale = 0., tau = 0.;
        if (sig > 0.) {
            const double n2 = fma(alpha, alpha, sig);
            double g, ih, s0;
            const double aa = fabs(alpha);
            if (n2 > 1e-280 && n2 < 1e280) {
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
            tau = fma(aa, ih, 1.0);
            scale = alpha >= 0. ? s0 : -s0;
        }
        trem -= dk;
        const bool need_exact = T.psd_tol > 0. && trem <= t_screen;
#define MC_V(i) ((i) > k + 1 ? sx[(i)] * scale : ((i) == k + 1 ? 1. : 0.))
        if (p == kb % P) {                                          // one (active) workgroup writes the outputs of the step
            for (int i = tid; i < n; i += MC_THREADS) T.V[i + (size_t)T.ldv * k] = MC_V(i);
            if (tid == 0) { T.D[k] = dk; T.E[k] = beta; T.tau[k] = tau; }
        }
        MCP(0);
        // ---- B: partial sums of y = A v over this lane's block, for the lane's 8 rows
        const bool active = owner && R >= kb && C >= kb;
        {
            double q = 0., dg = 0.;
            if (active) {
                double vJ[MC_T], vI[MC_T];
#pragma unroll
                for (int cc = 0; cc < MC_T; cc += 2) { const double2 t2 = *reinterpret_cast<const double2*>(sx + j0 + cc); vJ[cc] = t2.x * scale; vJ[cc + 1] = t2.y * scale; }
#pragma unroll
                for (int r = 0; r < MC_T; r += 2) { const double2 t2 = *reinterpret_cast<const double2*>(sx + i0 + r); vI[r] = t2.x * scale; vI[r + 1] = t2.y * scale; }
                if (j0 <= k + 1) {                                   // the block column that holds column k+1: v = 0 above it, 1 on it
#pragma unroll
                    for (int cc = 0; cc < MC_T; ++cc) vJ[cc] = j0 + cc > k + 1 ? vJ[cc] : (j0 + cc == k + 1 ? 1. : 0.);
                }
                if (i0 <= k + 1) {
#pragma unroll
                    for (int r = 0; r < MC_T; ++r) vI[r] = i0 + r > k + 1 ? vI[r] : (i0 + r == k + 1 ? 1. : 0.);
                }
                double* y1 = Ytab + (j0 >> 3) * MC_LD + rl * MC_T;
#pragma unroll
                for (int r = 0; r < MC_T; r += 2) {
                    double ta = 0., tb = 0.;
#pragma unroll
                    for (int cc = 0; cc < MC_T; ++cc) { ta = fma(a[r][cc], vJ[cc], ta); tb = fma(a[r + 1][cc], vJ[cc], tb); }
                    *reinterpret_cast<double2*>(y1 + r) = make_double2(ta, tb);
                    q = fma(vI[r], ta, q); q = fma(vI[r + 1], tb, q);
                }
                if (need_exact && i0 == j0) {
#pragma unroll
                    for (int r = 0; r < MC_T; ++r) dg += (i0 + r >= k + 1) ? a[r][r] : 0.;
                }
                if (newpanel && (j0 >> 3) == kb) {                  // the next block column as it is before this step's update
                    double* dst = s_raw + rl * MC_T;
#pragma unroll
                    for (int cc = 0; cc < MC_T; ++cc)
#pragma unroll
                        for (int r = 0; r < MC_T; ++r) dst[cc * 64 + r] = a[r][cc];
                }
            }
            q = wave_sum(q);
            if (need_exact) dg = wave_sum(dg);
            if (lane == 0) { s_red[wid] = q; s_red[16 + wid] = dg; }
        }
        MCP(1);
        __syncthreads();
        MCP(2);
        // ---- C1: the y of this workgroup's rows (8 lanes per row over the block columns kb.., fixed order), published with the step tag
        if (tid >= MC_THREADS - 8) {                                 // copy c by the c-th of the last 8 threads
            double qs = 0., ds = 0.;
#pragma unroll
            for (int w = 0; w < MC_THREADS / 64; ++w) { qs += s_red[w]; ds += s_red[16 + w]; }
            mc_u64* dst = sbuf + (((size_t)par * 8 + (tid - (MC_THREADS - 8))) * MC_PMAX + p) * 16;
            if (XP_PUB) mc_put(dst, qs, tag);
            if (XP_PUB) mc_put(dst + 2, need_exact ? ds : 0., tag);
        }
        {
            const int lr = tid >> 3, h = tid & 7;
            const int Rr = p + (lr >> 3) * P, i = MC_T * Rr + (lr & 7);
            const bool on = (lr >> 3) < nbl && Rr >= kb && i > k && i < n;
            double part = 0.;
            if (on) {
                const double* yr = Ytab + lr;
                int Cc = kb + h;
                for (; Cc + 24 < nb; Cc += 32) {                     // four partials per trip in flight (same order of addition)
                    const double t0_ = yr[Cc * MC_LD], t1_ = yr[(Cc + 8) * MC_LD], t2_ = yr[(Cc + 16) * MC_LD], t3_ = yr[(Cc + 24) * MC_LD];
                    part += t0_; part += t1_; part += t2_; part += t3_;
                }
                for (; Cc < nb; Cc += 8) part += yr[Cc * MC_LD];
            }
            part += dpp_quad<0xB1>(part);
            part += dpp_quad<0x4E>(part);
            part += __shfl_xor(part, 4);
            if (on && h == 0) if (XP_PUB) mc_put(ybuf + ((size_t)par * MC_MAXN + i) * 2, part, tag);
            if (on && i == k + 1) if (XP_PUB) mc_put(y1buf + ((size_t)par * 8 + h) * 16, part, tag);      // all 8 lanes of the row hold the sum
        }
        if (newpanel) {                                              // 8 columns x 64 local rows: one value per thread
            const int j = tid >> 6, lr = tid & 63;
            const int Rr = p + (lr >> 3) * P, i = MC_T * Rr + (lr & 7);
            if ((lr >> 3) < nbl && Rr >= kb && i > k && i < n) if (XP_PUB) mc_put(pbuf + ((size_t)i * 8 + j) * 2, s_raw[j * 64 + lr], tag);
        }
        MCP(3);
        // ---- C2: every row thread polls the granule of its row (and, when a block column opens, its 8 raw entries); the first 16 lanes of
        //      every wave also poll the workgroups' scalars, lane 16 the y of row k+1.  16-byte loads issued back to back from one asm
        //      block: a granule pair of 8-byte agent-scope atomic loads took twice as long per attempt.
        double vAv = 0., te = 0., K = 0., wk1 = 0., sgn = 0.;
        const int nrow = n - 1 - k;
#pragma unroll 1
        for (int tt = 0; tt < nrow || tt == 0; tt += MC_THREADS) {
            const int t = tt + tid, i = k + 1 + t;
            const bool row = i < n;
            const bool first = tt == 0;
            const bool lsc = first && Rmax_lane >= kb;
            const bool ly1 = first && lane == MC_PMAX;
            const mc_u64* ssrc = lane < MC_PMAX ? sbuf + (((size_t)par * 8 + wid) * MC_PMAX + lane) * 16 : y1buf + ((size_t)par * 8 + wid) * 16;
            const mc_u64* rsrc = ybuf + ((size_t)par * MC_MAXN + (row ? i : k + 1)) * 2;
            const mc_u64* psrc = pbuf + (size_t)(row ? i : k + 1) * 16;
            const unsigned long long scmask = __ballot(lsc || ly1);
            mc_u32x4 gr, ga, gb;
            ga.x = ga.y = ga.z = ga.w = 0; gb = ga; gr = ga;
            int spin = 0;
            for (;;) {
#ifdef MC_PROF
                if (XP_LD)
#endif
                {
                    unsigned long long sv;
                    asm volatile("s_mov_b64 %[sv], exec\n\t"
                                 "s_and_b64 exec, exec, %[m]\n\t"
                                 "global_load_dwordx4 %[g1], %[a1], off sc1\n\t"
                                 "global_load_dwordx4 %[g2], %[a1], off offset:16 sc1\n\t"
                                 "s_mov_b64 exec, %[sv]\n\t"
                                 "global_load_dwordx4 %[g0], %[a0], off sc1\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : [g0] "=&v"(gr), [g1] "+v"(ga), [g2] "+v"(gb), [sv] "=&s"(sv)
                                 : [a0] "v"(rsrc), [a1] "v"(ssrc), [m] "s"(scmask) : "memory");
                }
                bool ok = !row || (gr.y == tag && gr.w == tag);
                if (lsc || ly1) ok = ok && ga.y == tag && ga.w == tag;
                if (lsc) ok = ok && gb.y == tag && gb.w == tag;
#ifdef MC_PROF
                if (XP_NOVAL) ok = true;                    // EXPERIMENT: no validation
#endif
                if (ok) break;
                if (++spin > T.spin_max || ((spin & 255) == 0 && mc_ld(abortw) != 0)) { aborted = true; break; }
            }
#define MC_G2D(g) __hiloint2double((int)(g).z, (int)(g).x)
            const double vi = row ? MC_V(i) : 0.;                    // (before the raw block column overwrites the panel that holds column k)
            double xo = 0.;
            if (newpanel) {
                // the raw block column: 8 more granules of the row, from its owner, in two halves (32 more live registers would spill)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    mc_u32x4 p0, p1, p2, p3;
                    const mc_u64* src = psrc + 8 * half;
                    for (;;) {
                        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                                     "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                                     "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(src) : "memory");
                        bool ok = !row || aborted || (p0.y == tag && p0.w == tag && p1.y == tag && p1.w == tag && p2.y == tag && p2.w == tag && p3.y == tag && p3.w == tag);
#ifdef MC_PROF
                        if (XP_NOVAL) ok = true;
#endif
                        if (ok) break;
                        if (++spin > T.spin_max || ((spin & 255) == 0 && mc_ld(abortw) != 0)) { aborted = true; break; }
                    }
                    if (row && !aborted) {
                        if (half == 0) xo = MC_G2D(p0); else panel[4 * MC_MAXN + i] = MC_G2D(p0);
                        panel[(4 * half + 1) * MC_MAXN + i] = MC_G2D(p1);
                        panel[(4 * half + 2) * MC_MAXN + i] = MC_G2D(p2);
                        panel[(4 * half + 3) * MC_MAXN + i] = MC_G2D(p3);
                    }
                }
            } else if (row) xo = sxn[i];
#ifdef MC_PROF
            if (p == 0 && wid == pw && first) { if (lane == 0) prof[7] += spin; if (lane == 1) prof[8] += spin; if (lane == 20) prof[9] += spin; }
#endif
            if (first) {                                             // the lanes of the wave have reconverged: every lane's granules are in
                const double qv = (lsc || ly1) ? MC_G2D(ga) : 0.;
                double qs = lsc ? qv : 0., ds = lsc ? MC_G2D(gb) : 0.;   // lanes 0..31: a fixed tree over two rows of 16 lanes, the same in every wave of every workgroup
                qs += dpp_quad<0xB1>(qs); ds += dpp_quad<0xB1>(ds);
                qs += dpp_quad<0x4E>(qs); ds += dpp_quad<0x4E>(ds);
                qs += dpp_quad<0x141>(qs); ds += dpp_quad<0x141>(ds);
                qs += dpp_quad<0x140>(qs); ds += dpp_quad<0x140>(ds);
                // (workgroups 0..15) + (workgroups 16..31): up to 16 workgroups the second row holds zeros -- the bits of rounds 3-5
                vAv = mc_bcast(qs, 0) + mc_bcast(qs, 16); te = mc_bcast(ds, 0) + mc_bcast(ds, 16);
                const double yk1 = mc_bcast(qv, MC_PMAX);
                K = -0.5 * tau * tau * vAv;
                wk1 = tau * yk1 + K;                                  // v_{k+1} = 1
            }
            if (row && !aborted) {
                const double y = MC_G2D(gr);
                const double wi = fma(K, vi, tau * y);
                const double xn = (xo - vi * wk1) - wi;              // column k+1 after this step's update
                s_w[i] = wi;
                s_v[i] = vi;
                sxn[i] = xn;
                if (i >= k + 3) sgn = fma(xn, xn, sgn);
            }
#undef MC_G2D
        }
        sgn = wave_sum(sgn);
        if (lane == 0) s_red[24 + wid] = sgn;
        if (tid == 0) { s_w[k] = 0.; s_v[k] = 0.; }                 // row k has left the trailing matrix
        MCP(4);
        if (aborted) { __hip_atomic_store(abortw, (mc_u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *s_flag = 1; }
        __syncthreads();
        if (*s_flag) { aborted = true; break; }
        MCP(5);
        if (need_exact) {
            trem = te;
            if (te <= t_exit) { kexit = k; break; }                  // uniform over all workgroups: reflector k exists, rows k+1.. are dropped
        }
        // ---- D: A <- A - v w^T - w v^T on the registers, and on the other columns of the replicated panel
        if (active) {
            double vJ[MC_T], wJ[MC_T];
#pragma unroll
            for (int cc = 0; cc < MC_T; cc += 2) {
                const double2 tv = *reinterpret_cast<const double2*>(s_v + j0 + cc), tw = *reinterpret_cast<const double2*>(s_w + j0 + cc);
                vJ[cc] = tv.x; vJ[cc + 1] = tv.y; wJ[cc] = tw.x; wJ[cc + 1] = tw.y;
            }
#pragma unroll
            for (int r = 0; r < MC_T; r += 2) {
                const double2 tv = *reinterpret_cast<const double2*>(s_v + i0 + r), tw = *reinterpret_cast<const double2*>(s_w + i0 + r);
#pragma unroll
                for (int cc = 0; cc < MC_T; ++cc) { a[r][cc] = fma(-tv.x, wJ[cc], fma(-tw.x, vJ[cc], a[r][cc])); a[r + 1][cc] = fma(-tv.y, wJ[cc], fma(-tw.y, vJ[cc], a[r + 1][cc])); }
            }
        }
        if ((k + 2) >> 3 == kb) {                                    // columns k+2 .. end of the panel that holds column k+1
            const int c_lo = (k + 2) & 7;
            double vg[MC_T], wg[MC_T];
#pragma unroll
            for (int cc = 0; cc < MC_T; ++cc) { vg[cc] = s_v[MC_T * kb + cc]; wg[cc] = s_w[MC_T * kb + cc]; }   // (uniform addresses: broadcast reads; zero beyond n)
#pragma unroll 1
            for (int i = k + 1 + tid; i < n; i += MC_THREADS) {
                const double vi = s_v[i], wi = s_w[i];
#pragma unroll
                for (int cc = 1; cc < MC_T; ++cc) if (cc >= c_lo) { double* q = panel + cc * MC_MAXN + i; *q = fma(-vi, wg[cc], fma(-wi, vg[cc], *q)); }
            }
        }
        MCP(6);
        // no barrier here: the next step's A and B read panel column k+1 and s_red[24..31] (written before the barrier above); the panel
        // columns updated here are read after its first barrier, s_v / s_w are rewritten after it
    }
#undef MC_V
#ifdef MC_PROF
    if (p == 0 && lane == 0 && wid == pw && T.dbg) for (int i = 0; i < 8; ++i) T.dbg[i] = prof[i];
    if (p == 0 && lane == 1 && wid == pw && T.dbg) T.dbg[8] = prof[8];
    if (p == 0 && lane == 20 && wid == pw && T.dbg) T.dbg[9] = prof[9];
#endif
    if (aborted) { if (tid == 0) __hip_atomic_store(abortw + 1, (mc_u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (kexit >= 0) {
        const int kb = (kexit + 1) / MC_T;
        if (p == kb % P) {
            for (int i = kexit + 1 + tid; i < n; i += MC_THREADS) { T.D[i] = 0.; if (i < n - 1) { T.E[i] = 0.; T.tau[i] = 0.; } }
            if (tid == 0 && T.nref) T.nref[0] = (double)(kexit + 1);
        }
    } else if (p == (nb - 1) % P) {                                   // the workgroup that ran the last step
        if (tid == 0) { T.D[n - 1] = panel[((n - 1) & 7) * MC_MAXN + n - 1]; if (T.nref) T.nref[0] = (double)(n - 1); }
    }
}

// number of workgroups for an n x n matrix: the fewest whose members hold <= 8 block rows (64 rows) and <= MC_THREADS blocks
static int mc_workgroups(int n) {
    const int nb = (n + MC_T - 1) / MC_T;
    for (int P = 1; P <= MC_PMAX; ++P) {
        const int nbl = (nb - 1) / P + 1;
        if (nbl <= 8 && nbl * nb <= MC_THREADS) return P;
    }
    return 0;
}

size_t eigh_mc_xbuf_bytes() { return sizeof(mc_u64) * ((size_t)MC_XB_U64 + 8); }
int eigh_mc_max_n() { return MC_MAXN; }

// A (n x n symmetric, device) -> D, E, tau (tau[n-1] = number of reflectors), V on `st`.  xbuf: eigh_mc_xbuf_bytes() of device memory,
// zeroed once at allocation; *epoch is advanced per launch.  The kernel reports through xbuf's status word (eigh_mc_status_ptr).
// All P <= 32 workgroups must be resident at the same time (they wait for each other): P CUs with 151 KB of LDS each.
int eigh_mc_tridiagonalize(tnml_ctx* c, hipStream_t st, const double* A, int n, double* D, double* E, double* tau, double* V, double psd_tol,
                           void* xbuf, unsigned* epoch, long long* dbg, int xp, int spin_max) {
    if (n > MC_MAXN || n < 3) return tnml_fail(c, "eigh_mc_tridiagonalize: n=%d outside 3..%d", n, MC_MAXN);
    const int P = mc_workgroups(n);
    if (P == 0) return tnml_fail(c, "eigh_mc_tridiagonalize: no workgroup count fits n=%d", n);
    *epoch = (*epoch % 4000000u) + 1u;
    if (spin_max < 0) spin_max = MC_SPIN_MAX;                  // (0: the first failed poll aborts -- the fallback test)
    McArgs t{A, n, n, D, E, tau, V, n, tau + (n - 1), psd_tol, (mc_u64*)xbuf, *epoch * 1024u, P, xp, spin_max, dbg};
    // (per launch: the attribute is per device, a process may drive several, and this kernel runs for milliseconds)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_mc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MC_SMEM_DOUBLES * sizeof(double))) != hipSuccess)
        return tnml_fail(c, "eigh_mc_tridiagonalize: cannot reserve %zu bytes of LDS", MC_SMEM_DOUBLES * sizeof(double));
    hipLaunchKernelGGL(k_sytrd_mc, dim3(P), dim3(MC_THREADS), MC_SMEM_DOUBLES * sizeof(double), st, t);
    HIPCK(c, hipGetLastError());
    return 0;
}
int eigh_mc_workgroups(int n) { return mc_workgroups(n); }
// device address of the status word (non-zero after a launch that gave up waiting for a peer workgroup); the abort word sits 8 bytes in front of it
const void* eigh_mc_status_ptr(const void* xbuf) { return (const mc_u64*)xbuf + (size_t)MC_XB_U64 + 1; }
