// eigh_mc.hip -- Householder tridiagonalisation of the split's Gram matrix for 240 < n <= 640 (maxm up to 320) on a
// small CLUSTER of workgroups, the matrix resident in registers.
//
// Why: the re-split of a bond tensor at maxm = 300 (BASELINE config 5) needs the eigen-decomposition of a 600 x 600 Gram
// matrix; stock rocsolver_dsyevd takes 11.7 ms there (8 800 latrd launches of 3-4 us each, profiles/r03_prof_m300_before.txt),
// 77 % of a bond update.  The one-workgroup kernel of eigh.hip (k_sytrd_v3) keeps the lower triangle in the registers of
// ONE CU, which holds 240 x 240 and no more.  Here the 8 x 8 blocks of the lower triangle are dealt block-column cyclic to
// P <= 8 workgroups (one lane per block, as in k_sytrd_v3); per Householder step every workgroup forms the partial sums of
// y = A v over its blocks, and the partials are exchanged ALL-TO-ALL THROUGH L2/HBM WITHOUT A GRID BARRIER: the payload is
// written as 8-byte granules {32 data bits | 32-bit step tag} with relaxed agent-scope atomics (global_store_dwordx2 sc1),
// a consumer requests all granules it needs at once and spins until every tag is the current step's
// (MI355X_MICROARCH.md "handoff-1to1", data-tagged granules; tools/probe/probe_xwg.hip: 3.7 us per step for 8 workgroups
// at n = 600 with every word checked, against 4-5 us for one grid barrier and two of them per step).  The raw look-ahead
// column k+1 rides in the same exchange, so a step costs ONE hop.  Every workgroup sums the partials in workgroup order
// and derives v, w and the Householder scalars redundantly: results are bit-identical on every workgroup and from run
// to run (replicas of W on different ranks must stay bit-identical).
//
// Same Householder convention, outputs and rank-adaptive exit (positive semidefinite input) as k_sytrd_v3.
// A spin that does not complete (a workgroup that never got a CU) sets an abort word: every workgroup leaves, the host
// sees status != 0 and falls back to rocSOLVER -- the GPU is never left hanging.
#include "tnml_internal.h"

#define MC_T 8
#define MC_MAXN 640
#define MC_MAXNB (MC_MAXN / MC_T)
#define MC_NBL 10                         // block columns per workgroup at most
#define MC_PMAX 8
#define MC_LDY (MC_MAXN + 2)              // row stride of the row-partial table
#define MC_LDC (MC_T * MC_NBL + 2)        // row stride of the column-partial table
#define MC_NE (MC_MAXN / 64)              // vector registers per lane: rows lane + 64 e
#define MC_THREADS 512
#define MC_SLOT 6144                      // u64 per workgroup and parity in the exchange buffer: partial rows [2 n], raw column [2 n], scalars [4], reduced {y, x_raw} [4 n]
#define MC_RED (2 * (2 * MC_MAXN) + 8)
#define MC_SMEM_DOUBLES (MC_NBL * MC_LDY + MC_MAXNB * MC_LDC + 5 * MC_MAXN + 64 + MC_T * MC_THREADS)
#define MC_SPIN_MAX (1 << 19)

typedef unsigned long long mc_u64;
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
    mc_u64* xb;                            // exchange buffer [2][P][MC_SLOT], then [0] abort word, [1] status
    unsigned tag0;                         // (launch epoch) * 1024
    int P;
    int same_xcd;                          // 1: the grid is 8 P workgroups of which those with blockIdx % 8 == 0 work (all on one XCD, verified by the caller's probe): hand-offs through that XCD's L2
    int nap_first, nap_retry;              // back-off of the pollers, in units of 16 * 64 cycles
    int spin_max;                          // polls before a waiting thread gives up and aborts the launch
    long long* dbg;                        // MC_PROF builds: per-phase cycle counters of workgroup 0, wave dbg[15]
};

static __device__ __forceinline__ double mc_bcast(double x, int l) {             // l uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
// one double as two self-validating 8-byte granules {data half | tag}, written by ONE 16-byte write-through store: 8-byte sc1 stores are
// one fabric write each and cost 2.7x the time per byte of a 16-byte store (MI355X_MICROARCH.md), and the publish of a step is ~2 000 of
// them per workgroup.  Each half carries its own tag, so the pair needs no atomicity beyond that of an aligned 8-byte word.
typedef unsigned int mc_u32x4 __attribute__((ext_vector_type(4)));
template <bool L2ONLY>
static __device__ __forceinline__ void mc_put_t(mc_u64* p, double v, unsigned tag) {
    const mc_u64 b = (mc_u64)__double_as_longlong(v);
    mc_u32x4 g; g.x = (unsigned)b; g.y = tag; g.z = (unsigned)(b >> 32); g.w = tag;
    if (L2ONLY) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(g) : "memory"); return; }   // stays in the XCD's L2 (the L1 is write-through): a same-XCD sc1 load is served from there
    // (s_nop: a store of more than 8 bytes followed by a write of its data registers needs one wait state; the compiler cannot see
    // into the asm to insert it)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(g) : "memory");
}
#define mc_put(p, v, tag) do { if (T.same_xcd) mc_put_t<true>((p), (v), (tag)); else mc_put_t<false>((p), (v), (tag)); } while (0)
static __device__ __forceinline__ mc_u64 mc_ld(const mc_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(MC_THREADS) void k_sytrd_mc(McArgs T) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Yrow = smem;                                  // [MC_NBL][MC_LDY]   partials of rows i from local block column c
    double* Ycol = Yrow + MC_NBL * MC_LDY;                // [MC_MAXNB][MC_LDC] partials of the rows of a local block column from block row R
    double* s_x2 = Ycol + MC_MAXNB * MC_LDC;              // [2][MC_MAXN]: column k of the current matrix (rows k..), by step parity
    double* s_w = s_x2 + 2 * MC_MAXN;
    double* s_raw = s_w + MC_MAXN;                        // raw column k+1 staged by its owner
    double* s_red = s_raw + MC_MAXN;                      // [64]: 0..7 v^T A v partials, 16..23 trailing-trace partials, 24..31 |x[k+2:]|^2 partials, 32..39 trace(A) partials
    double* s_v = s_red + 64 + MC_T * MC_THREADS;        // [MC_MAXN]: v of the step for the rank-2 update (written with w during the exchange)
    double* s_a7 = s_red + 64;                            // [MC_T][MC_THREADS]: row 7 of every lane's block -- 512 lanes x 64 doubles + everything else is 16 VGPRs more than a lane has
    const int n = T.n, nb = (n + MC_T - 1) / MC_T, P = T.P;
    if (T.same_xcd && (blockIdx.x & 7)) return;
    const int p = T.same_xcd ? blockIdx.x >> 3 : blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    mc_u64* const xb = T.xb;
    mc_u64* const abortw = T.xb + (size_t)2 * MC_PMAX * MC_SLOT;

    // ---- block ownership: local block columns c = 0, 1, ... hold the global block columns C = p + c P; lanes enumerate the blocks
    //      (R = C .. nb-1) of the local columns in order
    const int nbl = p < nb ? (nb - 1 - p) / P + 1 : 0;
    int nblocks = 0;
    for (int c = 0; c < nbl; ++c) nblocks += nb - (p + c * P);
    const bool owner = tid < nblocks;
    int R = 0, C = 0, cl = 0;
    if (owner) {
        int cum = 0, c = 0;
        while (cum + (nb - (p + c * P)) <= tid) { cum += nb - (p + c * P); ++c; }
        cl = c; C = p + c * P; R = C + (tid - cum);
    }
    const int i0_ = MC_T * R, j0_ = MC_T * C;
    const int Cmax_wg = nbl > 0 ? p + (nbl - 1) * P : -1;

    double a[MC_T - 1][MC_T];                               // rows 0..6 of the block; row 7 lives in s_a7[cc][tid]
#pragma unroll
    for (int r = 0; r < MC_T; ++r)
#pragma unroll
        for (int cc = 0; cc < MC_T; ++cc) {
            const int i = i0_ + r, j = j0_ + cc;
            const double t = (owner && i < n && j < n) ? T.A[i + (size_t)T.lda * j] : 0.;
            if (r < MC_T - 1) a[r][cc] = t; else s_a7[cc * MC_THREADS + tid] = t;
        }
    {   // column 0, |x[2:]|^2 and trace(A), redundantly in every workgroup (fixed order: identical everywhere)
        double dg = 0., sg = 0.;
        for (int i = tid; i < MC_MAXN; i += MC_THREADS) {
            const double xi = i < n ? T.A[i] : 0.;
            s_x2[i] = xi; s_x2[MC_MAXN + i] = 0.; s_w[i] = 0.; s_raw[i] = 0.; s_v[i] = 0.;
            if (i >= 2) sg = fma(xi, xi, sg);
            if (i < n) dg += T.A[i + (size_t)T.lda * i];
        }
        dg = wave_sum(dg); sg = wave_sum(sg);
        if (lane == 0) { s_red[32 + wid] = dg; s_red[24 + wid] = sg; }
    }
    __syncthreads();
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
        if (Cmax_wg < kb) break;                                    // this workgroup holds nothing of the trailing matrix any more
        // the lane's table offsets are made opaque once per step: hoisted out of the loop they are ~20 loop-invariant address
        // registers that the allocator spills to scratch and reloads in front of every LDS access
        int i0 = i0_, j0 = j0_, tl = tid;
        asm volatile("" : "+v"(i0), "+v"(j0), "+v"(tl));
        const unsigned tag = T.tag0 + (unsigned)k + 1u;
        const int par = k & 1;
        const int kown = kb % P;                                    // the workgroup holding block column kb (column k+1)
        const double* const sx = s_x2 + par * MC_MAXN;              // column k: sx[k] = d_k, sx[k+1] = alpha, rows k+2.. = the part to eliminate
        double* const sxn = s_x2 + (par ^ 1) * MC_MAXN;
        // ---- A: Householder scalars, by every thread from the same LDS words (no loop over the column: its squared norm was
        //      accumulated by the threads that formed the column during the previous exchange)
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
        // v_i = 0 (i <= k), 1 (i = k+1), x_i * scale (i >= k+2): formed where it is used
#define MC_V(i) ((i) > k + 1 ? sx[(i)] * scale : ((i) == k + 1 ? 1. : 0.))
        if (p == kown) {                                            // one workgroup writes the outputs of the step
            for (int i = tid; i < n; i += MC_THREADS) T.V[i + (size_t)T.ldv * k] = MC_V(i);
            if (tid == 0) { T.D[k] = dk; T.E[k] = beta; T.tau[k] = tau; }
        }
        MCP(0);
        // ---- B: partial sums of y = A v over this lane's block
        const bool active = owner && C >= kb;
        {
            double q = 0., dg = 0.;
            if (active) {
                double vI[MC_T], vJ[MC_T], a7[MC_T];
#pragma unroll
                for (int cc = 0; cc < MC_T; ++cc) a7[cc] = s_a7[cc * MC_THREADS + tl];
#pragma unroll
                for (int r = 0; r < MC_T; ++r) vI[r] = MC_V(i0 + r);
#pragma unroll
                for (int cc = 0; cc < MC_T; ++cc) vJ[cc] = MC_V(j0 + cc);
                double* y1 = Yrow + cl * MC_LDY + i0;
#pragma unroll
                for (int r = 0; r < MC_T; r += 2) {                  // two rows at a time: one 16-byte store, few values live
                    double ta = 0., tb = 0.;
#pragma unroll
                    for (int cc = 0; cc < MC_T; ++cc) { ta = fma(a[r][cc], vJ[cc], ta); tb = fma(r + 1 < MC_T - 1 ? a[r + 1 < MC_T - 1 ? r + 1 : 0][cc] : a7[cc], vJ[cc], tb); }
                    *reinterpret_cast<double2*>(y1 + r) = make_double2(ta, tb);
                    q = fma(vI[r], ta, q); q = fma(vI[r + 1], tb, q);
                }
                if (R != C) {
                    double* y2 = Ycol + R * MC_LDC + cl * MC_T;
#pragma unroll
                    for (int cc = 0; cc < MC_T; cc += 2) {
                        double ta = 0., tb = 0.;
#pragma unroll
                        for (int r = 0; r < MC_T - 1; ++r) { ta = fma(a[r][cc], vI[r], ta); tb = fma(a[r][cc + 1], vI[r], tb); }
                        ta = fma(a7[cc], vI[MC_T - 1], ta); tb = fma(a7[cc + 1], vI[MC_T - 1], tb);
                        *reinterpret_cast<double2*>(y2 + cc) = make_double2(ta, tb);
                    }
                    q *= 2.;
                } else if (need_exact) {
#pragma unroll
                    for (int r = 0; r < MC_T - 1; ++r) dg += (i0 + r >= k + 1) ? a[r][r] : 0.;
                    dg += (i0 + MC_T - 1 >= k + 1) ? a7[MC_T - 1] : 0.;
                }
                if (C == kb) {                                       // column k+1 as it is before this step's update
                    double* dst = s_raw + i0;
#define MC_STAGE(J) { _Pragma("unroll") for (int r = 0; r < MC_T - 1; ++r) dst[r] = a[r][J]; dst[MC_T - 1] = a7[J]; } break
                    switch ((k + 1) & 7) {
                        case 0: MC_STAGE(0); case 1: MC_STAGE(1); case 2: MC_STAGE(2); case 3: MC_STAGE(3);
                        case 4: MC_STAGE(4); case 5: MC_STAGE(5); case 6: MC_STAGE(6); default: MC_STAGE(7);
                    }
#undef MC_STAGE
                }
            }
            q = wave_sum(q);
            if (need_exact) dg = wave_sum(dg);
            if (lane == 0) { s_red[wid] = q; s_red[16 + wid] = dg; }
        }
        MCP(1);
        __syncthreads();
        MCP(2);
#ifdef MC_PROF
        const long long tb1 = clock64();
#endif
        // ---- C1: this workgroup's partial of y, rows k+1 .. n-1, published with the step tag (and the raw column k+1 by its owner).
        //      Rows of this workgroup's own block columns also collect the transposed blocks below their diagonal block (up to 74
        //      table rows): a quad of lanes per such row, each lane a quarter of the block rows, combined by DPP in a fixed order.
        mc_u64* mine = xb + ((size_t)par * P + p) * MC_SLOT;
        const int c_lo = kb > p ? (kb - p + P - 1) / P : 0;
        if (tid == MC_THREADS - 1) {                                 // the scalars first (a thread without a quad duty)
            double qs = 0., ds = 0.;
#pragma unroll
            for (int w = 0; w < MC_THREADS / 64; ++w) { qs += s_red[w]; ds += s_red[16 + w]; }
            mc_put(mine + 2 * (2 * MC_MAXN), qs, tag);
            mc_put(mine + 2 * (2 * MC_MAXN + 1), need_exact ? ds : 0., tag);
        }
        if (tid < 4 * MC_T * MC_NBL) {
            const int cq = tid >> 2, h = tid & 3, c = cq >> 3;
            const int Ci = p + c * P, i = MC_T * Ci + (cq & 7);
            const bool on = c >= c_lo && c < nbl && i > k && i < n;
            double part = 0.;
            if (on) {
                const double* yc = Ycol + c * MC_T + (cq & 7);
                int Rr = Ci + 1 + h;
                for (; Rr + 12 < nb; Rr += 16) {                     // four table rows per trip: their loads are in flight together (same order of addition)
                    const double t0 = yc[Rr * MC_LDC], t1 = yc[(Rr + 4) * MC_LDC], t2 = yc[(Rr + 8) * MC_LDC], t3 = yc[(Rr + 12) * MC_LDC];
                    part += t0; part += t1; part += t2; part += t3;
                }
                for (; Rr < nb; Rr += 4) part += yc[Rr * MC_LDC];
            }
            part += dpp_quad<0xB1>(part);                           // quad_perm [1,0,3,2]
            part += dpp_quad<0x4E>(part);                           // quad_perm [2,3,0,1]: (h0 + h1) + (h2 + h3) in every lane
            if (on && h == 0) {
                double y = 0.;
#pragma unroll
                for (int cc = 0; cc < MC_NBL; ++cc) y += (cc >= c_lo && cc <= c) ? Yrow[cc * MC_LDY + i] : 0.;
                y += part;
                mc_put(mine + 2 * i, y, tag);
                if (p == kown) mc_put(mine + 2 * (MC_MAXN + i), s_raw[i], tag);
            }
        }
        const int nrow = n - 1 - k;                                  // rows k+1 .. n-1, dealt to the threads from row k+1 on
#pragma unroll 1
        for (int t = tid; t < nrow; t += MC_THREADS) {
            const int i = k + 1 + t, Ci = i >> 3;
            if (Ci >= p && (Ci - p) % P == 0) continue;             // one of this workgroup's own columns: published by its quad
            double y = 0.;
            const int c_hi = Ci >= p ? (Ci - p) / P : -1;
#pragma unroll
            for (int cc = 0; cc < MC_NBL; ++cc) y += (cc >= c_lo && cc <= c_hi) ? Yrow[cc * MC_LDY + i] : 0.;
            mc_put(mine + 2 * i, y, tag);
            if (p == kown) mc_put(mine + 2 * (MC_MAXN + i), s_raw[i], tag);
        }
        MCP(3);
        // ---- C2: the sum over workgroups in TWO hops (reduce-scatter, all-gather).  An all-to-all in one hop has every thread of every
        //      workgroup pulling P partials: ~10 000 loads per CU and step, and a hand-off costs what the consumer CU's memory queue
        //      holds (MI355X_MICROARCH.md "handoff-1to1": 1 us idle, 2.3-2.8 us behind 8 streaming waves).  Here the rows k+1.. are
        //      dealt in contiguous slices to the active workgroups; a slice owner gathers the P partials (and the raw column entry) of
        //      its <= 75 rows, sums them in workgroup order and publishes {y_i, x_raw_i}; then every thread fetches the ONE reduced
        //      pair of its row: ~2 400 loads per CU and step.
        int Pact = 0, sme = 0;
        for (int qq = 0; qq < P; ++qq) { const bool actq = (qq < nb ? qq + ((nb - 1 - qq) / P) * P : -1) >= kb; if (qq == p) sme = Pact; Pact += actq ? 1 : 0; }
        const int L = (nrow + Pact - 1) / Pact;                      // rows per slice
        {
            const int j = MC_THREADS - 1 - tid;                      // the slice rows go to the last threads (no quad duty in C1)
            const int t = sme * L + j;
            if (j < L && t < nrow && !aborted) {
                const int i = k + 1 + t;
                mc_u64 g[2 * MC_PMAX + 2];
#pragma unroll
                for (int jj = 0; jj < 2 * MC_PMAX + 2; ++jj) g[jj] = 0;
                int spin = 0;
#ifdef MC_PROF
                if (p == 0 && tid == MC_THREADS - 1) prof[6] += clock64() - tb1;
#endif
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int qq = 0; qq < MC_PMAX; ++qq) {
                        const bool actq = qq < P && (qq < nb ? qq + ((nb - 1 - qq) / P) * P : -1) >= kb;
                        if (actq) {
                            const mc_u64* src = xb + ((size_t)par * P + qq) * MC_SLOT + 2 * i;
                            g[2 * qq] = mc_ld(src); g[2 * qq + 1] = mc_ld(src + 1);
                        }
                    }
                    const mc_u64* src = xb + ((size_t)par * P + kown) * MC_SLOT + 2 * (MC_MAXN + i);
                    g[2 * MC_PMAX] = mc_ld(src); g[2 * MC_PMAX + 1] = mc_ld(src + 1);
#pragma unroll
                    for (int qq = 0; qq < MC_PMAX; ++qq) {
                        const bool actq = qq < P && (qq < nb ? qq + ((nb - 1 - qq) / P) * P : -1) >= kb;
                        if (actq) ok = ok && (unsigned)(g[2 * qq] >> 32) == tag && (unsigned)(g[2 * qq + 1] >> 32) == tag;
                    }
                    ok = ok && (unsigned)(g[2 * MC_PMAX] >> 32) == tag && (unsigned)(g[2 * MC_PMAX + 1] >> 32) == tag;
                    if (ok) break;
                    if (++spin > T.spin_max || ((spin & 255) == 0 && mc_ld(abortw) != 0)) { aborted = true; break; }
                    for (int jn = 0; jn < T.nap_retry; ++jn) __builtin_amdgcn_s_sleep(16);
                }
#ifdef MC_PROF
                if (p == 0 && tid == MC_THREADS - 1) { prof[8] += spin; prof[7] += clock64() - tb1; }
#endif
                double y = 0.;
#pragma unroll
                for (int qq = 0; qq < MC_PMAX; ++qq) {
                    const bool actq = qq < P && (qq < nb ? qq + ((nb - 1 - qq) / P) * P : -1) >= kb;
                    if (actq) y += __longlong_as_double((long long)((g[2 * qq] & 0xffffffffull) | (g[2 * qq + 1] << 32)));
                }
                const double xo = __longlong_as_double((long long)((g[2 * MC_PMAX] & 0xffffffffull) | (g[2 * MC_PMAX + 1] << 32)));
                if (!aborted) { mc_put(mine + MC_RED + 4 * i, y, tag); mc_put(mine + MC_RED + 4 * i + 2, xo, tag); }
            }
        }
        MCP(4);
        double vAv = 0., te = 0., K = 0., wk1 = 0., sgn = 0.;
        int q0 = 0;                                                  // the workgroup of slice 0 (row k+1)
        for (int qq = P - 1; qq >= 0; --qq) if ((qq < nb ? qq + ((nb - 1 - qq) / P) * P : -1) >= kb) q0 = qq;
#pragma unroll 1
        for (int t0 = 0; t0 < nrow || t0 == 0; t0 += MC_THREADS) {
            const int t = t0 + tid, i = k + 1 + t;
            const bool row = i < n;
            const bool first = t0 == 0;
            int qs = 0;                                              // the workgroup that reduced row i: the (t / L)-th active one
            {
                const int sl = t / L;
                int cnt = 0;
                for (int qq = 0; qq < P; ++qq) { const bool actq = (qq < nb ? qq + ((nb - 1 - qq) / P) * P : -1) >= kb; if (actq) { if (cnt == sl) qs = qq; ++cnt; } }
            }
            const bool lact = first && ((lane < P && (lane < nb ? lane + ((nb - 1 - lane) / P) * P : -1) >= kb) || lane == P);
            const mc_u64* ssrc = lane < P ? xb + ((size_t)par * P + lane) * MC_SLOT + 2 * (2 * MC_MAXN)
                                          : xb + ((size_t)par * P + q0) * MC_SLOT + MC_RED + 4 * (k + 1);
            const mc_u64* rsrc = xb + ((size_t)par * P + qs) * MC_SLOT + MC_RED + 4 * i;
            mc_u64 r0 = 0, r1 = 0, r2 = 0, r3 = 0, g0 = 0, g1 = 0, g2 = 0, g3 = 0;
            int spin = 0;
#ifdef MC_PROF
            if (p == 0 && tid == 0 && first) prof[6] += clock64() - tb1;
#endif
            if (first) for (int jn = 0; jn < T.nap_first; ++jn) __builtin_amdgcn_s_sleep(16);   // the slice owners' gathers go first
            for (;;) {
                bool ok = true;
                if (row) { r0 = mc_ld(rsrc); r1 = mc_ld(rsrc + 1); r2 = mc_ld(rsrc + 2); r3 = mc_ld(rsrc + 3); }
                if (lact) { g0 = mc_ld(ssrc); g1 = mc_ld(ssrc + 1); if (lane < P) { g2 = mc_ld(ssrc + 2); g3 = mc_ld(ssrc + 3); } }
                if (row) ok = (unsigned)(r0 >> 32) == tag && (unsigned)(r1 >> 32) == tag && (unsigned)(r2 >> 32) == tag && (unsigned)(r3 >> 32) == tag;
                if (lact) {
                    ok = ok && (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag;
                    if (lane < P) ok = ok && (unsigned)(g2 >> 32) == tag && (unsigned)(g3 >> 32) == tag;
                }
                if (ok) break;
                if (++spin > T.spin_max || ((spin & 255) == 0 && mc_ld(abortw) != 0)) { aborted = true; break; }
                for (int jn = 0; jn < T.nap_retry; ++jn) __builtin_amdgcn_s_sleep(16);
            }
#ifdef MC_PROF
            if (p == 0 && tid == 0 && first) prof[7] += clock64() - tb1;
            if (p == 0 && tid == 0) prof[9] += spin;
            if (p == 0 && tid == 448) prof[9] += spin;
#endif
            if (first) {                                             // the lanes of the wave have reconverged: every lane's granules are in
                const double qv = lact ? __longlong_as_double((long long)((g0 & 0xffffffffull) | (g1 << 32))) : 0.;
                const double dv = (lact && lane < P) ? __longlong_as_double((long long)((g2 & 0xffffffffull) | (g3 << 32))) : 0.;
                for (int qq = 0; qq < P; ++qq) { vAv += mc_bcast(qv, qq); te += mc_bcast(dv, qq); }
                const double yk1 = mc_bcast(qv, P);
                K = -0.5 * tau * tau * vAv;
                wk1 = tau * yk1 + K;                                  // v_{k+1} = 1
            }
            if (row && !aborted) {
                const double y = __longlong_as_double((long long)((r0 & 0xffffffffull) | (r1 << 32)));
                const double xo = __longlong_as_double((long long)((r2 & 0xffffffffull) | (r3 << 32)));
                const double vi = MC_V(i);
                const double wi = fma(K, vi, tau * y);
                const double xn = (xo - vi * wk1) - wi;              // column k+1 after this step's update
                s_w[i] = wi;
                s_v[i] = vi;
                sxn[i] = xn;
                if (i >= k + 3) sgn = fma(xn, xn, sgn);
            }
        }
        sgn = wave_sum(sgn);
        if (lane == 0) s_red[24 + wid] = sgn;
        MCP(5);
        if (tid <= k && tid < MC_MAXN) { s_w[tid] = 0.; s_v[tid] = 0.; }   // rows above the trailing matrix
        if (tid + MC_THREADS <= k && tid + MC_THREADS < MC_MAXN) { s_w[tid + MC_THREADS] = 0.; s_v[tid + MC_THREADS] = 0.; }
        if (aborted) __hip_atomic_store(abortw, (mc_u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__syncthreads_or(aborted ? 1 : 0)) { aborted = true; break; }
        MCP(6);
        if (need_exact) {
            trem = te;
            if (te <= t_exit) { kexit = k; break; }                  // uniform over all workgroups: reflector k exists, rows k+1.. are dropped
        }
        // ---- D: A <- A - v w^T - w v^T on the registers
        if (active) {
            double vJ[MC_T], wJ[MC_T];
#pragma unroll
            for (int cc = 0; cc < MC_T; ++cc) { vJ[cc] = s_v[j0 + cc]; wJ[cc] = s_w[j0 + cc]; }
#pragma unroll
            for (int r = 0; r < MC_T - 1; ++r) {
                const double vIr = s_v[i0 + r], wIr = s_w[i0 + r];
#pragma unroll
                for (int cc = 0; cc < MC_T; ++cc) a[r][cc] = fma(-vIr, wJ[cc], fma(-wIr, vJ[cc], a[r][cc]));
            }
            {
                const double vIr = s_v[i0 + MC_T - 1], wIr = s_w[i0 + MC_T - 1];
#pragma unroll
                for (int cc = 0; cc < MC_T; ++cc) { double* q7 = s_a7 + cc * MC_THREADS + tl; *q7 = fma(-vIr, wJ[cc], fma(-wIr, vJ[cc], *q7)); }
            }
        }
        MCP(7);
        // no barrier here: the next step reads the other parity of s_x2; s_w, s_raw, s_red and the tables are rewritten only after its
        // first barrier (s_red[24..31], written above, is read at its top: behind this step's second barrier)
    }
#undef MC_V
#ifdef MC_PROF
    if (p == 0 && lane == 0 && wid == pw && T.dbg) for (int i = 0; i < 8; ++i) T.dbg[i] = prof[i];
    if (p == 0 && tid == MC_THREADS - 1 && T.dbg) { T.dbg[8] = prof[8]; T.dbg[11] = prof[6]; T.dbg[12] = prof[7]; }
    if (p == 0 && tid == 0 && T.dbg) { T.dbg[9] = prof[9]; if (pw != 0) { T.dbg[13] = prof[6]; T.dbg[14] = prof[7]; } }
    if (p == 0 && tid == 448 && T.dbg) T.dbg[10] = prof[9];
#endif
    if (aborted) { if (tid == 0) __hip_atomic_store(abortw + 1, (mc_u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (kexit >= 0) {
        const int kb = (kexit + 1) / MC_T;
        if (p == kb % P) {
            for (int i = kexit + 1 + tid; i < n; i += MC_THREADS) { T.D[i] = 0.; if (i < n - 1) { T.E[i] = 0.; T.tau[i] = 0.; } }
            if (tid == 0 && T.nref) T.nref[0] = (double)(kexit + 1);
        }
    } else if (p == (nb - 1) % P) {                                   // the workgroup that ran the last step holds column n-1
        if (tid == 0) { T.D[n - 1] = s_x2[((n - 1) & 1) * MC_MAXN + n - 1]; if (T.nref) T.nref[0] = (double)(n - 1); }
    }
}


// ==========================================================================================================================
// k_sytrd_ro -- the same chain with the FULL symmetric matrix resident and block ROWS dealt cyclically to P <= 16 workgroups.
// With whole rows at home, y_i = (A v)_i is a local sum: the reduce-scatter hop of k_sytrd_mc disappears and a Householder
// step costs ONE exchange (every workgroup publishes the y of its <= 64 rows, every row thread polls the one y it needs).
// The price is twice the storage (both triangles: 13-14 workgroups at n = 600) and a replicated copy of the current
// PANEL of 8 columns in every workgroup's LDS: the column a step eliminates must be known everywhere, and instead of
// broadcasting one column per step the owners publish a raw block column every 8th step and every workgroup applies the
// rank-2 updates of the steps in between to its copy (16 FMAs per lane and step).
#define RO_PMAX 16
#define RO_LD 66                                   // stride of the partial table between block columns (64 local rows + 2)
#define RO_SMEM_DOUBLES (MC_MAXNB * RO_LD + 8 * MC_MAXN + 2 * MC_MAXN + 8 * 64 + 64)
#define RO_XB_U64 (2 * MC_MAXN * 2 + 2 * 8 * RO_PMAX * 16 + 2 * 8 * 16 + MC_MAXN * 16)

__global__ __launch_bounds__(MC_THREADS) void k_sytrd_ro(McArgs T) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ytab = smem;                                  // [MC_MAXNB][RO_LD]: partial of local row lr from block column C
    double* panel = Ytab + MC_MAXNB * RO_LD;                    // [8][MC_MAXN]: columns 8 c .. 8 c + 7 of the current matrix, replicated
    double* s_v = panel + 8 * MC_MAXN;
    double* s_w = s_v + MC_MAXN;
    double* s_raw = s_w + MC_MAXN;                        // [8][64]: this workgroup's rows of the next raw block column
    double* s_red = s_raw + 8 * 64;                       // [64]: 0..7 v^T A v partials, 16..23 trailing-trace partials, 24..31 |x[k+2:]|^2 partials, 32..39 trace(A) partials
    const int n = T.n, nb = (n + MC_T - 1) / MC_T, P = T.P, p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    mc_u64* const ybuf = T.xb;                            // [2][MC_MAXN][2]  y_i, by the owner of row i
    // what EVERY wave of every workgroup reads (the workgroups' scalars, y of row k+1) is published in 8 copies, one per wave index, each
    // workgroup's pair in a 128-byte line of its own: a line polled by 104 waves at once is served one reader after the other
    mc_u64* const sbuf = ybuf + 2 * MC_MAXN * 2;          // [2][8][RO_PMAX][16]  {v^T y partial, trailing-trace partial} per workgroup
    mc_u64* const y1buf = sbuf + 2 * 8 * RO_PMAX * 16;    // [2][8][16]           y of row k+1
    mc_u64* const pbuf = y1buf + 2 * 8 * 16;              // [MC_MAXN][8][2]      raw block column, by rows
    mc_u64* const abortw = T.xb + (size_t)2 * MC_PMAX * MC_SLOT;

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
    const bool XP_PUB = T.nap_first != 98, XP_LD = T.nap_first != 97;
    const bool XP_NOVAL = T.nap_first >= 97 && T.nap_first <= 99;
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
        asm volatile("" : "+v"(i0), "+v"(j0), "+v"(rl));          // (see k_sytrd_mc: keeps ~20 hoisted address registers out of scratch)
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
#define RO_V(i) ((i) > k + 1 ? sx[(i)] * scale : ((i) == k + 1 ? 1. : 0.))
        if (p == kb % P) {                                          // one (active) workgroup writes the outputs of the step
            for (int i = tid; i < n; i += MC_THREADS) T.V[i + (size_t)T.ldv * k] = RO_V(i);
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
                double* y1 = Ytab + (j0 >> 3) * RO_LD + rl * MC_T;
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
            mc_u64* dst = sbuf + (((size_t)par * 8 + (tid - (MC_THREADS - 8))) * RO_PMAX + p) * 16;
            if (XP_PUB) mc_put_t<false>(dst, qs, tag);
            if (XP_PUB) mc_put_t<false>(dst + 2, need_exact ? ds : 0., tag);
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
                    const double t0_ = yr[Cc * RO_LD], t1_ = yr[(Cc + 8) * RO_LD], t2_ = yr[(Cc + 16) * RO_LD], t3_ = yr[(Cc + 24) * RO_LD];
                    part += t0_; part += t1_; part += t2_; part += t3_;
                }
                for (; Cc < nb; Cc += 8) part += yr[Cc * RO_LD];
            }
            part += dpp_quad<0xB1>(part);
            part += dpp_quad<0x4E>(part);
            part += __shfl_xor(part, 4);
            if (on && h == 0) if (XP_PUB) mc_put_t<false>(ybuf + ((size_t)par * MC_MAXN + i) * 2, part, tag);
            if (on && i == k + 1) if (XP_PUB) mc_put_t<false>(y1buf + ((size_t)par * 8 + h) * 16, part, tag);      // all 8 lanes of the row hold the sum
        }
        if (newpanel) {                                              // 8 columns x 64 local rows: one value per thread
            const int j = tid >> 6, lr = tid & 63;
            const int Rr = p + (lr >> 3) * P, i = MC_T * Rr + (lr & 7);
            if ((lr >> 3) < nbl && Rr >= kb && i > k && i < n) if (XP_PUB) mc_put_t<false>(pbuf + ((size_t)i * 8 + j) * 2, s_raw[j * 64 + lr], tag);
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
            const bool ly1 = first && lane == RO_PMAX;
            const mc_u64* ssrc = lane < RO_PMAX ? sbuf + (((size_t)par * 8 + wid) * RO_PMAX + lane) * 16 : y1buf + ((size_t)par * 8 + wid) * 16;
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
#define RO_G2D(g) __hiloint2double((int)(g).z, (int)(g).x)
            const double vi = row ? RO_V(i) : 0.;                    // (before the raw block column overwrites the panel that holds column k)
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
                        if (half == 0) xo = RO_G2D(p0); else panel[4 * MC_MAXN + i] = RO_G2D(p0);
                        panel[(4 * half + 1) * MC_MAXN + i] = RO_G2D(p1);
                        panel[(4 * half + 2) * MC_MAXN + i] = RO_G2D(p2);
                        panel[(4 * half + 3) * MC_MAXN + i] = RO_G2D(p3);
                    }
                }
            } else if (row) xo = sxn[i];
#ifdef MC_PROF
            if (p == 0 && wid == pw && first) { if (lane == 0) prof[7] += spin; if (lane == 1) prof[8] += spin; if (lane == 20) prof[9] += spin; }
#endif
            if (first) {                                             // the lanes of the wave have reconverged: every lane's granules are in
                const double qv = (lsc || ly1) ? RO_G2D(ga) : 0.;
                double qs = lsc ? qv : 0., ds = lsc ? RO_G2D(gb) : 0.;   // lanes 0..15: a fixed tree over the row of 16 lanes, the same in every wave of every workgroup
                qs += dpp_quad<0xB1>(qs); ds += dpp_quad<0xB1>(ds);
                qs += dpp_quad<0x4E>(qs); ds += dpp_quad<0x4E>(ds);
                qs += dpp_quad<0x141>(qs); ds += dpp_quad<0x141>(ds);
                qs += dpp_quad<0x140>(qs); ds += dpp_quad<0x140>(ds);
                vAv = mc_bcast(qs, 0); te = mc_bcast(ds, 0);
                const double yk1 = mc_bcast(qv, RO_PMAX);
                K = -0.5 * tau * tau * vAv;
                wk1 = tau * yk1 + K;                                  // v_{k+1} = 1
            }
            if (row && !aborted) {
                const double y = RO_G2D(gr);
                const double wi = fma(K, vi, tau * y);
                const double xn = (xo - vi * wk1) - wi;              // column k+1 after this step's update
                s_w[i] = wi;
                s_v[i] = vi;
                sxn[i] = xn;
                if (i >= k + 3) sgn = fma(xn, xn, sgn);
            }
#undef RO_G2D
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
#undef RO_V
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

static int ro_workgroups(int n) {
    const int nb = (n + MC_T - 1) / MC_T;
    for (int P = 1; P <= RO_PMAX; ++P) {
        const int nbl = (nb - 1) / P + 1;
        if (nbl <= 8 && nbl * nb <= MC_THREADS) return P;
    }
    return 0;
}
static int g_mc_variant = 1;                                          // 0: k_sytrd_mc (block columns of the lower triangle), 1: k_sytrd_ro
void eigh_mc_set_variant(int v) { g_mc_variant = v; }

// number of workgroups for an n x n matrix: the fewest whose busiest member has <= MC_THREADS blocks and <= MC_NBL block columns
static int mc_workgroups(int n) {
    const int nb = (n + MC_T - 1) / MC_T;
    for (int P = 1; P <= MC_PMAX; ++P) {
        const int nbl = (nb - 1) / P + 1;
        if (nbl > MC_NBL) continue;
        int blocks = 0;
        for (int c = 0; c < nbl; ++c) blocks += nb - c * P;          // workgroup 0 is the busiest
        if (blocks <= MC_THREADS) return P;
    }
    return 0;
}

size_t eigh_mc_xbuf_bytes() { return sizeof(mc_u64) * ((size_t)2 * MC_PMAX * MC_SLOT + 8); }
int eigh_mc_max_n() { return MC_MAXN; }

// A (n x n symmetric, device) -> D, E, tau (tau[n-1] = number of reflectors), V on `st`.  xbuf: eigh_mc_xbuf_bytes() of device memory,
// zeroed once at allocation; *epoch is advanced per launch.  The kernel reports through xbuf's status word (eigh_mc_status).
int eigh_mc_tridiagonalize(tnml_ctx* c, hipStream_t st, const double* A, int n, double* D, double* E, double* tau, double* V, double psd_tol,
                           void* xbuf, unsigned* epoch, long long* dbg, int nap_first, int nap_retry, int same_xcd, int spin_max) {
    if (n > MC_MAXN || n < 3) return tnml_fail(c, "eigh_mc_tridiagonalize: n=%d outside 3..%d", n, MC_MAXN);
    const int P = mc_workgroups(n);
    if (P == 0) return tnml_fail(c, "eigh_mc_tridiagonalize: no workgroup count fits n=%d", n);
    *epoch = (*epoch % 4000000u) + 1u;
    if (spin_max < 0) spin_max = MC_SPIN_MAX;                  // (0: the first failed poll aborts -- the fallback test)
    if (g_mc_variant == 1) {
        const int Pr = ro_workgroups(n);
        if (Pr == 0) return tnml_fail(c, "eigh_mc_tridiagonalize: no workgroup count fits n=%d", n);
        McArgs t{A, n, n, D, E, tau, V, n, tau + (n - 1), psd_tol, (mc_u64*)xbuf, *epoch * 1024u, Pr, 0, nap_first, nap_retry, spin_max, dbg};
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_ro), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RO_SMEM_DOUBLES * sizeof(double))) != hipSuccess)
            return tnml_fail(c, "eigh_mc_tridiagonalize: cannot reserve %zu bytes of LDS", RO_SMEM_DOUBLES * sizeof(double));
        hipLaunchKernelGGL(k_sytrd_ro, dim3(Pr), dim3(MC_THREADS), RO_SMEM_DOUBLES * sizeof(double), st, t);
        HIPCK(c, hipGetLastError());
        return 0;
    }
    McArgs t{A, n, n, D, E, tau, V, n, tau + (n - 1), psd_tol, (mc_u64*)xbuf, *epoch * 1024u, P, same_xcd, nap_first, nap_retry, spin_max, dbg};
    // (per launch: the attribute is per device, a process may drive several, and this kernel runs for milliseconds)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_mc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MC_SMEM_DOUBLES * sizeof(double))) != hipSuccess)
        return tnml_fail(c, "eigh_mc_tridiagonalize: cannot reserve %zu bytes of LDS", MC_SMEM_DOUBLES * sizeof(double));
    hipLaunchKernelGGL(k_sytrd_mc, dim3(same_xcd ? 8 * P : P), dim3(MC_THREADS), MC_SMEM_DOUBLES * sizeof(double), st, t);
    HIPCK(c, hipGetLastError());
    return 0;
}
// device address of the status word (non-zero after a launch that gave up waiting for a peer workgroup)
const void* eigh_mc_status_ptr(const void* xbuf) { return (const mc_u64*)xbuf + (size_t)2 * MC_PMAX * MC_SLOT + 1; }
