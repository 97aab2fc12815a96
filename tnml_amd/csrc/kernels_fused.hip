// kernels_fused.hip -- the forward pass B*t.v (fixedL.cc:318,377,399,416) as ONE persistent kernel for the bonds whose Label
// index sits on an environment (m = 120, fp64 storage): the feature GEMM and the label dot of kernels_gemm.hip /
// kernels_stream.hip, run by different waves of the same workgroup so that the HBM stream of the Label-carrying environment
// (577 MB per pass at BASELINE config 3) hides behind the matrix pipe.
//
// Why not two queues: the label dot pulls its 6.3 TB/s with ~7 resident waves per CU that each keep ~10 KB of loads in flight;
// launched beside the feature GEMM it is starved, and over half the images it takes as long as over all of them
// (profiles/r02_ab_two_queue_forward.txt).  Here the streaming waves are residents of the GEMM's own workgroup:
//
//   workgroup = 16 waves, one workgroup per CU, looping over 64-image tiles (tile t, t + gridDim.x, ...):
//     waves 0..11  (4 image-row groups x 3 column groups, 5 column tiles each): T = X*M for tile t on v_mfma_f64_16x16x4_f64,
//                  X_n = EI_n (x) phiI_n built while staging; the epilogue folds the output-site feature and leaves
//                  U[q][image] (120 x 64 doubles) in LDS -- it never goes to HBM;
//     waves 12..15 (one lane per image): P[l][n] = sum_q EL[l][q][n] U[q][n] for the PREVIOUS tile of this workgroup, paced by the
//                  GEMM's own barriers: wave s takes q = s, s+4, ..., one of them per barrier interval (two per 16-deep reduction
//                  chunk), the loads of the next one in flight across the barrier; after the last chunk the four partial sums meet in LDS and 64
//                  lanes finish the tile exactly like k_labeldot (dP, per-label cost partials, argmax count, |P|^2).
//   Every wave executes the same barrier sequence (2 per chunk + 2 per tile); one extra "drain" round (two barriers, no pacing) lets
//   the streaming waves finish the last tile.  A double-buffered 8-deep variant with the chunk stream continuous across tiles was
//   measured slower (3.75-3.79 vs 3.66-3.68 ms per bond update, profiles/r02_ab_fused_variants.txt) and is not kept.  Deterministic: fixed summation order per image, per-tile partials reduced by k_reduce_partials.
#include "tnml_internal.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define FF_BM 64
#define FF_BN 240
#define FF_KT 16
#define FF_XS (FF_BM + 16)
#define FF_MS FF_BN
#define FF_MO 120
#define FF_CT 5
#define FF_NMMA 768          // threads of the 12 GEMM waves

static __device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
static __device__ __forceinline__ double ldnt(const double* p) { return __builtin_nontemporal_load(p); }

// per-image epilogue shared with k_labeldot's tail: cost buckets of one 64-image tile -> partials[tile][12]
static __device__ __forceinline__ void tile_partials(double val, int lab, int cor, bool pap, double* s_part, double* out, int lane) {
    if (pap) {
        const double s = wave_sum(val);
        if (lane < 12) out[lane] = lane == 11 ? s : 0.;
        return;
    }
    double mine = 0.;
#pragma unroll
    for (int t = 0; t < TNML_NL; ++t) {
        const double s = wave_sum(lab == t ? val : 0.);
        if (lane == t) mine = s;
    }
    const double sc = wave_sum((double)cor);
    if (lane == 10) mine = sc;
    if (lane < 12) out[lane] = mine;
    (void)s_part;
}

#define FF_PT(t) (t)
#define FF_LDS_DOUBLES (FF_KT * FF_XS + FF_KT * FF_MS + FF_MO * FF_BM + 4 * TNML_NL * FF_BM)

// the 12 GEMM waves of a workgroup: all tiles of this workgroup, then one drain round (barriers only)
static __device__ __forceinline__ void ff_gemm_role(const FwdFusedArgs& A, double* Xs, double* Ms, double* Ub, int tid, int lane, int wid) {
    const int wr = wid / 3, wc = wid % 3;                   // image group, column group
    const int NTp = A.NTp, G = gridDim.x;
    constexpr int NXI = (FF_KT / 2) * (FF_BM / 4);          // 128 4-image pieces of the environment per chunk
    constexpr int NMI = FF_KT * (FF_BN / 2);                // 1920 double2 pieces of M per chunk
    constexpr int NM = (NMI + FF_NMMA - 1) / FF_NMMA;       // 3
    constexpr int NCH = 240 / FF_KT;                        // 15 chunks
    const int xc4 = tid % (FF_BM / 4), xar = tid / (FF_BM / 4);
    const int g = lane >> 4;
    double xr[4] = {0., 0., 0., 0.}; double2 mr[NM];
    double p0[4] = {0., 0., 0., 0.}, p1[4] = {0., 0., 0., 0.};
    auto load_phi = [&](int n0) {
        if (tid < NXI) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { p0[e] = A.phiI[n0 + xc4 * 4 + e]; p1[e] = A.phiI[NTp + n0 + xc4 * 4 + e]; }
        }
    };
    auto load_chunk = [&](int n0, int k0) {
        if (tid < NXI) {
            const int a = k0 / 2 + xar;
            const double2 ea = ld2(A.EI + (size_t)a * NTp + n0 + xc4 * 4), eb = ld2(A.EI + (size_t)a * NTp + n0 + xc4 * 4 + 2);
            xr[0] = ea.x; xr[1] = ea.y; xr[2] = eb.x; xr[3] = eb.y;
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int idx = tid + q * FF_NMMA;
            const int r = idx / (FF_BN / 2), c2 = idx % (FF_BN / 2);
            mr[q] = make_double2(0., 0.);
            if (idx < NMI) mr[q] = ld2(A.M + (size_t)(k0 + r) * A.Np + c2 * 2);
        }
    };
    auto store_chunk = [&]() {
        if (tid < NXI) {
            double* x0 = &Xs[(2 * xar) * FF_XS + xc4 * 4];
            double* x1 = &Xs[(2 * xar + 1) * FF_XS + xc4 * 4];
            *reinterpret_cast<double2*>(x0) = make_double2(xr[0] * p0[0], xr[1] * p0[1]);
            *reinterpret_cast<double2*>(x0 + 2) = make_double2(xr[2] * p0[2], xr[3] * p0[3]);
            *reinterpret_cast<double2*>(x1) = make_double2(xr[0] * p1[0], xr[1] * p1[1]);
            *reinterpret_cast<double2*>(x1 + 2) = make_double2(xr[2] * p1[2], xr[3] * p1[3]);
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int idx = tid + q * FF_NMMA;
            if (idx < NMI) { const int r = idx / (FF_BN / 2), c2 = idx % (FF_BN / 2); *reinterpret_cast<double2*>(&Ms[r * FF_MS + c2 * 2]) = mr[q]; }
        }
    };
    int tile = blockIdx.x;
    if (tile < A.ntiles) { load_phi(FF_PT(tile) * FF_BM); load_chunk(FF_PT(tile) * FF_BM, 0); }
    for (; tile < A.ntiles + G; tile += G) {
        const bool has = tile < A.ntiles;
        if (!has) { __syncthreads(); __syncthreads(); continue; }     // drain round: the streaming waves run free between two barriers
        const int n0 = FF_PT(tile) * FF_BM;
        const bool has_next = tile + G < A.ntiles;
        // the output-site feature of this lane's image, needed by the epilogue: fetched now, off the tile's critical path
        const double ph = A.phiO[(size_t)(g & 1) * NTp + n0 + wr * 16 + (lane & 15)];
        f64x4 acc[FF_CT];
#pragma unroll
        for (int c = 0; c < FF_CT; ++c) acc[c] = f64x4{0., 0., 0., 0.};
        store_chunk();                                       // chunk 0: its loads were issued before the previous tile's epilogue
        __syncthreads();
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch + 1 < NCH) load_chunk(n0, (ch + 1) * FF_KT);
#pragma unroll
            for (int ks = 0; ks < FF_KT / 4; ++ks) {
                const int krow = 4 * ks + (lane >> 4);
                const double xf = Xs[krow * FF_XS + wr * 16 + (lane & 15)];
#pragma unroll
                for (int c = 0; c < FF_CT; ++c) {
                    const double mf = Ms[krow * FF_MS + (wc * FF_CT + c) * 16 + (lane & 15)];
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(mf, xf, acc[c], 0, 0, 0);
                }
            }
            __syncthreads();                                 // every GEMM wave is done reading this chunk
            if (ch + 1 < NCH) store_chunk();
            __syncthreads();
        }
        // the first chunk of this workgroup's next tile: its loads fly during the epilogue and the tile barrier
        if (has_next) { load_phi(FF_PT(tile + G) * FF_BM); load_chunk(FF_PT(tile + G) * FF_BM, 0); }
        // lane (g = lane>>4, i = lane&15) holds rows g+4e of column tile c for image i; rows 2q, 2q+1 (site index t = 0,1 of
        // output link q) sit on lane groups g and g^1
#pragma unroll
        for (int c = 0; c < FF_CT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = (wc * FF_CT + c) * 16 + g + 4 * e;
                double v = acc[c][e] * ph;
                v += __shfl_xor(v, 16);
                if ((g & 1) == 0) Ub[(j >> 1) * FF_BM + wr * 16 + (lane & 15)] = v;
            }
        __syncthreads();
    }
}

// the 4 streaming waves: idle in the first round, then the label dot of the tile the GEMM waves finished one round earlier
static __device__ __forceinline__ void ff_stream_role(const FwdFusedArgs& A, const double* Ub, double* red, int lane, int sw) {
    const int NTp = A.NTp, G = gridDim.x;
    constexpr int NCH = 240 / FF_KT;
    for (int tile = blockIdx.x; tile < A.ntiles + G; tile += G) {
        const int stile = tile - G;
        const bool has = stile >= 0;
        const int ptile = has ? FF_PT(stile) : 0;
        const int ns = ptile * FF_BM + lane;
        double px[TNML_NL];
#pragma unroll
        for (int l = 0; l < TNML_NL; ++l) px[l] = 0.;
        // one row of the contraction index per barrier interval (two intervals per chunk), the next row's loads in flight:
        // 10 + 10 doubles of staging keep the role inside the 128 VGPRs that 16 waves per CU allow
        double ea[TNML_NL], eb[TNML_NL];
        auto stream_load = [&](int k, double (&e)[TNML_NL]) {
            const double* ep = A.EL + (size_t)(sw + 4 * k) * NTp + ns;
#pragma unroll
            for (int l = 0; l < TNML_NL; ++l) e[l] = ldnt(ep + (size_t)l * A.EL_lstride);
        };
        auto consume = [&](int k, const double (&e)[TNML_NL]) {
            const double u = Ub[(sw + 4 * k) * FF_BM + lane];
#pragma unroll
            for (int l = 0; l < TNML_NL; ++l) px[l] = fma(e[l], u, px[l]);
        };
        if (tile >= A.ntiles) {
            // drain round (the GEMM waves only pass the two barriers): no pacing, three rows in flight -- with two the round is
            // latency bound and a forward pass 30 us longer (this loop costs 25 spilled VGPRs, used here only)
            __syncthreads();
            if (has) {
                double ec[TNML_NL];
                stream_load(0, ea); stream_load(1, eb);
                for (int k = 0; k < 2 * NCH; k += 3) {               // 30 rows
                    stream_load(k + 2 < 2 * NCH ? k + 2 : k, ec); consume(k, ea);
                    stream_load(k + 3 < 2 * NCH ? k + 3 : k, ea); consume(k + 1, eb);
                    stream_load(k + 4 < 2 * NCH ? k + 4 : k, eb); consume(k + 2, ec);
                }
            }
        } else {
            if (has) stream_load(0, ea);
            __syncthreads();
            for (int ch = 0; ch < NCH; ++ch) {
                if (has) { stream_load(2 * ch + 1, eb); consume(2 * ch, ea); }
                __syncthreads();
                if (has) { if (ch + 1 < NCH) stream_load(2 * ch + 2, ea); consume(2 * ch + 1, eb); }
                __syncthreads();
            }
        }
        if (has) {
#pragma unroll
            for (int l = 0; l < TNML_NL; ++l) red[(sw * TNML_NL + l) * FF_BM + lane] = px[l];
        }
        __syncthreads();
        if (sw == 0 && has) {                                // 64 lanes = the 64 images of the finished tile
            double P[TNML_NL];
#pragma unroll
            for (int l = 0; l < TNML_NL; ++l) {
                double s = 0.;
#pragma unroll
                for (int w = 0; w < 4; ++w) s += red[(w * TNML_NL + l) * FF_BM + lane];     // fixed order
                P[l] = s;
            }
            const int lab = A.label[ns];
            double val = 0.; int cor = 0;
            if (A.mode == LD_MODE_PAP) {
#pragma unroll
                for (int l = 0; l < TNML_NL; ++l) {
                    val = fma(P[l], P[l], val);                                            // sqr(norm(pv)), :400
                    if (A.P) A.P[(size_t)l * NTp + ns] = P[l];
                }
                if (lab < 0) val = 0.;
            } else {
                double best = fabs(P[0]); int arg = 0;
#pragma unroll
                for (int l = 0; l < TNML_NL; ++l) {
                    const double tgt = l == lab ? 1. : 0.;
                    const double d = lab >= 0 ? tgt - P[l] : 0.;                           // deltas[t.l] - P
                    val = fma(d, d, val);
                    if (A.dP) A.dP[(size_t)l * NTp + ns] = d;
                    if (A.P) A.P[(size_t)l * NTp + ns] = P[l];
                    const double wgt = fabs(P[l]);
                    if (wgt > best) { best = wgt; arg = l; }                               // first maximum
                }
                cor = (lab >= 0 && arg == lab) ? 1 : 0;
            }
            tile_partials(val, lab, cor, A.mode == LD_MODE_PAP, nullptr, A.partials + (size_t)ptile * 12, lane);
        }
        // (the first barrier of the next round orders these reads of `red` and of Ub against the next round's writes)
    }
}

__global__ __launch_bounds__(1024) void k_fwd_fused(FwdFusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double ff_lds[];
    double* Xs = ff_lds;
    double* Ms = Xs + FF_KT * FF_XS;
    double* Ub = Ms + FF_KT * FF_MS;                        // U[q][image] of the tile the streaming waves work on
    double* red = Ub + FF_MO * FF_BM;                       // [stream wave][label][image]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (wid < 12) ff_gemm_role(A, Xs, Ms, Ub, tid, lane, wid);
    else          ff_stream_role(A, Ub, red, lane, wid - 12);
}

int launch_fwd_fused(tnml_ctx* c, const FwdFusedArgs& a) {
    if (a.NTp % FF_BM || a.Np != FF_BN || a.Kp != 240 || a.mO != FF_MO || a.mI != 120) return tnml_fail(c, "fwd_fused: shape not supported");
    if (a.ntiles > c->partial_cap) return tnml_fail(c, "fwd_fused: partial buffer too small");
    if (!c->cu_count) { hipDeviceProp_t pr; c->cu_count = hipGetDeviceProperties(&pr, c->cfg.device) == hipSuccess ? pr.multiProcessorCount : 256; }
    int ncu = c->cu_count;
    if (c->fused_fwd > 2 && c->fused_fwd < ncu) ncu = c->fused_fwd;      // test knob: fewer workgroups -> several rounds each
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    const size_t lds = sizeof(double) * FF_LDS_DOUBLES;
    if (!c->attr_fused) {                                                // per device: remembered in the context
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return tnml_fail(c, "fwd_fused: cannot reserve %zu bytes of LDS", lds);
        c->attr_fused = true;
    }
    {
        ProfScope ps(c, KC_FWD_FUSED);
        hipLaunchKernelGGL(k_fwd_fused, dim3(grid), dim3(1024), lds, c->stream, a);
    }
    HIPCK(c, hipGetLastError());
    return 0;
}
