// kernels_gemm.hip -- the MFMA kernels of the bond contraction (gfx950).
//
// k_fgemm64 ("feature GEMM", forward): replaces B*t.v (fixedL.cc:318,377,399,416) and, in its shift form, the per-image
//          (t.A(c)*W.A(c))*E products of init/shiftE (fixedL.cc:144-148,223-227).
//   out[l][q][n] = sum_t phiO[t][n] * sum_{a,s} phiI[s][n] * EI[l?][a][n] * M[l?][2a+s][2q+t]
//   i.e. T = X*M with X_n = EI_n (x) phiI_n built on the fly in LDS (the dense t.v of the reference is never formed),
//   followed by the contraction with the output-site feature.  GEMM shape: (NTp images) x (Np = 2*mO) x (Kp = 2*mI).
// k_bgemm64 ("gradient GEMM", backward): replaces tensors[nt] += dP*dag(t.v) (fixedL.cc:379,418)
//   G[l][2a+s][2q+t] = sum_n (phiI[s][n] EI[a][n]) * (w[l][n] phiO[t][n] Zq[q][n]),  Zq = sum_l EL[l] dP[l] built in the
//   operand staging; GEMM shape Kp x Np with the reduction over images (split-K over image ranges into fp64 slabs that
//   k_slab_reduce64 sums in a fixed order -> deterministic).
//
// Arithmetic: v_mfma_f64_16x16x4_f64 (fp64 operands and accumulation, 78.6 TF peak) is the default and the parity path
// (TNML_F64: fp64 storage; TNML_F64_E32: environments and features stored in fp32, widened while staging).  The f64 kernels
// put the IMAGES on the MFMA column index: the f64 C fragment is col = lane&15, row = (lane>>4) + 4*reg, so 16 consecutive
// lanes hold 16 consecutive images of one output row -> 128-byte coalesced stores into the image-fastest output.
// k_fgemm / k_bgemm are the v_mfma_f32_16x16x4_f32 counterparts over fp32 storage (TNML_F32, the tolerance-study mode).
#include <cstdlib>

#include "tnml_internal.h"

// ==========================================================================================
// fp64 MFMA flavour
// ==========================================================================================
typedef double f64x4 __attribute__((ext_vector_type(4)));

// 4 / 2 consecutive environment (or feature) elements, stored as fp32 (TNML_F64) or fp64 (TNML_F64_STRICT)
template <typename T> struct V4;
template <> struct V4<float> {
    float4 v;
    static __device__ __forceinline__ V4 load(const float* p) { V4 r; r.v = *reinterpret_cast<const float4*>(p); return r; }
    static __device__ __forceinline__ V4 zero() { V4 r; r.v = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
    __device__ __forceinline__ double x() const { return v.x; }
    __device__ __forceinline__ double y() const { return v.y; }
    __device__ __forceinline__ double z() const { return v.z; }
    __device__ __forceinline__ double w() const { return v.w; }
};
template <> struct V4<double> {
    double2 a, b;
    static __device__ __forceinline__ V4 load(const double* p) { V4 r; r.a = *reinterpret_cast<const double2*>(p); r.b = *reinterpret_cast<const double2*>(p + 2); return r; }
    static __device__ __forceinline__ V4 zero() { V4 r; r.a = make_double2(0., 0.); r.b = make_double2(0., 0.); return r; }
    __device__ __forceinline__ double x() const { return a.x; }
    __device__ __forceinline__ double y() const { return a.y; }
    __device__ __forceinline__ double z() const { return b.x; }
    __device__ __forceinline__ double w() const { return b.y; }
};
template <typename T> struct V2;
template <> struct V2<float> {
    float2 v;
    static __device__ __forceinline__ V2 load(const float* p) { V2 r; r.v = *reinterpret_cast<const float2*>(p); return r; }
    static __device__ __forceinline__ V2 load_nt(const float* p) {            // streamed once: non-temporal
        typedef float f2v __attribute__((ext_vector_type(2)));
        const f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
        V2 r; r.v = make_float2(t.x, t.y); return r;
    }
    static __device__ __forceinline__ V2 zero() { V2 r; r.v = make_float2(0.f, 0.f); return r; }
    __device__ __forceinline__ double x() const { return v.x; }
    __device__ __forceinline__ double y() const { return v.y; }
};
template <> struct V2<double> {
    double2 v;
    static __device__ __forceinline__ V2 load(const double* p) { V2 r; r.v = *reinterpret_cast<const double2*>(p); return r; }
    static __device__ __forceinline__ V2 load_nt(const double* p) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
        V2 r; r.v = make_double2(t.x, t.y); return r;
    }
    static __device__ __forceinline__ V2 zero() { V2 r; r.v = make_double2(0., 0.); return r; }
    __device__ __forceinline__ double x() const { return v.x; }
    __device__ __forceinline__ double y() const { return v.y; }
};

// RT: image tiles per wave, CT: output-column tiles per wave, KT: reduction chunk.
// Software pipelined: the global loads of chunk k+1 are issued before the MFMA phase of chunk k and
// land in registers; they are widened to fp64 and written to LDS after the MFMAs (two barriers per
// chunk, one LDS buffer), so the L2/HBM latency of the operands hides behind the matrix pipe.
// (Double- and triple-buffered LDS variants and the ablation switches of rounds 1-3 are gone: their numbers are in
// profiles/r01_tune_fgemm64*.txt, the kernel is the one-buffer form every launcher uses.)
// TE: storage type of environments and features; TO = 2: contract with the output-site feature
// (forward pass), TO = 1: no output site index (environment shift in strict fp64 mode).
template <int RT, int CT, int WR, int WC, int KT, typename TE = float, int TO = 2>
__global__ __launch_bounds__(64 * WR * WC) void k_fgemm64(Fgemm64Args A) {
    constexpr int T = 64 * WR * WC, BM = 16 * RT * WR, BN = 16 * CT * WC;
    constexpr int XS = BM + 16;                          // doubles; (XS*2) % 64 == 32 -> conflict-free ds_read_b64
    constexpr int MS = BN + ((BN % 32 == 16) ? 0 : 16);
    constexpr int NXI = (KT / 2) * (BM / 4);             // float4 env loads per chunk
    constexpr int NMI = KT * (BN / 2);                   // double2 matrix loads per chunk
    constexpr int NX = (NXI + T - 1) / T, NM = (NMI + T - 1) / T;
    static_assert(T % (BM / 4) == 0, "feature columns must be fixed per thread");
    constexpr int LB = KT * XS + KT * MS;                // doubles per LDS buffer
    __shared__ __attribute__((aligned(16))) double lds[LB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int n0 = A.n_off + blockIdx.x * BM, j0 = blockIdx.y * BN, l = blockIdx.z;
    const TE* E = static_cast<const TE*>(A.EI) + (size_t)l * A.EI_lstride;
    const TE* phiI = static_cast<const TE*>(A.phiI);
    const TE* phiO = static_cast<const TE*>(A.phiO);
    const double* M = A.M + (size_t)l * A.M_lstride;
    const int NTp = A.NTp;

    // this thread's feature columns never change across chunks
    const int xc4 = tid % (BM / 4);
    const V4<TE> p0 = V4<TE>::load(phiI + n0 + xc4 * 4);
    const V4<TE> p1 = V4<TE>::load(phiI + NTp + n0 + xc4 * 4);

    f64x4 acc[CT][RT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[c][r] = f64x4{0., 0., 0., 0.};

    V4<TE> xr[NX];
    double2 mr[NM];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int idx = tid + q * T;
            const int ar = idx / (BM / 4);
            const int a = k0 / 2 + ar;
            xr[q] = V4<TE>::zero();
            if (idx < NXI && a < A.mI) xr[q] = V4<TE>::load(E + (size_t)a * NTp + n0 + xc4 * 4);   // (non-temporal: no gain, matrix-pipe bound)
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int idx = tid + q * T;
            const int r = idx / (BN / 2), c2 = idx % (BN / 2);
            const int j = j0 + c2 * 2;
            mr[q] = make_double2(0., 0.);
            if (idx < NMI && j < A.Np) mr[q] = *reinterpret_cast<const double2*>(M + (size_t)(k0 + r) * A.Np + j);
        }
    };
    auto store_chunk = [&](double* Xs, double* Ms) {
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int idx = tid + q * T;
            if (idx < NXI) {
                const int ar = idx / (BM / 4);
                const V4<TE> e = xr[q];
                double* x0 = &Xs[(2 * ar) * XS + xc4 * 4];
                double* x1 = &Xs[(2 * ar + 1) * XS + xc4 * 4];
                *reinterpret_cast<double2*>(x0) = make_double2(e.x() * p0.x(), e.y() * p0.y());
                *reinterpret_cast<double2*>(x0 + 2) = make_double2(e.z() * p0.z(), e.w() * p0.w());
                *reinterpret_cast<double2*>(x1) = make_double2(e.x() * p1.x(), e.y() * p1.y());
                *reinterpret_cast<double2*>(x1 + 2) = make_double2(e.z() * p1.z(), e.w() * p1.w());
            }
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int idx = tid + q * T;
            if (idx < NMI) {
                const int r = idx / (BN / 2), c2 = idx % (BN / 2);
                *reinterpret_cast<double2*>(&Ms[r * MS + c2 * 2]) = mr[q];
            }
        }
    };

    auto frag_load = [&](const double* Xb, const double* Mb, int ks, double* xf, double* mf) {
        const int krow = 4 * ks + (lane >> 4);
#pragma unroll
        for (int r = 0; r < RT; ++r) xf[r] = Xb[krow * XS + (wr * RT + r) * 16 + (lane & 15)];
#pragma unroll
        for (int c = 0; c < CT; ++c) mf[c] = Mb[krow * MS + (wc * CT + c) * 16 + (lane & 15)];
    };
    auto mfma_step = [&](const double* xf, const double* mf) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < RT; ++r) {        // D[i = column j of M][j = image]
                acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(mf[c], xf[r], acc[c][r], 0, 0, 0);
            }
    };
    constexpr int KS = KT / 4;
    static_assert(KS % 2 == 0, "fragment double buffer assumes an even number of k-steps per chunk");
    double xf[2][RT], mf[2][CT];

    load_chunk(0);
    store_chunk(lds, lds + KT * XS);
    __syncthreads();
    const double* Xb = lds;
    const double* Mb = lds + KT * XS;
    for (int k0 = 0; k0 < A.Kp; k0 += KT) {
        const bool more = k0 + KT < A.Kp;
        if (more) load_chunk(k0 + KT);                       // in flight during the MFMA phase
        frag_load(Xb, Mb, 0, xf[0], mf[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) frag_load(Xb, Mb, ks + 1, xf[(ks + 1) & 1], mf[(ks + 1) & 1]);
            mfma_step(xf[ks & 1], mf[ks & 1]);
        }
        if (more) {
            __syncthreads();                                 // every wave is done reading this chunk
            store_chunk(lds, lds + KT * XS);
            __syncthreads();
        }
    }

    // epilogue: lane (g = lane>>4, i = lane&15) holds rows g+4e of column-tile c for image i of tile r;
    // rows 2q, 2q+1 (site index t = 0,1 of output link q) sit on lane groups g and g^1
    double* out = A.out + (size_t)l * A.out_lstride;
    const int g = lane >> 4;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int n = n0 + (wr * RT + r) * 16 + (lane & 15);
        const double ph = TO == 2 ? (double)phiO[(size_t)(g & 1) * NTp + n] : 1.;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + (wc * CT + c) * 16 + g + 4 * e;
                if (TO == 2) {
                    double v = acc[c][r][e] * ph;
                    v += __shfl_xor(v, 16);
                    const int q = j >> 1;
                    if ((g & 1) == 0 && q < A.mO) out[(size_t)q * NTp + n] = v;
                } else if (A.out32) {                                        // fp32-stored environment: fp64 arithmetic, ONE rounding on the store
                    if (j < A.mO) (reinterpret_cast<float*>(A.out) + (size_t)l * A.out_lstride)[(size_t)j * NTp + n] = (float)acc[c][r][e];
                } else {
                    if (j < A.mO) out[(size_t)j * NTp + n] = acc[c][r][e];     // (non-temporal stores: no gain)
                }
            }
        }
    }
}

template <int RT, int CT, int WR, int WC, int KT>
static void fgemm64_go(tnml_ctx* c, const Fgemm64Args& a) {
    constexpr int BM = 16 * RT * WR, BN = 16 * CT * WC;
    const int cnt = a.n_cnt ? a.n_cnt : a.NTp;
    dim3 grid((cnt + BM - 1) / BM, (a.Np + BN - 1) / BN, a.L);
    dim3 block(64 * WR * WC);
    hipStream_t st = a.st ? a.st : c->stream;
    if (!a.phiO) {                                        // shift form (TO = 1): no second feature on the columns
        if constexpr (CT != 5) {
            if (a.env64) hipLaunchKernelGGL((k_fgemm64<RT, CT, WR, WC, KT, double, 1>), grid, block, 0, st, a);
            else         hipLaunchKernelGGL((k_fgemm64<RT, CT, WR, WC, KT, float, 1>), grid, block, 0, st, a);
        }
    }
    else if (!a.env64) hipLaunchKernelGGL((k_fgemm64<RT, CT, WR, WC, KT, float, 2>), grid, block, 0, st, a);
    else               hipLaunchKernelGGL((k_fgemm64<RT, CT, WR, WC, KT, double, 2>), grid, block, 0, st, a);
}

int launch_fgemm64(tnml_ctx* c, const Fgemm64Args& a) {
    ProfScope ps(c, a.kclass >= 0 ? a.kclass : (a.phiO ? KC_FGEMM_FWD : KC_FGEMM_SHIFT), a.st);
    if (a.NTp % TNML_NTPAD) return tnml_fail(c, "fgemm64: NTp not padded");
    if (a.n_cnt && (a.n_cnt % 128 || a.n_off % 128)) return tnml_fail(c, "fgemm64: image range must be a multiple of 128");
    // Tile choices: the winners of the tuning runs recorded under profiles/ (r01_tune_fgemm64*.txt, r01_tune_shift.txt, r02_tune_m60.txt,
    // r03_tune_m300.txt, tools/tune_shard.sh); the losing instantiations are gone.  Option "fg64_cfg" = 2 forces the large-image-count
    // tiles at any image count (the parity tests run BASELINE config 3's instantiations at oracle-sized image counts).
    const bool big = c->opt_fg64_cfg == 2;
    if (a.Np == 240 && a.phiO) {                             // m = 120: exactly 15 column tiles, no padding waste
        // a rank with few images (multi-GPU shards, small sets) gets smaller row tiles so that every CU has a workgroup
        if (big || a.NTp / 128 >= 192) fgemm64_go<2, 5, 4, 3, 16>(c, a);   // 128 x 240, 12 waves
        else if (a.NTp / 64 >= 192)    fgemm64_go<1, 5, 4, 3, 16>(c, a);   // 64 x 240, 12 waves
        else                           fgemm64_go<1, 4, 4, 2, 16>(c, a);   // 64 x 128 (two column tiles), 8 waves: 7500 images 30.5 TF vs 26.2 with 32 x 240
    }
    else if (a.Np > 64 && !a.phiO) fgemm64_go<2, 4, 4, 2, 8>(c, a);         // shift form at m up to 128 (Label-carrying: grid.z = 10): 128 x 128, 8 waves, KT 8
    else if (a.Np > 256)           fgemm64_go<2, 5, 4, 2, 16>(c, a);        // forward pass at maxm > 120 (BASELINE config 5: 600 columns): 128 x 160, 8 waves
    else if (a.Np > 64) {                                     // forward pass at 32 < m <= 64 (bonds that have shrunk towards minm)
        if (big || a.NTp >= 128 * 192) fgemm64_go<2, 4, 4, 2, 8>(c, a);    // 128 x 128, 8 waves, KT 8: 46 us at m = 60, 60 000 images
        else                           fgemm64_go<2, 4, 2, 2, 16>(c, a);   // 64 x 128, 4 waves
    }
    else if (a.Np > 32)   fgemm64_go<2, 2, 2, 2, 16>(c, a);     // 64 x 64
    else                  fgemm64_go<2, 1, 2, 2, 16>(c, a);     // 64 x 32
    HIPCK(c, hipGetLastError());
    return 0;
}

struct Bgemm64KArgs {
    Bgemm64Args a;
    double* slab;
    int nsplit, imgs_per_split;
    int nt;                       // non-temporal loads of the Label-carrying environment (read once per launch)
};

// Software pipelined like k_fgemm64: the image chunk n+1 is fetched into registers while chunk n feeds
// the matrix pipe from LDS.
// FUSE: the B operand is built from the Label-carrying environment itself,
//   Z[q][n] = sum_l EL[l][q][n] * dP[l][n]   (the first half of dP*dag(t.v), fixedL.cc:379,418),
// so the separate k_zprime pass (a second full stream of EL plus a Z' round trip) disappears and the
// HBM stream of EL overlaps the matrix pipe.
template <int RT, int CT, int WR, int WC, int FUSE = 0, typename TE = float>
__global__ __launch_bounds__(64 * WR * WC) void k_bgemm64(Bgemm64KArgs K) {
    constexpr int T = 64 * WR * WC, BMr = 16 * RT * WR, BNc = 16 * CT * WC, KTn = 32, ST = KTn + 2;   // doubles
    constexpr int NAI = (BMr / 2) * (KTn / 4), NBI = (BNc / 2) * (KTn / 4);
    constexpr int NA = (NAI + T - 1) / T, NB = (NBI + T - 1) / T;
    static_assert(!FUSE || T >= TNML_NL * KTn, "dP tile needs one lane per entry");
    static_assert(!FUSE || T >= 4 * KTn, "feature tile needs one lane per entry");
    __shared__ __attribute__((aligned(16))) double lds[(BMr + BNc) * ST + (FUSE ? (TNML_NL + 4) * KTn : 0)];
    double* As = lds;
    double* Bs = lds + BMr * ST;
    const Bgemm64Args& A = K.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int i0 = blockIdx.x * BMr, j0 = blockIdx.y * BNc;
    const int zs = blockIdx.z % K.nsplit, l = blockIdx.z / K.nsplit;
    const int split = zs;
    const int NTp = A.NTp;
    const int nbeg = split * K.imgs_per_split;
    const int nend = min(nbeg + K.imgs_per_split, NTp);
    const double* w = A.w ? A.w + (size_t)l * A.w_lstride : nullptr;
    const TE* EIp = static_cast<const TE*>(A.EI);
    const TE* ELp = static_cast<const TE*>(A.EL);
    const TE* Zq32 = static_cast<const TE*>(A.Zq32);
    const TE* phiI = static_cast<const TE*>(A.phiI);
    const TE* phiO = static_cast<const TE*>(A.phiO);

    f64x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f64x4{0., 0., 0., 0.};

    V4<TE> ea[NA], pa0[NA], pa1[NA];
    double zb[NB][4]; V4<TE> pb0[NB], pb1[NB];
    constexpr int NFI = (BNc / 2) * (KTn / 2), NF = (NFI + T - 1) / T;   // FUSE: B tasks are (q-row, image pair)
    V2<TE> el[FUSE ? NF : 1][FUSE ? TNML_NL : 1];   // Label-carrying env rows of the chunk in flight (FUSE)
    double dpr = 0.;                               // this lane's entry of the dP tile [10][KTn] (FUSE)
    double* dPs = lds + (BMr + BNc) * ST;          // [10][KTn]
    double phr = 0.;                               // and of the feature tile [phiI s0, s1, phiO s0, s1][KTn]:
    double* phs = dPs + TNML_NL * KTn;             // staged with dP so the build phase never waits on L2
    auto load_chunk = [&](int nb) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int idx = tid + q * T;
            const int ar = idx / (KTn / 4), c4 = idx % (KTn / 4);
            const int a = i0 / 2 + ar, n = nb + c4 * 4;
            ea[q] = V4<TE>::zero();
            if (idx < NAI && a < A.mI) ea[q] = V4<TE>::load(EIp + (size_t)a * NTp + n);
            if (!FUSE) {
                pa0[q] = V4<TE>::load(phiI + n);
                pa1[q] = V4<TE>::load(phiI + NTp + n);
            }
        }
        if (FUSE) {
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                const int idx = tid + q * T;
                const int qr = idx / (KTn / 2), c2 = idx % (KTn / 2);
                const int qq = j0 / 2 + qr, n = nb + c2 * 2;
#pragma unroll
                for (int ll = 0; ll < TNML_NL; ++ll) {
                    el[q][ll] = V2<TE>::zero();
                    if (idx < NFI && qq < A.mO) {
                        const TE* ep = ELp + (size_t)ll * A.EL_lstride + (size_t)qq * NTp + n;
                        el[q][ll] = K.nt ? V2<TE>::load_nt(ep) : V2<TE>::load(ep);
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int idx = tid + q * T;
                const int qr = idx / (KTn / 4), c4 = idx % (KTn / 4);
                const int qq = j0 / 2 + qr, n = nb + c4 * 4;
                zb[q][0] = zb[q][1] = zb[q][2] = zb[q][3] = 0.;
                if (idx < NBI && qq < A.mO) {
                    if (A.Zq64) {
                        const double2 za = *reinterpret_cast<const double2*>(A.Zq64 + (size_t)qq * NTp + n);
                        const double2 zc = *reinterpret_cast<const double2*>(A.Zq64 + (size_t)qq * NTp + n + 2);
                        zb[q][0] = za.x; zb[q][1] = za.y; zb[q][2] = zc.x; zb[q][3] = zc.y;
                    } else {
                        const V4<TE> zf = V4<TE>::load(Zq32 + (size_t)qq * NTp + n);
                        zb[q][0] = zf.x(); zb[q][1] = zf.y(); zb[q][2] = zf.z(); zb[q][3] = zf.w();
                    }
                }
                if (w) {
                    const double2 wa = *reinterpret_cast<const double2*>(w + n);
                    const double2 wb = *reinterpret_cast<const double2*>(w + n + 2);
                    zb[q][0] *= wa.x; zb[q][1] *= wa.y; zb[q][2] *= wb.x; zb[q][3] *= wb.y;
                }
                pb0[q] = V4<TE>::load(phiO + n);
                pb1[q] = V4<TE>::load(phiO + NTp + n);
            }
        }
        if (FUSE && tid < TNML_NL * KTn) dpr = A.dPz[(size_t)(tid / KTn) * NTp + nb + (tid % KTn)];
        if (FUSE && tid < 4 * KTn) {
            const int wh = tid / KTn, n = nb + (tid % KTn);
            phr = (double)((wh < 2 ? phiI : phiO)[(size_t)(wh & 1) * NTp + n]);
        }
    };
    auto store_dp = [&]() {
        if (FUSE && tid < TNML_NL * KTn) dPs[tid] = dpr;
        if (FUSE && tid < 4 * KTn) phs[tid] = phr;
    };
    auto store_chunk = [&](int nb) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int idx = tid + q * T;
            if (idx < NAI) {
                const int ar = idx / (KTn / 4), c4 = idx % (KTn / 4);
                const V4<TE> e = ea[q];
                double* x0 = &As[(2 * ar) * ST + c4 * 4];
                double* x1 = &As[(2 * ar + 1) * ST + c4 * 4];
                if (FUSE) {
                    const double2 p0a = *reinterpret_cast<const double2*>(&phs[c4 * 4]), p0b = *reinterpret_cast<const double2*>(&phs[c4 * 4 + 2]);
                    const double2 p1a = *reinterpret_cast<const double2*>(&phs[KTn + c4 * 4]), p1b = *reinterpret_cast<const double2*>(&phs[KTn + c4 * 4 + 2]);
                    *reinterpret_cast<double2*>(x0) = make_double2(e.x() * p0a.x, e.y() * p0a.y);
                    *reinterpret_cast<double2*>(x0 + 2) = make_double2(e.z() * p0b.x, e.w() * p0b.y);
                    *reinterpret_cast<double2*>(x1) = make_double2(e.x() * p1a.x, e.y() * p1a.y);
                    *reinterpret_cast<double2*>(x1 + 2) = make_double2(e.z() * p1b.x, e.w() * p1b.y);
                } else {
                    const V4<TE> p0 = pa0[q], p1 = pa1[q];
                    *reinterpret_cast<double2*>(x0) = make_double2(e.x() * p0.x(), e.y() * p0.y());
                    *reinterpret_cast<double2*>(x0 + 2) = make_double2(e.z() * p0.z(), e.w() * p0.w());
                    *reinterpret_cast<double2*>(x1) = make_double2(e.x() * p1.x(), e.y() * p1.y());
                    *reinterpret_cast<double2*>(x1 + 2) = make_double2(e.z() * p1.z(), e.w() * p1.w());
                }
            }
        }
        if (FUSE) {
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                const int idx = tid + q * T;
                if (idx < NFI) {
                    const int qr = idx / (KTn / 2), c2 = idx % (KTn / 2);
                    double z0 = 0., z1 = 0.;
#pragma unroll
                    for (int ll = 0; ll < TNML_NL; ++ll) {
                        const double2 d = *reinterpret_cast<const double2*>(&dPs[ll * KTn + c2 * 2]);
                        z0 = fma(el[q][ll].x(), d.x, z0);
                        z1 = fma(el[q][ll].y(), d.y, z1);
                    }
                    const double2 f0 = *reinterpret_cast<const double2*>(&phs[2 * KTn + c2 * 2]);
                    const double2 f1 = *reinterpret_cast<const double2*>(&phs[3 * KTn + c2 * 2]);
                    *reinterpret_cast<double2*>(&Bs[(2 * qr) * ST + c2 * 2]) = make_double2(z0 * f0.x, z1 * f0.y);
                    *reinterpret_cast<double2*>(&Bs[(2 * qr + 1) * ST + c2 * 2]) = make_double2(z0 * f1.x, z1 * f1.y);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int idx = tid + q * T;
                if (idx < NBI) {
                    const int qr = idx / (KTn / 4), c4 = idx % (KTn / 4);
                    const V4<TE> p0 = pb0[q], p1 = pb1[q];
                    double* b0 = &Bs[(2 * qr) * ST + c4 * 4];
                    double* b1 = &Bs[(2 * qr + 1) * ST + c4 * 4];
                    *reinterpret_cast<double2*>(b0) = make_double2(zb[q][0] * p0.x(), zb[q][1] * p0.y());
                    *reinterpret_cast<double2*>(b0 + 2) = make_double2(zb[q][2] * p0.z(), zb[q][3] * p0.w());
                    *reinterpret_cast<double2*>(b1) = make_double2(zb[q][0] * p1.x(), zb[q][1] * p1.y());
                    *reinterpret_cast<double2*>(b1 + 2) = make_double2(zb[q][2] * p1.z(), zb[q][3] * p1.w());
                }
            }
        }
    };

    if (nbeg < nend) {
        load_chunk(nbeg);
        if (FUSE) { store_dp(); __syncthreads(); }
        store_chunk(nbeg);
        __syncthreads();
    }
    for (int nb = nbeg; nb < nend; nb += KTn) {
        const bool more = nb + KTn < nend;
        if (more) load_chunk(nb + KTn);
#pragma unroll
        for (int kk = 0; kk < KTn; kk += 8) {
            // lane group g owns images kk+2g, kk+2g+1; MFMA step e uses element e of every group
            double2 a[RT], b[CT];
            const int ko = kk + 2 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = *reinterpret_cast<const double2*>(&As[((wr * RT + r) * 16 + (lane & 15)) * ST + ko]);
#pragma unroll
            for (int c = 0; c < CT; ++c) b[c] = *reinterpret_cast<const double2*>(&Bs[((wc * CT + c) * 16 + (lane & 15)) * ST + ko]);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r].x, b[c].x, acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r].y, b[c].y, acc[r][c], 0, 0, 0);
                }
        }
        if (more) {
            __syncthreads();
            if (FUSE) { store_dp(); __syncthreads(); }      // the dP tile feeds the Z build below
            store_chunk(nb + KTn);
            __syncthreads();
        }
    }

    double* slab = K.slab + ((size_t)split * A.L + l) * A.Kp * A.Np;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int j = j0 + (wc * CT + c) * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + (wr * RT + r) * 16 + (lane >> 4) + 4 * e;     // f64 C map: row = g + 4*reg
                if (i < A.Kp && j < A.Np) slab[(size_t)i * A.Np + j] = acc[r][c][e];
            }
        }
}

__global__ void k_slab_reduce64(const double* __restrict__ slab, double* __restrict__ G, size_t n, int nsplit) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.;
#pragma unroll 8
    for (int k = 0; k < nsplit; ++k) s += slab[(size_t)k * n + i];      // same order, eight loads in flight
    G[i] = s;
}

template <int RT, int CT, int WR, int WC, int FUSE = 0>
static int bgemm64_go(tnml_ctx* c, const Bgemm64Args& a, double* G, int default_wgs = 768) {
    constexpr int BMr = 16 * RT * WR, BNc = 16 * CT * WC;
    const int tiles = ((a.Kp + BMr - 1) / BMr) * ((a.Np + BNc - 1) / BNc) * a.L;
    const int target_wgs = c->bgemm_wgs > 0 ? c->bgemm_wgs : default_wgs;
    int nsplit = (target_wgs + tiles - 1) / tiles;
    const int chunks = a.NTp / 32;
    if (nsplit > chunks) nsplit = chunks;
    if (nsplit < 1) nsplit = 1;
    const size_t n = (size_t)a.L * a.Kp * a.Np;
    const size_t cap = c->slab_bytes / sizeof(double);
    while (nsplit > 1 && (size_t)nsplit * n > cap) --nsplit;
    if ((size_t)nsplit * n > cap) return tnml_fail(c, "bgemm64: slab workspace too small");
    int per = ((chunks + nsplit - 1) / nsplit) * 32;
    if (c->bgemm_per > 0 && (size_t)((a.NTp + c->bgemm_per * 32 - 1) / (c->bgemm_per * 32)) * n <= cap) per = c->bgemm_per * 32;      // probe knob: images per slab / 32
    nsplit = (a.NTp + per - 1) / per;
    Bgemm64KArgs K{a, (double*)c->slab, nsplit, per, 1};       // non-temporal loads of the Label-carrying environment: +1.3 % (profiles/r01_ab_nt_loads.txt)
    {
        ProfScope ps(c, KC_BGEMM);
        dim3 grid((a.Kp + BMr - 1) / BMr, (a.Np + BNc - 1) / BNc, nsplit * a.L);
        if (a.env64) hipLaunchKernelGGL((k_bgemm64<RT, CT, WR, WC, FUSE, double>), grid, dim3(64 * WR * WC), 0, c->stream, K);
        else         hipLaunchKernelGGL((k_bgemm64<RT, CT, WR, WC, FUSE, float>), grid, dim3(64 * WR * WC), 0, c->stream, K);
    }
    if (c->defer_slab) c->slab_pending = nsplit;              // the CG vector kernel that consumes G sums the slabs itself (same order: same bits)
    else {
        ProfScope ps(c, KC_SLABRED);
        hipLaunchKernelGGL(k_slab_reduce64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->slab, G, n, nsplit);
    }
    HIPCK(c, hipGetLastError());
    return 0;
}

void launch_slab_reduce64(tnml_ctx* c, const double* slab, double* G, size_t n, int nsplit) {
    hipLaunchKernelGGL(k_slab_reduce64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, slab, G, n, nsplit);
}

int launch_bgemm64(tnml_ctx* c, const Bgemm64Args& a, double* G) {
    // Tile choices: the winners of the tuning runs recorded under profiles/ (r01 tune_bgemm, r02_tune_m60.txt, r03_tune_m300.txt,
    // r03_tune_bgemm_wide_tiles.txt, r03_ab_bgemm_double_buffer_and_ablation.txt); the losing instantiations are gone.
    if (grad_quad_applies(c, a)) return launch_grad_quad(c, a, G);
    if (a.EL) {                                             // fused Z build: >= 320 lanes per workgroup
        if (a.Kp % 240 == 0 && a.Np % 240 == 0) return bgemm64_go<5, 1, 3, 4, 1>(c, a, G, 256 * a.L);   // 240 x 64, 12 waves: 181 us vs 153 + 90 unfused
        if (a.Kp % 80 == 0 && a.Np % 80 == 0)   return bgemm64_go<1, 5, 5, 1, 1>(c, a, G);               // 80 x 80, 5 waves
        if (a.Kp % 128 == 0 && a.Np % 64 == 0)  return bgemm64_go<2, 2, 4, 2, 1>(c, a, G);               // m = 33..64: 128 x 64, 8 waves: 70 us at m = 60
        if (a.Kp >= 256 && a.Np >= 256)         return bgemm64_go<5, 1, 2, 6, 1>(c, a, G);               // maxm > 120 (BASELINE config 5): 160 x 96, 12 waves: 198 us at m = 300, 7 500 images
        return bgemm64_go<2, 2, 3, 2, 1>(c, a, G);                                                     // 96 x 64, 6 waves
    }
    if (a.Kp % 240 == 0 && a.Np % 240 == 0) return bgemm64_go<5, 1, 3, 5>(c, a, G, 255 * a.L);          // 240 x 80 tiles, 15 waves, one workgroup per CU
    if (a.Kp % 80 == 0 && a.Np % 80 == 0) return bgemm64_go<1, 5, 5, 1>(c, a, G);
    if (a.Kp > 32 && a.Np > 32) return bgemm64_go<2, 2, 2, 2>(c, a, G);
    return bgemm64_go<1, 1, 2, 2>(c, a, G);
}
