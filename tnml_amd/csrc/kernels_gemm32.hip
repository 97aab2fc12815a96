// kernels_gemm32.hip -- the reduced-precision study modes of the two image-proportional GEMMs (TNML_F32, TNML_BF16, TNML_BF16X3;
// BASELINE config 5's "bf16 MFMA bond contraction vs fp32"): k_fgemm (feature GEMM / environment shift on v_mfma_f32_16x16x4_f32),
// k_fgemm_bf16 (forward feature GEMM on v_mfma_f32_16x16x32_bf16, plain and hi + lo split), k_bgemm (gradient GEMM, fp32 MFMA),
// k_bgemm_bf16 (gradient GEMM on the bf16 pipe).
// The default arithmetic (fp64 MFMA) lives in kernels_gemm.hip; the algebra and the tile maps are the same.
#include "tnml_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// ------------------------------------------------------------------------------------------
template <int RT, int CT, int WR, int WC, int TO>
__global__ __launch_bounds__(64 * WR * WC) void k_fgemm(FgemmArgs A) {
    constexpr int T = 64 * WR * WC, BM = 16 * RT * WR, BN = 16 * CT * WC, KT = 16;
    constexpr int XS = BM + 16;                         // row stride == 16 (mod 32): conflict-free fragment reads
    constexpr int MS = BN + ((BN % 32 == 16) ? 0 : 16);
    __shared__ __attribute__((aligned(16))) float lds[KT * XS + KT * MS];
    float* Xs = lds;
    float* Ms = lds + KT * XS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int n0 = blockIdx.x * BM, j0 = blockIdx.y * BN, l = blockIdx.z;
    const float* E = A.EI + (size_t)l * A.EI_lstride;
    const float* M = A.M + (size_t)l * A.M_lstride;
    const int NTp = A.NTp;

    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < A.Kp; k0 += KT) {
        // stage X: KT/2 environment rows, each expanded to its two site-index rows
        for (int idx = tid; idx < (KT / 2) * (BM / 4); idx += T) {
            const int ar = idx / (BM / 4), c4 = idx % (BM / 4);
            const int a = k0 / 2 + ar, n = n0 + c4 * 4;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a < A.mI) e = *reinterpret_cast<const float4*>(E + (size_t)a * NTp + n);
            const float4 p0 = *reinterpret_cast<const float4*>(A.phiI + n);
            const float4 p1 = *reinterpret_cast<const float4*>(A.phiI + NTp + n);
            *reinterpret_cast<float4*>(&Xs[(2 * ar) * XS + c4 * 4]) = mul4(e, p0);
            *reinterpret_cast<float4*>(&Xs[(2 * ar + 1) * XS + c4 * 4]) = mul4(e, p1);
        }
        // stage M: KT rows of the (zero padded) bond matrix
        for (int idx = tid; idx < KT * (BN / 4); idx += T) {
            const int r = idx / (BN / 4), c4 = idx % (BN / 4);
            const int j = j0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < A.Np) v = *reinterpret_cast<const float4*>(M + (size_t)(k0 + r) * A.Np + j);
            *reinterpret_cast<float4*>(&Ms[r * MS + c4 * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            float a[RT], b[CT];
            const int krow = kk + (lane >> 4);
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = Xs[krow * XS + (wr * RT + r) * 16 + (lane & 15)];
#pragma unroll
            for (int c = 0; c < CT; ++c) b[c] = Ms[krow * MS + (wc * CT + c) * 16 + (lane & 15)];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[c], acc[r][c], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C fragment = 4 consecutive images (rows) x 1 column per lane
    float* out = A.out + (size_t)l * A.out_lstride;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int n = n0 + (wr * RT + r) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int j = j0 + (wc * CT + c) * 16 + (lane & 15);
            if (TO == 2) {
                const int t = j & 1, q = j >> 1;
                const float4 ph = *reinterpret_cast<const float4*>(A.phiO + (size_t)t * NTp + n);
                float4 v = make_float4(acc[r][c][0] * ph.x, acc[r][c][1] * ph.y, acc[r][c][2] * ph.z, acc[r][c][3] * ph.w);
                v.x += __shfl_xor(v.x, 1); v.y += __shfl_xor(v.y, 1);
                v.z += __shfl_xor(v.z, 1); v.w += __shfl_xor(v.w, 1);
                if (t == 0 && q < A.mO) *reinterpret_cast<float4*>(out + (size_t)q * NTp + n) = v;
            } else {
                if (j < A.mO)
                    *reinterpret_cast<float4*>(out + (size_t)j * NTp + n) = make_float4(acc[r][c][0], acc[r][c][1], acc[r][c][2], acc[r][c][3]);
            }
        }
    }
}

template <int RT, int CT, int WR, int WC>
static void fgemm_go(tnml_ctx* c, const FgemmArgs& a) {
    constexpr int BM = 16 * RT * WR, BN = 16 * CT * WC;
    dim3 grid(a.NTp / BM, (a.Np + BN - 1) / BN, a.L);
    dim3 block(64 * WR * WC);
    if (a.phiO) hipLaunchKernelGGL((k_fgemm<RT, CT, WR, WC, 2>), grid, block, 0, c->stream, a);
    else        hipLaunchKernelGGL((k_fgemm<RT, CT, WR, WC, 1>), grid, block, 0, c->stream, a);
}

// ------------------------------------------------------------------------------------------
// k_fgemm_bf16 -- the forward feature GEMM on the bf16 matrix pipe (TNML_BF16 / TNML_BF16X3; BASELINE config 5's "bf16 MFMA
// bond contraction", a tolerance study): same tiling, same epilogue and the same C-fragment map as k_fgemm
// (v_mfma_f32_16x16x32_bf16: col = lane & 15, row = 4 (lane >> 4) + reg), fp32 storage, fp32 accumulation.  The operands are
// rounded to bf16 (round to nearest even) while they are staged: X_n = EI_n (x) phiI_n is formed in fp32 and then rounded, so
// is every element of the bond matrix.  An A / B fragment is 8 consecutive reduction indices of one row / column
// (k = 8 (lane >> 4) + 0..7), so the LDS tiles are [row][k] with k contiguous (80-byte rows: the four lane groups of a
// 16-byte fragment read land on distinct banks).  SPLIT: every operand x = hi + lo with hi = bf16(x), lo = bf16(x - hi) and
// three MFMAs per product, hi*hi + hi*lo + lo*hi (the lo*lo term is below fp32 round-off): ~16 mantissa bits.
// ------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ unsigned short f2bf(float x) {
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static __device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <int RT, int CT, int WR, int WC, int TO, int SPLIT>
__global__ __launch_bounds__(64 * WR * WC) void k_fgemm_bf16(FgemmArgs A) {
    constexpr int T = 64 * WR * WC, BM = 16 * RT * WR, BN = 16 * CT * WC, KT = 32, KS = KT + 8;     // KS: row stride in bf16 elements
    constexpr int NP = SPLIT ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short lds[NP * (BM + BN) * KS];
    unsigned short* Xs = lds;                               // [NP][BM][KS]
    unsigned short* Ms = lds + NP * BM * KS;                // [NP][BN][KS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int n0 = blockIdx.x * BM, j0 = blockIdx.y * BN, l = blockIdx.z;
    const float* E = A.EI + (size_t)l * A.EI_lstride;
    const float* M = A.M + (size_t)l * A.M_lstride;
    const int NTp = A.NTp;

    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < A.Kp; k0 += KT) {
        // stage X: KT/2 environment rows, each expanded to its two site-index rows, transposed to [image][k]
        for (int idx = tid; idx < (KT / 2) * (BM / 4); idx += T) {
            const int ar = idx / (BM / 4), c4 = idx % (BM / 4);
            const int a = k0 / 2 + ar, n = n0 + c4 * 4;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a < A.mI) e = *reinterpret_cast<const float4*>(E + (size_t)a * NTp + n);
            const float4 p0 = *reinterpret_cast<const float4*>(A.phiI + n);
            const float4 p1 = *reinterpret_cast<const float4*>(A.phiI + NTp + n);
            const float x0[4] = {e.x * p0.x, e.y * p0.y, e.z * p0.z, e.w * p0.w};
            const float x1[4] = {e.x * p1.x, e.y * p1.y, e.z * p1.z, e.w * p1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = c4 * 4 + q;
                const unsigned short h0 = f2bf(x0[q]), h1 = f2bf(x1[q]);
                *reinterpret_cast<unsigned*>(&Xs[row * KS + 2 * ar]) = (unsigned)h0 | ((unsigned)h1 << 16);
                if (SPLIT) {
                    const unsigned short l0 = f2bf(x0[q] - bf2f(h0)), l1 = f2bf(x1[q] - bf2f(h1));
                    *reinterpret_cast<unsigned*>(&Xs[BM * KS + row * KS + 2 * ar]) = (unsigned)l0 | ((unsigned)l1 << 16);
                }
            }
        }
        // stage M: KT rows of the (zero padded) bond matrix, transposed to [column][k]
        for (int idx = tid; idx < KT * (BN / 4); idx += T) {
            const int r = idx / (BN / 4), c4 = idx % (BN / 4);
            const int j = j0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < A.Np) v = *reinterpret_cast<const float4*>(M + (size_t)(k0 + r) * A.Np + j);
            const float mv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned short h = f2bf(mv[q]);
                Ms[(c4 * 4 + q) * KS + r] = h;
                if (SPLIT) Ms[BN * KS + (c4 * 4 + q) * KS + r] = f2bf(mv[q] - bf2f(h));
            }
        }
        __syncthreads();
        {
            const int ko = 8 * (lane >> 4);
            bf16x8 ah[RT], bh[CT], al[SPLIT ? RT : 1], bl[SPLIT ? CT : 1];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                ah[r] = *reinterpret_cast<const bf16x8*>(&Xs[((wr * RT + r) * 16 + (lane & 15)) * KS + ko]);
                if (SPLIT) al[r] = *reinterpret_cast<const bf16x8*>(&Xs[BM * KS + ((wr * RT + r) * 16 + (lane & 15)) * KS + ko]);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                bh[c] = *reinterpret_cast<const bf16x8*>(&Ms[((wc * CT + c) * 16 + (lane & 15)) * KS + ko]);
                if (SPLIT) bl[c] = *reinterpret_cast<const bf16x8*>(&Ms[BN * KS + ((wc * CT + c) * 16 + (lane & 15)) * KS + ko]);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (SPLIT) {                                   // small terms first
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[r], bh[c], acc[r][c], 0, 0, 0);
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bl[c], acc[r][c], 0, 0, 0);
                    }
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bh[c], acc[r][c], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // epilogue: C fragment = 4 consecutive images (rows) x 1 column per lane (as k_fgemm)
    float* out = A.out + (size_t)l * A.out_lstride;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int n = n0 + (wr * RT + r) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int j = j0 + (wc * CT + c) * 16 + (lane & 15);
            if (TO == 2) {
                const int t = j & 1, q = j >> 1;
                const float4 ph = *reinterpret_cast<const float4*>(A.phiO + (size_t)t * NTp + n);
                float4 v = make_float4(acc[r][c][0] * ph.x, acc[r][c][1] * ph.y, acc[r][c][2] * ph.z, acc[r][c][3] * ph.w);
                v.x += __shfl_xor(v.x, 1); v.y += __shfl_xor(v.y, 1);
                v.z += __shfl_xor(v.z, 1); v.w += __shfl_xor(v.w, 1);
                if (t == 0 && q < A.mO) *reinterpret_cast<float4*>(out + (size_t)q * NTp + n) = v;
            } else {
                if (j < A.mO)
                    *reinterpret_cast<float4*>(out + (size_t)j * NTp + n) = make_float4(acc[r][c][0], acc[r][c][1], acc[r][c][2], acc[r][c][3]);
            }
        }
    }
}

template <int RT, int CT, int WR, int WC>
static void fgemm_bf16_go(tnml_ctx* c, const FgemmArgs& a, int split) {
    constexpr int BM = 16 * RT * WR, BN = 16 * CT * WC;
    dim3 grid(a.NTp / BM, (a.Np + BN - 1) / BN, a.L);
    dim3 block(64 * WR * WC);
    if (split) hipLaunchKernelGGL((k_fgemm_bf16<RT, CT, WR, WC, 2, 1>), grid, block, 0, c->stream, a);
    else       hipLaunchKernelGGL((k_fgemm_bf16<RT, CT, WR, WC, 2, 0>), grid, block, 0, c->stream, a);
}

int launch_fgemm(tnml_ctx* c, const FgemmArgs& a) {
    ProfScope ps(c, a.phiO ? KC_FGEMM_FWD : KC_FGEMM_SHIFT);
    if (a.NTp % TNML_NTPAD) return tnml_fail(c, "fgemm: NTp not padded");
    if (c->bf16() && a.phiO) {                              // forward pass on the bf16 matrix pipe (the reduction runs in chunks of 32: Kp is a multiple of 16, the
        if (a.Kp % 32) return tnml_fail(c, "fgemm (bf16): the padded reduction dimension %d is not a multiple of 32", a.Kp);   // bond plan pads to 32 in these modes)
        if (a.Np > 64) fgemm_bf16_go<4, 4, 2, 2>(c, a, c->bf16() == 2);   // 128 x 128
        else           fgemm_bf16_go<4, 2, 2, 2>(c, a, c->bf16() == 2);   // 128 x 64
        HIPCK(c, hipGetLastError());
        return 0;
    }
    if (a.Np == 240)      fgemm_go<4, 5, 2, 3>(c, a);      // m = 120: exactly 15 column tiles, no padding waste
    else if (a.Np > 64)   fgemm_go<4, 4, 2, 2>(c, a);      // 128 x 128 tiles
    else if (a.Np > 32)   fgemm_go<4, 2, 2, 2>(c, a);      // 128 x 64
    else                  fgemm_go<4, 1, 2, 2>(c, a);      // 128 x 32
    HIPCK(c, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
struct BgemmKArgs {
    BgemmArgs a;
    float* slab;
    int nsplit, imgs_per_split;
};

template <int RT, int CT, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC) void k_bgemm(BgemmKArgs K) {
    constexpr int T = 64 * WR * WC, BMr = 16 * RT * WR, BNc = 16 * CT * WC, KTn = 32, ST = KTn + 4;
    __shared__ __attribute__((aligned(16))) float lds[(BMr + BNc) * ST];
    float* As = lds;
    float* Bs = lds + BMr * ST;
    const BgemmArgs& A = K.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int i0 = blockIdx.x * BMr, j0 = blockIdx.y * BNc;
    const int split = blockIdx.z % K.nsplit, l = blockIdx.z / K.nsplit;
    const int NTp = A.NTp;
    const int nbeg = split * K.imgs_per_split;
    const int nend = min(nbeg + K.imgs_per_split, NTp);
    const float* w = A.w ? A.w + (size_t)l * A.w_lstride : nullptr;

    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int nb = nbeg; nb < nend; nb += KTn) {
        for (int idx = tid; idx < (BMr / 2) * (KTn / 4); idx += T) {
            const int ar = idx / (KTn / 4), c4 = idx % (KTn / 4);
            const int a = i0 / 2 + ar, n = nb + c4 * 4;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a < A.mI) e = *reinterpret_cast<const float4*>(A.EI + (size_t)a * NTp + n);
            const float4 p0 = *reinterpret_cast<const float4*>(A.phiI + n);
            const float4 p1 = *reinterpret_cast<const float4*>(A.phiI + NTp + n);
            *reinterpret_cast<float4*>(&As[(2 * ar) * ST + c4 * 4]) = mul4(e, p0);
            *reinterpret_cast<float4*>(&As[(2 * ar + 1) * ST + c4 * 4]) = mul4(e, p1);
        }
        for (int idx = tid; idx < (BNc / 2) * (KTn / 4); idx += T) {
            const int qr = idx / (KTn / 4), c4 = idx % (KTn / 4);
            const int q = j0 / 2 + qr, n = nb + c4 * 4;
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < A.mO) z = *reinterpret_cast<const float4*>(A.Zq + (size_t)q * NTp + n);
            if (w) z = mul4(z, *reinterpret_cast<const float4*>(w + n));
            const float4 p0 = *reinterpret_cast<const float4*>(A.phiO + n);
            const float4 p1 = *reinterpret_cast<const float4*>(A.phiO + NTp + n);
            *reinterpret_cast<float4*>(&Bs[(2 * qr) * ST + c4 * 4]) = mul4(z, p0);
            *reinterpret_cast<float4*>(&Bs[(2 * qr + 1) * ST + c4 * 4]) = mul4(z, p1);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KTn; kk += 16) {
            // lane group g = lane>>4 owns images kk+4g..kk+4g+3; MFMA step e uses element e of every
            // group (a permutation of the reduction index, identical for A and B)
            float4 a[RT], b[CT];
            const int ko = kk + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = *reinterpret_cast<const float4*>(&As[((wr * RT + r) * 16 + (lane & 15)) * ST + ko]);
#pragma unroll
            for (int c = 0; c < CT; ++c) b[c] = *reinterpret_cast<const float4*>(&Bs[((wc * CT + c) * 16 + (lane & 15)) * ST + ko]);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, b[c].x, acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, b[c].y, acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].z, b[c].z, acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].w, b[c].w, acc[r][c], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    float* slab = K.slab + ((size_t)split * A.L + l) * A.Kp * A.Np;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int j = j0 + (wc * CT + c) * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + (wr * RT + r) * 16 + (lane >> 4) * 4 + e;
                if (i < A.Kp && j < A.Np) slab[(size_t)i * A.Np + j] = acc[r][c][e];
            }
        }
}

// k_bgemm_bf16 -- the gradient GEMM dP*dag(t.v) (fixedL.cc:379,418) on the bf16 matrix pipe (TNML_BF16 / TNML_BF16X3): k_bgemm's tiling,
// split-K slabs and C-fragment map; both operands are formed in fp32 (X_n = EI_n (x) phiI_n, Z_n phiO_n w_n), rounded to bf16 (SPLIT: hi + lo
// with three MFMAs per product, small terms first) while they are staged, fp32 accumulation.  The reduction index is the image, which is
// already the contiguous index of both operands in memory: an A / B fragment is 8 consecutive images of one row (k = 8 (lane >> 4) + 0..7),
// one 32-image chunk is ONE MFMA step per tile.  With k_fgemm_bf16 this puts the whole bond contraction of the bf16 modes on the bf16 pipe.
template <int RT, int CT, int WR, int WC, int SPLIT>
__global__ __launch_bounds__(64 * WR * WC) void k_bgemm_bf16(BgemmKArgs K) {
    constexpr int T = 64 * WR * WC, BMr = 16 * RT * WR, BNc = 16 * CT * WC, KTn = 32, KS = KTn + 8;     // KS: row stride in bf16 elements (80 bytes)
    constexpr int NP = SPLIT ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short lds[NP * (BMr + BNc) * KS];
    unsigned short* As = lds;                               // [NP][BMr][KS]
    unsigned short* Bs = lds + NP * BMr * KS;               // [NP][BNc][KS]
    const BgemmArgs& A = K.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int i0 = blockIdx.x * BMr, j0 = blockIdx.y * BNc;
    const int split = blockIdx.z % K.nsplit, l = blockIdx.z / K.nsplit;
    const int NTp = A.NTp;
    const int nbeg = split * K.imgs_per_split;
    const int nend = min(nbeg + K.imgs_per_split, NTp);
    const float* w = A.w ? A.w + (size_t)l * A.w_lstride : nullptr;

    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto put4 = [&](unsigned short* dst, unsigned short* dst_lo, float4 v) {          // four consecutive images of one row
        const float x[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = f2bf(x[q]); if (SPLIT) lo[q] = f2bf(x[q] - bf2f(h[q])); }
        *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        if (SPLIT) *reinterpret_cast<uint2*>(dst_lo) = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
    };
    for (int nb = nbeg; nb < nend; nb += KTn) {
        for (int idx = tid; idx < (BMr / 2) * (KTn / 4); idx += T) {
            const int ar = idx / (KTn / 4), c4 = idx % (KTn / 4);
            const int a = i0 / 2 + ar, n = nb + c4 * 4;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a < A.mI) e = *reinterpret_cast<const float4*>(A.EI + (size_t)a * NTp + n);
            const float4 p0 = *reinterpret_cast<const float4*>(A.phiI + n);
            const float4 p1 = *reinterpret_cast<const float4*>(A.phiI + NTp + n);
            put4(&As[(2 * ar) * KS + c4 * 4], &As[BMr * KS + (2 * ar) * KS + c4 * 4], mul4(e, p0));
            put4(&As[(2 * ar + 1) * KS + c4 * 4], &As[BMr * KS + (2 * ar + 1) * KS + c4 * 4], mul4(e, p1));
        }
        for (int idx = tid; idx < (BNc / 2) * (KTn / 4); idx += T) {
            const int qr = idx / (KTn / 4), c4 = idx % (KTn / 4);
            const int q = j0 / 2 + qr, n = nb + c4 * 4;
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < A.mO) z = *reinterpret_cast<const float4*>(A.Zq + (size_t)q * NTp + n);
            if (w) z = mul4(z, *reinterpret_cast<const float4*>(w + n));
            const float4 p0 = *reinterpret_cast<const float4*>(A.phiO + n);
            const float4 p1 = *reinterpret_cast<const float4*>(A.phiO + NTp + n);
            put4(&Bs[(2 * qr) * KS + c4 * 4], &Bs[BNc * KS + (2 * qr) * KS + c4 * 4], mul4(z, p0));
            put4(&Bs[(2 * qr + 1) * KS + c4 * 4], &Bs[BNc * KS + (2 * qr + 1) * KS + c4 * 4], mul4(z, p1));
        }
        __syncthreads();
        {
            const int ko = 8 * (lane >> 4);
            bf16x8 ah[RT], bh[CT], al[SPLIT ? RT : 1], bl[SPLIT ? CT : 1];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                ah[r] = *reinterpret_cast<const bf16x8*>(&As[((wr * RT + r) * 16 + (lane & 15)) * KS + ko]);
                if (SPLIT) al[r] = *reinterpret_cast<const bf16x8*>(&As[BMr * KS + ((wr * RT + r) * 16 + (lane & 15)) * KS + ko]);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                bh[c] = *reinterpret_cast<const bf16x8*>(&Bs[((wc * CT + c) * 16 + (lane & 15)) * KS + ko]);
                if (SPLIT) bl[c] = *reinterpret_cast<const bf16x8*>(&Bs[BNc * KS + ((wc * CT + c) * 16 + (lane & 15)) * KS + ko]);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (SPLIT) {
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[r], bh[c], acc[r][c], 0, 0, 0);
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bl[c], acc[r][c], 0, 0, 0);
                    }
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bh[c], acc[r][c], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    float* slab = K.slab + ((size_t)split * A.L + l) * A.Kp * A.Np;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int j = j0 + (wc * CT + c) * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + (wr * RT + r) * 16 + (lane >> 4) * 4 + e;
                if (i < A.Kp && j < A.Np) slab[(size_t)i * A.Np + j] = acc[r][c][e];
            }
        }
}

__global__ void k_slab_reduce(const float* __restrict__ slab, double* __restrict__ G, size_t n, int nsplit) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.;
    for (int k = 0; k < nsplit; ++k) s += (double)slab[(size_t)k * n + i];
    G[i] = s;
}

template <int RT, int CT, int WR, int WC>
static int bgemm_go(tnml_ctx* c, const BgemmArgs& a, double* G) {
    constexpr int BMr = 16 * RT * WR, BNc = 16 * CT * WC;
    const int tiles = ((a.Kp + BMr - 1) / BMr) * ((a.Np + BNc - 1) / BNc) * a.L;
    int nsplit = (1024 + tiles - 1) / tiles;
    const int chunks = a.NTp / 32;
    if (nsplit > chunks) nsplit = chunks;
    if (nsplit < 1) nsplit = 1;
    const size_t n = (size_t)a.L * a.Kp * a.Np;
    const size_t cap = c->slab_bytes / sizeof(float);
    while (nsplit > 1 && (size_t)nsplit * n > cap) --nsplit;
    if ((size_t)nsplit * n > cap) return tnml_fail(c, "bgemm: slab workspace too small");
    int per = ((chunks + nsplit - 1) / nsplit) * 32;
    nsplit = (a.NTp + per - 1) / per;
    BgemmKArgs K{a, (float*)c->slab, nsplit, per};
    {
        ProfScope ps(c, KC_BGEMM);
        dim3 grid((a.Kp + BMr - 1) / BMr, (a.Np + BNc - 1) / BNc, nsplit * a.L);
        if (a.bf16 == 2)      hipLaunchKernelGGL((k_bgemm_bf16<RT, CT, WR, WC, 1>), grid, dim3(64 * WR * WC), 0, c->stream, K);
        else if (a.bf16 == 1) hipLaunchKernelGGL((k_bgemm_bf16<RT, CT, WR, WC, 0>), grid, dim3(64 * WR * WC), 0, c->stream, K);
        else                  hipLaunchKernelGGL((k_bgemm<RT, CT, WR, WC>), grid, dim3(64 * WR * WC), 0, c->stream, K);
    }
    {
        ProfScope ps(c, KC_SLABRED);
        hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->slab, G, n, nsplit);
    }
    HIPCK(c, hipGetLastError());
    return 0;
}

int launch_bgemm(tnml_ctx* c, const BgemmArgs& a, double* G) {
    if (a.Kp % 80 == 0 && a.Np % 80 == 0) return bgemm_go<1, 5, 5, 1>(c, a, G);   // 80 x 80 tiles (m = 40k: 240 = 3*80)
    if (a.Kp > 32 && a.Np > 32) return bgemm_go<2, 2, 2, 2>(c, a, G);             // 64 x 64
    return bgemm_go<1, 1, 2, 2>(c, a, G);                                         // 32 x 32
}

