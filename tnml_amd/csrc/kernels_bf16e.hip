// kernels_bf16e.hip -- the bf16 study modes (TNML_BF16 / TNML_BF16X3; BASELINE config 5's "bf16 MFMA bond contraction") with their operands
// CONVERTED ONCE (round 6).  Through round 5 both bf16 GEMMs rounded fp32 -> bf16 while staging, on every launch and once per column block
// of the grid: conversion kernels with MFMAs attached (k_fgemm_bf16: 66 us for the 5.5 GF of an m = 300 forward pass on a 2.5 PF pipe).
//
// The forward pass B*t.v (fixedL.cc:318,377,399,416), Label index on an environment:
//     T[n][q] = sum_t phiO[t][n] sum_s phiI[s][n] ( E[n][:] . M[2a + s][2q + t] )
// -- the even / odd identity of kernels_res.hip: the RAW environment is the MFMA operand and both site features are applied to the fp32
// accumulators, so nothing per image has to be multiplied before it is rounded.
//   k_env_bf16t   once per bond (the Label-free input environment changes only when a shift writes it): E fp32 [mI][NTp] -> EbT[n][KH]
//                 bf16, reduction index fastest (KH = mI rounded up to 32, zeros beyond), hi and -- TNML_BF16X3 -- lo planes
//   k_m_bf16t     once per launch (the bond vector changes every launch; replaces the fp64 -> fp32 copy of the old path):
//                 M fp64 M-layout -> MbT[s][t][q][KH] bf16
//   k_fgemm_bf16e 128 images x 64 links per workgroup, 4 waves; per 32-index chunk of the reduction 24 fragment blocks of 1 KB (8 row tiles
//                 of the images, 16 (s, t, q-tile) column tiles) go global -> LDS by LDS-DMA in FRAGMENT ORDER (lane L's 16 bytes at
//                 block + 16 L: the lane that reads a fragment reads what its twin lane wrote, no padding, no conflicts, no VALU),
//                 double buffered, one barrier per chunk; 32 MFMAs (v_mfma_f32_16x16x32_bf16) per wave and chunk, 96 with hi + lo operands
//                 (lo hi + hi lo + hi hi: ~16 mantissa bits); both features in the epilogue.
// Rounding: bf16(E) * phi in fp32 instead of bf16(E * phi) -- the same 8 (16) mantissa bits on the same operand.
#include "tnml_internal.h"

typedef short bf16x8e __attribute__((ext_vector_type(8)));
typedef float f32x4e __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ unsigned short e_f2bf(float x) {
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static __device__ __forceinline__ float e_bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// E fp32 [mI][NTp] -> out[plane][n][KH]; grid (NTp / 64, KH / 32), 256 lanes
__global__ __launch_bounds__(256) void k_env_bf16t(const float* __restrict__ E, int mI, int NTp, int KH, unsigned short* __restrict__ out, int split) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x, n0 = blockIdx.x * 64, a0 = blockIdx.y * 32;
    for (int idx = tid; idx < 32 * 64; idx += 256) {
        const int a = idx >> 6, n = idx & 63;
        tile[a][n] = (a0 + a < mI) ? E[(size_t)(a0 + a) * NTp + n0 + n] : 0.f;
    }
    __syncthreads();
    const int n = tid >> 2, g = tid & 3;
    unsigned short h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = tile[8 * g + i][n];
        h[i] = e_f2bf(v);
        l[i] = e_f2bf(v - e_bf2f(h[i]));
    }
    unsigned short* dst = out + (size_t)(n0 + n) * KH + a0 + 8 * g;
    *reinterpret_cast<uint4*>(dst) = make_uint4((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16));
    if (split)
        *reinterpret_cast<uint4*>(dst + (size_t)NTp * KH) = make_uint4((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16), (unsigned)l[4] | ((unsigned)l[5] << 16), (unsigned)l[6] | ((unsigned)l[7] << 16));
}

// M fp64 M-layout [Kp][Np] (row 2a + s, column 2q + t) -> out[plane][s][t][q][KH]; one lane per (s, t, q, 8 consecutive a), q fastest
__global__ __launch_bounds__(256) void k_m_bf16t(const double* __restrict__ M, int Kp, int Np, int mO, int QP, int KH, unsigned short* __restrict__ out, int split) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int ng = KH >> 3;
    if (idx >= 4 * QP * ng) return;
    const int q = idx % QP, rest = idx / QP;
    const int a8 = rest % ng, st = rest / ng;
    const int s = st >> 1, t = st & 1;
    unsigned short h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 2 * (8 * a8 + i) + s, col = 2 * q + t;
        const float v = (q < mO && row < Kp && col < Np) ? (float)M[(size_t)row * Np + col] : 0.f;
        h[i] = e_f2bf(v);
        l[i] = e_f2bf(v - e_bf2f(h[i]));
    }
    unsigned short* dst = out + ((size_t)st * QP + q) * KH + 8 * a8;
    *reinterpret_cast<uint4*>(dst) = make_uint4((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16));
    if (split)
        *reinterpret_cast<uint4*>(dst + (size_t)4 * QP * KH) = make_uint4((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16), (unsigned)l[4] | ((unsigned)l[5] << 16), (unsigned)l[6] | ((unsigned)l[7] << 16));
}

struct FgemmBf16eArgs {
    const unsigned short* EbT; const unsigned short* MbT;     // [planes][NTp][KH], [planes][2][2][QP][KH]
    const float* phiI; const float* phiO;                     // [2][NTp] each
    float* out;                                               // U[mO][NTp]
    int mO, NTp, KH, QP;
};

static __device__ __forceinline__ void e_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int SPLIT>
__global__ __launch_bounds__(256, 2) void k_fgemm_bf16e(FgemmBf16eArgs A) {      // (two workgroups per CU: 300 workgroups of an m = 300 shard are resident at once)
    constexpr int NPL = SPLIT ? 2 : 1;                  // operand planes (hi, lo)
    constexpr int NBLK = 24 * NPL;                      // 1 KB fragment blocks per chunk: plane x (8 image tiles, then 16 (s, t, q-tile) column tiles)
    extern __shared__ __attribute__((aligned(16))) unsigned char e_lds[];      // [2][NBLK][1024]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int n0 = blockIdx.x * 128, q0 = blockIdx.y * 64;
    const int KH = A.KH, QP = A.QP, NTp = A.NTp;
    const int li = lane & 15, g = lane >> 4;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)e_lds));
    // this lane's 16 bytes of a fragment block: row / column li, reduction indices 8 g .. 8 g + 7 of the chunk
    const unsigned short* const srcA = A.EbT + (size_t)(n0 + li) * KH + 8 * g;          // + 16 tile KH (image tile), + plane NTp KH
    const unsigned short* const srcB = A.MbT + (size_t)(q0 + li) * KH + 8 * g;          // + (st QP + 16 qt) KH, + plane 4 QP KH
    auto stage = [&](int k0, int buf) {
#pragma unroll
        for (int r = 0; r < NBLK / 4; ++r) {
            const int B = w + 4 * r;                        // uniform
            const int pl = B / 24, b = B - 24 * pl;
            const unsigned short* src = b < 8 ? srcA + (size_t)pl * NTp * KH + (size_t)(16 * b) * KH + k0
                                              : srcB + (size_t)pl * 4 * QP * KH + ((size_t)((b - 8) >> 2) * QP + 16 * ((b - 8) & 3)) * KH + k0;
            const unsigned dst = lds0 + (unsigned)((buf * NBLK + B) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    f32x4e acc[4][2][2][2];                             // [image tile r][q tile c][s][t]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[r][c][s][t] = f32x4e{0.f, 0.f, 0.f, 0.f};
    const int nchunk = KH >> 5;
    stage(0, 0);
    e_barrier();
    for (int k = 0; k < nchunk; ++k) {
        const int cur = k & 1;
        if (k + 1 < nchunk) stage(32 * (k + 1), cur ^ 1);
        const unsigned char* base = e_lds + (size_t)cur * NBLK * 1024 + lane * 16;
        bf16x8e ah[4], al[SPLIT ? 4 : 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ah[r] = *reinterpret_cast<const bf16x8e*>(base + (4 * wr + r) * 1024);
            if (SPLIT) al[r] = *reinterpret_cast<const bf16x8e*>(base + (24 + 4 * wr + r) * 1024);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int blk = 8 + (2 * s + t) * 4 + 2 * wc + c;
                    const bf16x8e bh = *reinterpret_cast<const bf16x8e*>(base + blk * 1024);
                    bf16x8e bl = bh;
                    if (SPLIT) bl = *reinterpret_cast<const bf16x8e*>(base + (24 + blk) * 1024);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (SPLIT) {                                   // small terms first
                            acc[r][c][s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[r], bh, acc[r][c][s][t], 0, 0, 0);
                            acc[r][c][s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bl, acc[r][c][s][t], 0, 0, 0);
                        }
                        acc[r][c][s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bh, acc[r][c][s][t], 0, 0, 0);
                    }
                }
        e_barrier();                                    // the next chunk has landed; every wave is done with this one
    }
    // epilogue: the C fragment is 4 consecutive images (rows 4 g + reg) x 1 link (column li) per lane; both features in fp32
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + (4 * wr + r) * 16 + 4 * g;
        const float4 pI0 = *reinterpret_cast<const float4*>(A.phiI + n), pI1 = *reinterpret_cast<const float4*>(A.phiI + NTp + n);
        const float4 pO0 = *reinterpret_cast<const float4*>(A.phiO + n), pO1 = *reinterpret_cast<const float4*>(A.phiO + NTp + n);
        const float i0[4] = {pI0.x, pI0.y, pI0.z, pI0.w}, i1[4] = {pI1.x, pI1.y, pI1.z, pI1.w};
        const float o0[4] = {pO0.x, pO0.y, pO0.z, pO0.w}, o1[4] = {pO1.x, pO1.y, pO1.z, pO1.w};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int q = q0 + 16 * (2 * wc + c) + li;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u0 = i0[e] * acc[r][c][0][0][e] + i1[e] * acc[r][c][1][0][e];      // t = 0
                const float u1 = i0[e] * acc[r][c][0][1][e] + i1[e] * acc[r][c][1][1][e];      // t = 1
                v[e] = o0[e] * u0 + o1[e] * u1;
            }
            if (q < A.mO) *reinterpret_cast<float4*>(A.out + (size_t)q * NTp + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// workspace (elements of 2 bytes) for a context of bond dimension maxm on NTp images
size_t bf16e_env_elems(int maxm, int NTp, int split) { return (size_t)(split ? 2 : 1) * NTp * ((maxm + 31) / 32 * 32); }
size_t bf16e_m_elems(int maxm, int split) { return (size_t)(split ? 2 : 1) * 4 * ((maxm + 63) / 64 * 64) * ((maxm + 31) / 32 * 32); }

// the forward feature GEMM of the bf16 modes with operands converted once: U[q][n] (fp32) from the Label-free environment EI (fp32 [mI][NTp]) and
// the bond vector vec (fp64 M-layout [Kp][Np]).  The environment's bf16 copy is reused while (EI, mI, env_epoch) stand.
int launch_fgemm_bf16e(tnml_ctx* c, const float* EI, int mI, const float* phiI, const double* vec, int Kp, int Np, const float* phiO, float* out, int mO) {
    const int split = c->bf16() == 2, NTp = c->NTp;
    const int KH = (mI + 31) / 32 * 32, QP = (mO + 63) / 64 * 64;
    if (NTp % 128) return tnml_fail(c, "fgemm_bf16e: image count not a multiple of 128");
    if (!c->ebt || !c->mbt || bf16e_env_elems(mI, NTp, split) > c->ebt_cap || (size_t)(split ? 2 : 1) * 4 * QP * KH > c->mbt_cap)
        return tnml_fail(c, "fgemm_bf16e: workspace too small for a %d x %d bond", mI, mO);
    if (c->ebt_src != EI || c->ebt_m != mI || c->ebt_epoch != c->env_epoch) {
        ProfScope ps(c, KC_PACK);
        hipLaunchKernelGGL(k_env_bf16t, dim3(NTp / 64, KH / 32), dim3(256), 0, c->stream, EI, mI, NTp, KH, c->ebt, split);
        c->ebt_src = EI; c->ebt_m = mI; c->ebt_epoch = c->env_epoch;
    }
    {
        ProfScope ps(c, KC_PACK);
        const int nthreads = 4 * QP * (KH / 8);
        hipLaunchKernelGGL(k_m_bf16t, dim3((nthreads + 255) / 256), dim3(256), 0, c->stream, vec, Kp, Np, mO, QP, KH, c->mbt, split);
    }
    FgemmBf16eArgs a{c->ebt, c->mbt, phiI, phiO, out, mO, NTp, KH, QP};
    const size_t lds = (size_t)2 * 24 * (split ? 2 : 1) * 1024;
    if (!c->attr_bf16e) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fgemm_bf16e<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 24 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_fgemm_bf16e<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 48 * 1024) != hipSuccess)
            return tnml_fail(c, "fgemm_bf16e: cannot reserve %zu bytes of LDS", lds);
        c->attr_bf16e = true;
    }
    {
        ProfScope ps(c, KC_FGEMM_FWD);
        const dim3 grid(NTp / 128, QP / 64);
        if (split) hipLaunchKernelGGL(k_fgemm_bf16e<1>, grid, dim3(256), lds, c->stream, a);
        else       hipLaunchKernelGGL(k_fgemm_bf16e<0>, grid, dim3(256), lds, c->stream, a);
    }
    HIPCK(c, hipGetLastError());
    return 0;
}
