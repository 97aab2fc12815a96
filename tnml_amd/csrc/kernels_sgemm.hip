// kernels_sgemm.hip -- the few-hundred-square fp64 products of the bond-tensor split (fixedL.cc:494,519-521,527) on
// v_mfma_f64_16x16x4_f64, one wave per output tile, operands straight from L2.
//
// Why not rocBLAS: at 120..640 squared these products are 2-14 MFLOP -- microseconds of a few CUs -- and a rocBLAS call costs
// 7-17 us whatever it computes (macro tile 64 x 64 or larger: 2-16 workgroups, a prologue sized for big problems;
// profiles/r04_bench_c3_f64_kernel_summary.txt: eight Cijk_* launches per bond update at 1-2 % matrix-pipe busy).  Here a launch is
// bound by its own latency: <= 60 dependent MFMAs per wave at n = 240, every operand element loaded exactly once per tile.
// Deterministic (one wave owns a tile, fixed k order).  The Label-on-B bonds (reduction length 2400) keep the rocBLAS strips.
//
// C (M x N, column-major) = op(A) op(B), op = identity or transpose.  MFMA operand roles are swapped against the usual reading so
// that a lane's four accumulator values are FOUR COLUMNS of one row i and 16 consecutive lanes store 16 consecutive rows (128-byte
// stores into the column-major C): first operand a[j][k] = op(B)[k][j], second b[k][i] = op(A)[i][k], acc[e] = C[i0 + (lane & 15)][j0 + (lane >> 4) + 4 e].
// The four k of one MFMA are k = 16 kb + 4 (lane >> 4) + u, u = the MFMA's number inside the block of 16: a lane's four k are
// consecutive in memory (transposed operands read 32 contiguous bytes per block), and a sum over k does not care about its order.
#include "tnml_internal.h"

typedef double f64x4s __attribute__((ext_vector_type(4)));

// BMODE 1: op(B) = 1.5 I - 0.5 S with S = the symmetric matrix passed as B (the Newton-Schulz factor of the split's polish step,
// formed while loading), and dev[0] = max |S - I| by an integer atomic max over non-negative doubles (dev[0] zeroed by the producer of S)
// KS waves share a tile: wave w takes the blocks of 16 k with index = w (mod KS) and the partial tiles are added in wave order through LDS
// (a latency chain of 60 MFMAs with their loads becomes 15: the K = 240 products of the split, 10.5 -> ~7 us)
template <int TA, int TB, int TI, int TJ, int BMODE, int KS>
__global__ __launch_bounds__(64 * KS) void k_dgemm_small(SmallGemmArgs P) {
    __shared__ double s_acc[KS > 1 ? (KS - 1) * 64 * 4 * TI * TJ : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * 16 * TI, j0 = blockIdx.y * 16 * TJ;
    const int M = P.M, N = P.N, K = P.K;
    f64x4s acc[TJ][TI];
#pragma unroll
    for (int sj = 0; sj < TJ; ++sj)
#pragma unroll
        for (int si = 0; si < TI; ++si) acc[sj][si] = f64x4s{0., 0., 0., 0.};
    // row / column of this lane in every sub-tile, clamped for the loads (values outside the matrix are never stored)
    int ia[TI], jb[TJ];
#pragma unroll
    for (int si = 0; si < TI; ++si) { const int i = i0 + 16 * si + r; ia[si] = i < M ? i : M - 1; }
#pragma unroll
    for (int sj = 0; sj < TJ; ++sj) { const int j = j0 + 16 * sj + r; jb[sj] = j < N ? j : N - 1; }
    double dmax = 0.;
    for (int kb = 16 * wv; kb < K; kb += 16 * KS) {
        double af[TI][4], bf[TJ][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + 4 * g + u;
            const bool ok = k < K;
            const int kc = ok ? k : K - 1;
#pragma unroll
            for (int si = 0; si < TI; ++si) {
                const double v = TA ? P.A[kc + (size_t)P.lda * ia[si]] : P.A[ia[si] + (size_t)P.lda * kc];
                af[si][u] = ok ? v : 0.;
            }
#pragma unroll
            for (int sj = 0; sj < TJ; ++sj) {
                double v = TB ? P.B[jb[sj] + (size_t)P.ldb * kc] : P.B[kc + (size_t)P.ldb * jb[sj]];
                if (BMODE == 1) {
                    const double id = kc == jb[sj] ? 1. : 0.;
                    if (blockIdx.x == 0) dmax = fmax(dmax, ok && j0 + 16 * sj + r < N ? fabs(v - id) : 0.);     // the tiles of the first row block cover S once
                    if (!(fabs(v - id) < 1e300)) dmax = fmax(dmax, 1e300);                                   // NaN / Inf must not hide behind fmax
                    v = 1.5 * id - 0.5 * v;
                }
                bf[sj][u] = ok ? v : 0.;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int sj = 0; sj < TJ; ++sj)
#pragma unroll
                for (int si = 0; si < TI; ++si) acc[sj][si] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[sj][u], af[si][u], acc[sj][si], 0, 0, 0);
    }
    if (KS > 1) {
        if (wv > 0) {
#pragma unroll
            for (int sj = 0; sj < TJ; ++sj)
#pragma unroll
                for (int si = 0; si < TI; ++si)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_acc[(((wv - 1) * TJ + sj) * TI + si) * 256 + e * 64 + lane] = acc[sj][si][e];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int w = 1; w < KS; ++w)
#pragma unroll
                for (int sj = 0; sj < TJ; ++sj)
#pragma unroll
                    for (int si = 0; si < TI; ++si)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[sj][si][e] += s_acc[(((w - 1) * TJ + sj) * TI + si) * 256 + e * 64 + lane];
        }
    }
    if (wv == 0) {
#pragma unroll
    for (int sj = 0; sj < TJ; ++sj)
#pragma unroll
        for (int si = 0; si < TI; ++si) {
            const int i = i0 + 16 * si + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + 16 * sj + g + 4 * e;
                if (i < M && j < N) P.C[i + (size_t)P.ldc * j] = acc[sj][si][e];
            }
        }
    }
    if (P.chk_src && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        // (launched after the polish step: the check values are final) -> pinned host mirror; bad = the test svd_split_device applies
        const double d0 = P.chk_src[0], d1 = P.chk_src[1];
        P.chk_host[0] = d0; P.chk_host[1] = d1; P.chk_host[2] = P.chk_src[2]; P.chk_host[3] = P.chk_src[3];
        const double bad = (!(d0 < 1e-6) || d1 != 0.) ? 1. : 0.;
        P.chk_host[4] = bad;
        if (P.chk_bad) P.chk_bad[0] = bad;
    }
    if (BMODE == 1 && blockIdx.x == 0) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(P.dev), (unsigned long long)__double_as_longlong(dmax));      // (every wave of the tile its own share of k)
    }
}

template <int TA, int TB, int BMODE>
static void dgemm_small_go(hipStream_t st, const SmallGemmArgs& a) {
    // 16 x 16 tiles while they fit one wave per CU (latency: a quarter of the MFMA chain of a 32 x 32 tile), 32 x 32 beyond; four waves
    // per tile from a reduction length of 64 on (>= one block of 16 per wave) while the tiles are few
    const int t16 = ((a.M + 15) / 16) * ((a.N + 15) / 16);
    if (t16 <= 256 && a.K >= 64)  hipLaunchKernelGGL((k_dgemm_small<TA, TB, 1, 1, BMODE, 4>), dim3((a.M + 15) / 16, (a.N + 15) / 16), dim3(256), 0, st, a);
    else if (t16 <= 512)          hipLaunchKernelGGL((k_dgemm_small<TA, TB, 1, 1, BMODE, 1>), dim3((a.M + 15) / 16, (a.N + 15) / 16), dim3(64), 0, st, a);
    else                          hipLaunchKernelGGL((k_dgemm_small<TA, TB, 2, 2, BMODE, 1>), dim3((a.M + 31) / 32, (a.N + 31) / 32), dim3(64), 0, st, a);
}
int launch_dgemm_small(tnml_ctx* c, const SmallGemmArgs& a) {
    if (a.M < 1 || a.N < 1 || a.K < 1) return tnml_fail(c, "dgemm_small: empty product");
    hipStream_t st = c->stream;
    if (a.bmode == 1) {
        if (a.ta || a.tb || a.K != a.N) return tnml_fail(c, "dgemm_small: the Newton-Schulz form is A (1.5 I - 0.5 S) with S square");
        dgemm_small_go<0, 0, 1>(st, a);
    } else if (!a.ta && !a.tb) dgemm_small_go<0, 0, 0>(st, a);
    else if (a.ta && !a.tb)    dgemm_small_go<1, 0, 0>(st, a);
    else if (!a.ta && a.tb)    dgemm_small_go<0, 1, 0>(st, a);
    else                       dgemm_small_go<1, 1, 0>(st, a);
    HIPCK(c, hipGetLastError());
    return 0;
}

// the side job alone (the product it would have ridden in went to rocBLAS)
__global__ void k_split_check_mirror(const double* __restrict__ src, double* __restrict__ host, double* __restrict__ bad) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double d0 = src[0], d1 = src[1];
        host[0] = d0; host[1] = d1; host[2] = src[2]; host[3] = src[3];
        const double b = (!(d0 < 1e-6) || d1 != 0.) ? 1. : 0.;
        host[4] = b;
        if (bad) bad[0] = b;
    }
}
int launch_split_check_mirror(tnml_ctx* c, const double* src, double* host, double* bad) {
    hipLaunchKernelGGL(k_split_check_mirror, dim3(1), dim3(64), 0, c->stream, src, host, bad);
    HIPCK(c, hipGetLastError());
    return 0;
}
