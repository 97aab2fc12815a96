// Follow-up to probe_valu_mfma.hip: does a VALU wave get issue slots when its SIMD hosts ONE MFMA wave (4 chains) instead of two?
// and what does the MFMA wave pay?  768 lanes: waves 0..NMM-1 run fp64 MFMAs back to back (CH independent accumulators), waves NMM..7
// exit, waves 8..11 run groups of 10 fp64 FMAs.  Reports ticks per group of 10 FMAs and ticks per MFMA of wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NMM, int CH, int VALU_ON, int NV = 4>
__global__ __launch_bounds__(1024) void k(long long* out, double* sink, int iters) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wid < 8) {
        if (wid >= NMM) return;
        f64x4 a[CH];
        for (int c = 0; c < CH; ++c) a[c] = f64x4{0, 0, 0, 0};
        double x = lane * 1e-3, y = 1.0 + lane * 1e-4;
        const long long t0 = clock64();
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
                for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[c], 0, 0, 0);
        }
        const long long t1 = clock64();
        double s = 0; for (int c = 0; c < CH; ++c) s += a[c][0];
        if (s == 1.2345) sink[0] = s;
        if (lane == 0 && blockIdx.x == 0 && wid == 0) out[4] = t1 - t0;
    } else {
        if (!VALU_ON || wid >= 8 + NV) return;
        double d[10];
        for (int i = 0; i < 10; ++i) d[i] = lane + i;
        const double dm = 1.0000001;
        const long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 10; ++i) d[i] = fma(d[i], dm, 1e-9);
            asm volatile("" : "+v"(d[0]));
        }
        const long long t1 = clock64();
        double s = 0; for (int i = 0; i < 10; ++i) s += d[i];
        if (s == 1.2345) sink[1] = s;
        if (lane == 0 && blockIdx.x == 0 && wid < 12) out[wid - 8] = t1 - t0;
    }
}
template <int NMM, int CH, int VALU_ON, int NV = 4> static void run(const char* tag, long long* d, double* sink) {
    const int iters = 2000;
    hipMemset(d, 0, 64);
    hipLaunchKernelGGL((k<NMM, CH, VALU_ON, NV>), dim3(256), dim3(NV > 4 ? 1024 : 768), 0, 0, d, sink, iters); hipDeviceSynchronize();
    long long h[5]; hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("%-78s %8.1f ticks per 10 FMAs | %6.2f ticks per MFMA of wave 0\n", tag, (double)h[0] / iters, (double)h[4] / (iters * 64.0));
}
int main() {
    long long* d; double* sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
    run<8, 2, 0>("2 MFMA waves per SIMD (2 chains each), no VALU waves", d, sink);
    run<8, 2, 1>("2 MFMA waves per SIMD (2 chains each) + VALU waves", d, sink);
    run<4, 4, 0>("1 MFMA wave per SIMD (4 chains), no VALU waves", d, sink);
    run<4, 4, 1>("1 MFMA wave per SIMD (4 chains) + VALU waves", d, sink);
    run<4, 2, 1>("1 MFMA wave per SIMD (2 chains) + VALU waves", d, sink);
    run<4, 1, 1>("1 MFMA wave per SIMD (1 chain: dependent MFMAs) + VALU waves", d, sink);
    run<8, 1, 1>("2 MFMA waves per SIMD (1 chain each) + VALU waves", d, sink);
    run<0, 1, 1>("no MFMA waves, VALU waves alone", d, sink);
    run<8, 2, 1, 8>("2 MFMA waves per SIMD + TWO VALU waves per SIMD (1024 lanes)", d, sink);
    run<4, 4, 1, 8>("1 MFMA wave per SIMD + TWO VALU waves per SIMD", d, sink);
    return 0;
}
