// How long does a VALU instruction of a third wave take while the two other waves of its SIMD issue fp64 MFMAs back to back?
// 768 lanes: waves 0..7 run MFMAs (MM = 1) or idle (MM = 0); waves 8..11 run groups of 10 independent VALU operations (fp64 FMA,
// fp32 FMA or integer multiply-add) and report cycles per group.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MM, int KIND, int PRIO, int FIRST = 0>
__global__ __launch_bounds__(768) void k(long long* out, double* sink, int iters) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mm_role = FIRST ? wid >= 4 : wid < 8;
    if (mm_role) {
        if (!MM) return;
        f64x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        double x = lane * 1e-3, y = 1.0 + lane * 1e-4;
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0); }
        }
        if (a0[0] + a1[3] == 1.2345) sink[0] = a0[0];
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        double d[10]; float f[10]; int n[10];
        for (int i = 0; i < 10; ++i) { d[i] = lane + i; f[i] = lane + i; n[i] = lane + i; }
        const double dm = 1.0000001; const float fm = 1.0000001f;
        const long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                if (KIND == 0) d[i] = fma(d[i], dm, 1e-9);
                if (KIND == 1) f[i] = fmaf(f[i], fm, 1e-9f);
                if (KIND == 2) n[i] = n[i] * 3 + 1;
            }
            asm volatile("" : "+v"(d[0]), "+v"(f[0]), "+v"(n[0]));
        }
        const long long t1 = clock64();
        double s = 0; for (int i = 0; i < 10; ++i) s += d[i] + f[i] + n[i];
        if (s == 1.2345) sink[1] = s;
        if (lane == 0 && blockIdx.x == 0) out[FIRST ? wid : wid - 8] = t1 - t0;
    }
}
template <int MM, int KIND, int PRIO, int FIRST = 0> static void run(const char* tag, long long* d, double* sink) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MM, KIND, PRIO, FIRST>), dim3(256), dim3(768), 0, 0, d, sink, iters); hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%-64s %7.1f ticks per group of 10 (waves: %lld %lld %lld %lld)\n", tag, (double)h[0] / iters, h[0], h[1], h[2], h[3]);
}
int main() {
    long long* d; double* sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
    run<0, 0, 0>("10 fp64 FMAs, no MFMA waves", d, sink);
    run<1, 0, 0>("10 fp64 FMAs beside 2 MFMA waves per SIMD", d, sink);
    run<1, 0, 1>("10 fp64 FMAs beside 2 MFMA waves per SIMD, s_setprio 3", d, sink);
    run<0, 1, 0>("10 fp32 FMAs, no MFMA waves", d, sink);
    run<1, 1, 0>("10 fp32 FMAs beside 2 MFMA waves per SIMD", d, sink);
    run<1, 1, 1>("10 fp32 FMAs beside 2 MFMA waves per SIMD, s_setprio 3", d, sink);
    run<0, 2, 0>("10 integer mads, no MFMA waves", d, sink);
    run<1, 2, 0>("10 integer mads beside 2 MFMA waves per SIMD", d, sink);
    run<1, 2, 1>("10 integer mads beside 2 MFMA waves per SIMD, s_setprio 3", d, sink);
    run<1, 0, 0, 1>("10 fp64 FMAs in waves 0..3, MFMA waves 4..11", d, sink);
    run<1, 0, 1, 1>("10 fp64 FMAs in waves 0..3 (s_setprio 3), MFMA waves 4..11", d, sink);
    run<1, 1, 0, 1>("10 fp32 FMAs in waves 0..3, MFMA waves 4..11", d, sink);
    return 0;
}
