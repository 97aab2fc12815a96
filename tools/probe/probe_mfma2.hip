// fp64 MFMA issue-rate probe in the shape of k_fwd_res: 8 waves per CU (2 per SIMD), NACC accumulator chains per wave, the A operand
// from a register array (distinct values), the B operand constant or read from LDS (one ds_read_b64 per NACC MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC, int LDSR, int NREG>
__global__ __launch_bounds__(512) void k(double* out, const double* in, int iters) {
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = in[i & 255];
    __syncthreads();
    double m[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) m[r] = in[r * 64 + (threadIdx.x & 63)];
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    const double* lp = lds + (threadIdx.x & 63);
    double b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            double bb = b;
            if (LDSR) bb = lp[(r * 64 + it * 7) & 4095];
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(m[r], LDSR ? bb : b + i, acc[i], 0, 0, 0);
        }
    }
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1.2345) out[0] = s;
}
template <int NACC, int LDSR, int NREG>
static void run(const char* tag, double* dC, const double* dI) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    hipLaunchKernelGGL((k<NACC, LDSR, NREG>), dim3(256), dim3(512), 0, 0, dC, dI, 10); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<NACC, LDSR, NREG>), dim3(256), dim3(512), 0, 0, dC, dI, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 256.0 * 8 * iters * NREG * NACC * 2048.0;
    printf("%-60s %6.1f TF (%4.1f %%)\n", tag, fl / ms / 1e9, 100. * fl / ms / 1e9 / 78.6);
}
int main() {
    double *dC, *dI; hipMalloc(&dC, 1 << 20); hipMalloc(&dI, 1 << 20); hipMemset(dI, 0, 1 << 20);
    run<2, 0, 30>("2 chains/wave, B const, 30 A regs", dC, dI);
    run<4, 0, 30>("4 chains/wave, B const, 30 A regs", dC, dI);
    run<8, 0, 30>("8 chains/wave, B const, 30 A regs", dC, dI);
    run<2, 1, 30>("2 chains/wave, B from LDS, 30 A regs", dC, dI);
    run<4, 1, 30>("4 chains/wave, B from LDS, 30 A regs", dC, dI);
    run<8, 1, 30>("8 chains/wave, B from LDS, 30 A regs", dC, dI);
    run<4, 0, 1>("4 chains/wave, B const, 1 A reg", dC, dI);
    run<8, 0, 1>("8 chains/wave, B const, 1 A reg", dC, dI);
    return 0;
}
