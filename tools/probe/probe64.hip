// f64 MFMA probe: fragment map check of v_mfma_f64_16x16x4_f64 and its issue-rate ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_layout(const double* A, const double* B, double* C) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];      // A[i][k] 16x4
    double b = B[(l >> 4) * 16 + (l & 15)];     // B[k][j] 4x16
    f64x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
template <int NACC>
__global__ void k_rate64(double* out, int iters) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1.2345) out[0] = s;
}
template <int NACC>
__global__ void k_rate32(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1.2345f) out[0] = s;
}
int main() {
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = ((i * 7 + 3) % 11) - 5.0; B[i] = ((i * 5 + 1) % 13) - 6.0; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(C[i] - R[i]));
    printf("mfma_f64_16x16x4f64 layout (row = (lane>>4)+4*reg): max err %g (%s)\n", e, e == 0 ? "OK" : "MISMATCH");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int waves : {4, 8, 16}) {
        k_rate64<8><<<256 * 4, 64 * (waves / 4)>>>(dC, 100); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_rate64<8><<<256 * 4, 64 * (waves / 4)>>>(dC, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double fl = 256.0 * 4 * (waves / 4) * iters * 8 * 2048.0;
        printf("f64 mfma 16x16x4: %d waves/CU: %.1f TF\n", waves, fl / ms / 1e9);
        CK(hipEventRecord(e0)); k_rate32<8><<<256 * 4, 64 * (waves / 4)>>>((float*)dC, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("f32 mfma 16x16x4: %d waves/CU: %.1f TF\n", waves, fl / ms / 1e9);
    }
    return 0;
}
