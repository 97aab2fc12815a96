// Round-1 hardware probe (not product code): verifies the f32 MFMA fragment maps this
// repo's kernels rely on, times rocSOLVER SVD/eig candidates at the bond-tensor shapes,
// and measures streaming-read bandwidth for 4/8/16-byte-per-lane loads.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_mfma16(const float* A, const float* B, float* C) {
    // A: 16x4 row-major, B: 4x16 row-major, C: 16x16 row-major
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)];
    float b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
__global__ void k_mfma32(const float* A, const float* B, float* C) {
    // A: 32x2 row-major, B: 2x32 row-major, C: 32x32 row-major
    int l = threadIdx.x;
    float a = A[(l & 31) * 2 + (l >> 5)];
    float b = B[(l >> 5) * 32 + (l & 31)];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        C[row * 32 + (l & 31)] = acc[r];
    }
}

template <typename T>
__global__ void k_stream(const T* __restrict__ p, size_t n, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0;
    for (; i < n; i += stride) {
        T v = p[i];
        const float* f = reinterpret_cast<const float*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += f[k];
    }
    if (acc == 123.456f) out[0] = acc;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename F>
static double time_ms(F&& f, int reps) {
    f(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int i = 0; i < reps; ++i) {
        double t0 = now_ms(); f(); CK(hipDeviceSynchronize()); double t1 = now_ms();
        if (t1 - t0 < best) best = t1 - t0;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s arch=%s CUs=%d LDS/blk=%zu mem=%.1f GB clock=%d MHz\n", prop.name, prop.gcnArchName,
           prop.multiProcessorCount, prop.sharedMemPerBlock, prop.totalGlobalMem / 1e9, prop.clockRate / 1000);
    // ---- MFMA layout checks
    {
        std::vector<float> A(64), B(64), C(256), R(256, 0.f);
        for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 7 + 3) % 11) - 5.f; B[i] = (float)((i * 5 + 1) % 13) - 6.f; }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
        float *dA, *dB, *dC; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dC, 1024));
        CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
        k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(C[i] - R[i]));
        printf("mfma_f32_16x16x4f32 layout check: max err %g (%s)\n", e, e == 0 ? "OK" : "MISMATCH");
    }
    {
        std::vector<float> A(64), B(64), C(1024), R(1024, 0.f);
        for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 7 + 3) % 11) - 5.f; B[i] = (float)((i * 5 + 1) % 13) - 6.f; }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 2; ++k) R[i * 32 + j] += A[i * 2 + k] * B[k * 32 + j];
        float *dA, *dB, *dC; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
        k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - R[i]));
        printf("mfma_f32_32x32x2f32 layout check: max err %g (%s)\n", e, e == 0 ? "OK" : "MISMATCH");
    }
    // ---- streaming read bandwidth
    {
        size_t bytes = (size_t)2 << 30; float* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
        float* o; CK(hipMalloc(&o, 4));
        for (int blocks : {1024, 2048, 4096, 8192}) {
            double t4 = time_ms([&] { k_stream<float><<<blocks, 256>>>(d, bytes / 4, o); }, 3);
            double t8 = time_ms([&] { k_stream<float2><<<blocks, 256>>>((float2*)d, bytes / 8, o); }, 3);
            double t16 = time_ms([&] { k_stream<float4><<<blocks, 256>>>((float4*)d, bytes / 16, o); }, 3);
            printf("stream 2GiB blocks=%d: 4B %.0f GB/s  8B %.0f GB/s  16B %.0f GB/s\n", blocks, bytes / t4 / 1e6, bytes / t8 / 1e6, bytes / t16 / 1e6);
        }
        CK(hipFree(d));
    }
    // ---- launch latency
    {
        float* o; CK(hipMalloc(&o, 4));
        double t0 = now_ms();
        for (int i = 0; i < 1000; ++i) k_stream<float><<<1, 64>>>(o, 0, o);
        CK(hipDeviceSynchronize());
        printf("1000 empty launches: %.3f ms (%.2f us each)\n", now_ms() - t0, (now_ms() - t0));
        double s0 = now_ms();
        for (int i = 0; i < 200; ++i) { k_stream<float><<<1, 64>>>(o, 0, o); CK(hipDeviceSynchronize()); }
        printf("launch+sync: %.2f us each\n", (now_ms() - s0) * 1000 / 200);
    }
    // ---- rocSOLVER timings
    rocblas_handle h; rocblas_create_handle(&h);
    auto run_svd = [&](int m, int n) {
        int k = m < n ? m : n;
        std::vector<double> A((size_t)m * n); srand(1);
        for (auto& x : A) x = rand() / (double)RAND_MAX - 0.5;
        std::vector<float> Af(A.begin(), A.end());
        double *dA, *dS, *dU, *dV, *dE, *dres; int *dinfo, *dns;
        CK(hipMalloc(&dA, sizeof(double) * m * n)); CK(hipMalloc(&dS, sizeof(double) * k)); CK(hipMalloc(&dU, sizeof(double) * m * k));
        CK(hipMalloc(&dV, sizeof(double) * k * n)); CK(hipMalloc(&dE, sizeof(double) * k)); CK(hipMalloc(&dinfo, 4)); CK(hipMalloc(&dns, 4)); CK(hipMalloc(&dres, 8));
        float *fA = (float*)dA, *fS = (float*)dS, *fU = (float*)dU, *fV = (float*)dV, *fE = (float*)dE, *fres = (float*)dres;
        double t;
        t = time_ms([&] { CK(hipMemcpy(dA, A.data(), sizeof(double) * m * n, hipMemcpyHostToDevice));
            rocsolver_dgesvd(h, rocblas_svect_singular, rocblas_svect_singular, m, n, dA, m, dS, dU, m, dV, k, dE, rocblas_outofplace, dinfo); }, 3);
        printf("dgesvd  %dx%d: %.3f ms\n", m, n, t);
        t = time_ms([&] { CK(hipMemcpy(fA, Af.data(), sizeof(float) * m * n, hipMemcpyHostToDevice));
            rocsolver_sgesvd(h, rocblas_svect_singular, rocblas_svect_singular, m, n, fA, m, fS, fU, m, fV, k, fE, rocblas_outofplace, dinfo); }, 3);
        printf("sgesvd  %dx%d: %.3f ms\n", m, n, t);
        t = time_ms([&] { CK(hipMemcpy(dA, A.data(), sizeof(double) * m * n, hipMemcpyHostToDevice));
            rocsolver_dgesvdj(h, rocblas_svect_singular, rocblas_svect_singular, m, n, dA, m, 0.0, dres, 30, dns, dS, dU, m, dV, k, dinfo); }, 3);
        printf("dgesvdj %dx%d: %.3f ms\n", m, n, t);
        t = time_ms([&] { CK(hipMemcpy(fA, Af.data(), sizeof(float) * m * n, hipMemcpyHostToDevice));
            rocsolver_sgesvdj(h, rocblas_svect_singular, rocblas_svect_singular, m, n, fA, m, 0.f, fres, 30, dns, fS, fU, m, fV, k, dinfo); }, 3);
        printf("sgesvdj %dx%d: %.3f ms\n", m, n, t);
        hipFree(dA); hipFree(dS); hipFree(dU); hipFree(dV); hipFree(dE); hipFree(dinfo); hipFree(dns); hipFree(dres);
    };
    auto run_eig = [&](int n) {
        std::vector<double> A((size_t)n * n), G((size_t)n * n, 0.0); srand(2);
        for (auto& x : A) x = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += A[i + (size_t)k * n] * A[j + (size_t)k * n]; G[i + (size_t)j * n] = s; }
        std::vector<float> Gf(G.begin(), G.end());
        double *dA, *dD, *dE, *dres, *dZ; int *dinfo, *dns, *dnev, *dfail;
        CK(hipMalloc(&dA, 8 * n * n)); CK(hipMalloc(&dZ, 8 * n * n)); CK(hipMalloc(&dD, 8 * n)); CK(hipMalloc(&dE, 8 * n)); CK(hipMalloc(&dres, 8));
        CK(hipMalloc(&dinfo, 4)); CK(hipMalloc(&dns, 4)); CK(hipMalloc(&dnev, 4)); CK(hipMalloc(&dfail, 4 * n));
        float *fA = (float*)dA, *fD = (float*)dD, *fE = (float*)dE, *fres = (float*)dres;
        double t;
        t = time_ms([&] { CK(hipMemcpy(dA, G.data(), 8 * n * n, hipMemcpyHostToDevice)); rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dD, dE, dinfo); }, 3);
        printf("dsyevd  %d: %.3f ms\n", n, t);
        t = time_ms([&] { CK(hipMemcpy(fA, Gf.data(), 4 * n * n, hipMemcpyHostToDevice)); rocsolver_ssyevd(h, rocblas_evect_original, rocblas_fill_upper, n, fA, n, fD, fE, dinfo); }, 3);
        printf("ssyevd  %d: %.3f ms\n", n, t);
        t = time_ms([&] { CK(hipMemcpy(dA, G.data(), 8 * n * n, hipMemcpyHostToDevice)); rocsolver_dsyev(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dD, dE, dinfo); }, 3);
        printf("dsyev   %d: %.3f ms\n", n, t);
        t = time_ms([&] { CK(hipMemcpy(dA, G.data(), 8 * n * n, hipMemcpyHostToDevice)); rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, n, dA, n, 0.0, dres, 30, dns, dD, dinfo); }, 3);
        printf("dsyevj  %d: %.3f ms\n", n, t);
        t = time_ms([&] { CK(hipMemcpy(fA, Gf.data(), 4 * n * n, hipMemcpyHostToDevice)); rocsolver_ssyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, n, fA, n, 0.f, fres, 30, dns, fD, dinfo); }, 3);
        printf("ssyevj  %d: %.3f ms\n", n, t);
        t = time_ms([&] { CK(hipMemcpy(dA, G.data(), 8 * n * n, hipMemcpyHostToDevice)); rocsolver_dsyevdx(h, rocblas_evect_original, rocblas_erange_index, rocblas_fill_upper, n, dA, n, 0, 0, n / 2 + 1, n, dnev, dD, dZ, n, dinfo); }, 3);
        printf("dsyevdx %d (top half): %.3f ms\n", n, t);
        hipFree(dA); hipFree(dZ); hipFree(dD); hipFree(dE); hipFree(dres); hipFree(dinfo); hipFree(dns); hipFree(dnev); hipFree(dfail);
    };
    for (int n : {40, 240, 600}) run_eig(n);
    run_svd(40, 40); run_svd(240, 240); run_svd(2400, 240); run_svd(600, 600);
    // ---- host
    printf("host: "); fflush(stdout); (void)!system("nproc; grep -m1 'model name' /proc/cpuinfo; free -g | head -2");
    return 0;
}
