// latency of the stock orthonormalisation building blocks at the sizes of the split (m = 120 kept vectors of length 240):
// rocsolver_dpotrf, rocblas_dtrsm (right, lower, transposed), rocblas_dtrtri, and dgemm for scale
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int n = 240;
    rocblas_handle h; rocblas_create_handle(&h);
    hipStream_t st; HC(hipStreamCreate(&st)); rocblas_set_stream(h, st);
    for (int m : {120, 64, 32}) {
        std::vector<double> Q((size_t)n * m), S((size_t)m * m);
        srand(3);
        for (auto& v : Q) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += Q[k + (size_t)n * i] * Q[k + (size_t)n * j]; S[i + (size_t)m * j] = s; }
        double *dQ, *dS, *dS0, *dQ1, *dInv; int* info;
        HC(hipMalloc(&dQ, 8 * n * m)); HC(hipMalloc(&dQ1, 8 * n * m)); HC(hipMalloc(&dS, 8 * m * m)); HC(hipMalloc(&dS0, 8 * m * m)); HC(hipMalloc(&dInv, 8 * m * m)); HC(hipMalloc(&info, 4));
        HC(hipMemcpy(dQ, Q.data(), 8 * n * m, hipMemcpyHostToDevice)); HC(hipMemcpy(dS0, S.data(), 8 * m * m, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        const double one = 1.0, zero = 0.0;
        auto timeit = [&](const char* name, auto fn) {
            for (int w = 0; w < 3; ++w) fn();
            HC(hipStreamSynchronize(st));
            HC(hipEventRecord(e0, st));
            for (int r = 0; r < 20; ++r) fn();
            HC(hipEventRecord(e1, st)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1));
            printf("m=%3d %-44s %8.1f us\n", m, name, 1e3 * ms / 20);
            return 0;
        };
        timeit("copy S (D2D memcpyAsync)", [&] { (void)hipMemcpyAsync(dS, dS0, 8 * m * m, hipMemcpyDeviceToDevice, st); });
        timeit("copy + rocsolver_dpotrf(lower)", [&] { (void)hipMemcpyAsync(dS, dS0, 8 * m * m, hipMemcpyDeviceToDevice, st); rocsolver_dpotrf(h, rocblas_fill_lower, m, dS, m, info); });
        timeit("copyQ + rocblas_dtrsm(right,lower,T) 240 x m", [&] { (void)hipMemcpyAsync(dQ1, dQ, 8 * n * m, hipMemcpyDeviceToDevice, st);
            rocblas_dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, m, &one, dS, m, dQ1, n); });
        timeit("rocblas_dtrtri(lower) m x m", [&] { rocblas_dtrtri(h, rocblas_fill_lower, rocblas_diagonal_non_unit, m, dS, m, dInv, m); });
        timeit("rocblas_dgemm 240 x m x m", [&] { rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, m, m, &one, dQ, n, dInv, m, &zero, dQ1, n); });
        timeit("rocblas_dgemm m x m x 240 (Q^T Q)", [&] { rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, m, m, n, &one, dQ, n, dQ, n, &zero, dS, m); });
        timeit("rocsolver_dgeqrf 240 x m", [&] { (void)hipMemcpyAsync(dQ1, dQ, 8 * n * m, hipMemcpyDeviceToDevice, st); rocsolver_dgeqrf(h, n, m, dQ1, n, dInv); });
    }
    return 0;
}
