// rocBLAS dgemm at the sizes of the split, as one call and as a strided batch of column strips of C (more, smaller workgroups)
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <typename F> static float time_it(hipStream_t st, int reps, const F& f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    rocblas_handle h; rocblas_create_handle(&h);
    hipStream_t st; HC(hipStreamCreate(&st)); rocblas_set_stream(h, st);
    rocblas_set_atomics_mode(h, rocblas_atomics_not_allowed);
    double *A, *B, *C; HC(hipMalloc(&A, 8 * 2400 * 240)); HC(hipMalloc(&B, 8 * 2400 * 240)); HC(hipMalloc(&C, 8 * 240 * 240));
    HC(hipMemset(A, 0, 8 * 2400 * 240)); HC(hipMemset(B, 0, 8 * 2400 * 240));
    const double one = 1., zero = 0.;
    struct Case { const char* name; rocblas_operation ta, tb; int M, N, K, lda, ldb, ldc; };
    const Case cases[] = {
        {"Q = Q0 C      (NN 240 x 120 x 120)", rocblas_operation_none, rocblas_operation_none, 240, 120, 120, 240, 120, 240},
        {"S = Q0^T Q0   (TN 120 x 120 x 240)", rocblas_operation_transpose, rocblas_operation_none, 120, 120, 240, 240, 240, 120},
        {"G = M M^T     (NT 240 x 240 x 240)", rocblas_operation_none, rocblas_operation_transpose, 240, 240, 240, 240, 240, 240},
        {"G = M^T M     (TN 240 x 240 x 240)", rocblas_operation_transpose, rocblas_operation_none, 240, 240, 240, 240, 240, 240},
        {"SV = U^T M    (TN 120 x 240 x 240)", rocblas_operation_transpose, rocblas_operation_none, 120, 240, 240, 240, 240, 120},
        {"B = A1 A2     (NN 240 x 240 x 120)", rocblas_operation_none, rocblas_operation_none, 240, 240, 120, 240, 120, 240},
        {"U S = M V     (NN 240 x 120 x 240)", rocblas_operation_none, rocblas_operation_none, 240, 120, 240, 240, 240, 240},
        {"Gram, K=2400  (NT 240 x 240 x 2400)", rocblas_operation_none, rocblas_operation_transpose, 240, 240, 2400, 240, 240, 240},
    };
    for (const Case& c : cases) {
        printf("%s:", c.name);
        const float t1 = time_it(st, 50, [&] { rocblas_dgemm(h, c.ta, c.tb, c.M, c.N, c.K, &one, A, c.lda, B, c.ldb, &zero, C, c.ldc); });
        printf("  one call %.1f us |", t1);
        for (int strips : {2, 4, 8}) {
            if (c.N % strips) continue;
            const int ns = c.N / strips;
            // op(B) column j0..j0+ns: not transposed -> B + j0*ldb; transposed -> B + j0
            const rocblas_stride sb = c.tb == rocblas_operation_none ? (rocblas_stride)ns * c.ldb : ns, sc = (rocblas_stride)ns * c.ldc;
            const float t = time_it(st, 50, [&] { rocblas_dgemm_strided_batched(h, c.ta, c.tb, c.M, ns, c.K, &one, A, c.lda, 0, B, c.ldb, sb, &zero, C, c.ldc, sc, strips); });
            printf("  %d strips %.1f", strips, t);
        }
        printf("\n");
    }
    return 0;
}
