// k_dgemm_small (tnml_amd/csrc/kernels_sgemm.hip) against a host reference and against rocBLAS at the sizes of the bond-tensor split.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Itnml_amd/csrc -Iinclude tools/probe/probe_sgemm.hip -o tools/probe/probe_sgemm -lrocblas
#include "../../tnml_amd/csrc/kernels_sgemm.hip"
#include <cmath>
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*, hipStream_t) {}
void prof_end(tnml_ctx*, int, hipEvent_t, hipStream_t) {}
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    tnml_ctx ctx;
    rocblas_handle h; rocblas_create_handle(&h);
    struct Case { int M, N, K, ta, tb, bmode; const char* what; };
    const Case cases[] = {
        {240, 240, 240, 0, 1, 0, "Gram M M^T"}, {240, 240, 240, 1, 0, 0, "Gram M^T M"}, {120, 120, 240, 1, 0, 0, "Q^T Q"}, {240, 120, 120, 0, 0, 0, "Q R^-1"},
        {240, 120, 120, 0, 0, 1, "Q (1.5 I - 0.5 S)"}, {120, 240, 240, 1, 0, 0, "U^T M"}, {240, 120, 240, 0, 0, 0, "M V"}, {240, 240, 120, 0, 0, 0, "A_b A_b+1"},
        {37, 21, 13, 1, 1, 0, "odd sizes T T"}, {33, 50, 70, 0, 1, 0, "odd sizes N T"}, {50, 50, 50, 0, 0, 1, "odd polish"}, {600, 600, 600, 1, 0, 0, "Gram n = 600"},
        {300, 300, 600, 1, 0, 0, "Q^T Q n = 600"}, {600, 300, 300, 0, 0, 0, "Q R^-1 n = 600"}, {240, 2400, 120, 0, 0, 0, "A_b A_b+1, Label on the right site"},
    };
    for (const Case& cs : cases) {
        const int M = cs.M, N = cs.N, K = cs.K;
        const int ar = cs.ta ? K : M, ac = cs.ta ? M : K, br = cs.tb ? N : K, bc = cs.tb ? K : N;
        std::vector<double> A((size_t)ar * ac), B((size_t)br * bc), C((size_t)M * N), R((size_t)M * N);
        srand(5);
        for (auto& v : A) v = rand() / (double)RAND_MAX - 0.5;
        for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
        if (cs.bmode) { for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) { const double t = i == j ? 1. + 1e-3 * (B[i + (size_t)K * j]) : 1e-3 * 0.5 * (B[i + (size_t)K * j] + B[j + (size_t)K * i]); B[i + (size_t)K * j] = t; } for (int i = 0; i < K; ++i) for (int j = 0; j < i; ++j) B[i + (size_t)K * j] = B[j + (size_t)K * i]; }
        double devref = 0.;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0.;
                for (int k = 0; k < K; ++k) {
                    const double a = cs.ta ? A[k + (size_t)ar * i] : A[i + (size_t)ar * k];
                    double b = cs.tb ? B[j + (size_t)br * k] : B[k + (size_t)br * j];
                    if (cs.bmode) { devref = std::fmax(devref, std::fabs(b - (k == j ? 1. : 0.))); b = (k == j ? 1.5 : 0.) - 0.5 * b; }
                    s += a * b;
                }
                R[i + (size_t)M * j] = s;
            }
        double *dA, *dB, *dC, *dDev;
        HC(hipMalloc(&dA, 8 * A.size())); HC(hipMalloc(&dB, 8 * B.size())); HC(hipMalloc(&dC, 8 * C.size())); HC(hipMalloc(&dDev, 8)); HC(hipMemset(dDev, 0, 8));
        HC(hipMemcpy(dA, A.data(), 8 * A.size(), hipMemcpyHostToDevice)); HC(hipMemcpy(dB, B.data(), 8 * B.size(), hipMemcpyHostToDevice));
        SmallGemmArgs g{dA, ar, dB, br, dC, M, M, N, K, cs.ta, cs.tb, cs.bmode, dDev};
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        float best = 1e9f, bestr = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            HC(hipEventRecord(e0)); if (launch_dgemm_small(&ctx, g)) return 1; HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = fminf(best, ms);
        }
        HC(hipMemcpy(C.data(), dC, 8 * C.size(), hipMemcpyDeviceToHost));
        double dev = 0.; HC(hipMemcpy(&dev, dDev, 8, hipMemcpyDeviceToHost));
        double err = 0., mx = 0.;
        for (size_t i = 0; i < C.size(); ++i) { err = std::fmax(err, std::fabs(C[i] - R[i])); mx = std::fmax(mx, std::fabs(R[i])); }
        if (!cs.bmode) {
            const double one = 1., zero = 0.;
            for (int rep = 0; rep < 6; ++rep) {
                HC(hipEventRecord(e0));
                rocblas_dgemm(h, cs.ta ? rocblas_operation_transpose : rocblas_operation_none, cs.tb ? rocblas_operation_transpose : rocblas_operation_none, M, N, K, &one, dA, ar, dB, br, &zero, dC, M);
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (rep) bestr = fminf(bestr, ms);
            }
        }
        printf("%-36s %4d x %4d x %4d ta=%d tb=%d | k_dgemm_small %6.1f us  rocBLAS %6.1f us | max err %.1e (max |C| %.1e)", cs.what, M, N, K, cs.ta, cs.tb, best * 1e3f, cs.bmode ? 0.f : bestr * 1e3f, err, mx);
        if (cs.bmode) printf(" | max|S - I| %.6e (host %.6e)", dev, devref);
        printf("\n");
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dDev);
    }
    return 0;
}
