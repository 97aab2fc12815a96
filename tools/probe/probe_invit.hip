// where the time of k_tridiag_invit goes: the kernel cut after the factorisation (1), after the reciprocal pivots (2),
// with one sweep instead of two (3), without the zero fill outside the block (4); and the eigenvalue kernels beside it
#include "../../tnml_amd/csrc/eigh.hip"
#include "_invit_var.inc"
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*, hipStream_t) {}
void prof_end(tnml_ctx*, int, hipEvent_t, hipStream_t) {}
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int VAR> static float run(TeigArgs t, int mk, size_t lds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k_tridiag_invit_v<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_tridiag_invit_v<VAR>, dim3((mk + IV_L - 1) / IV_L), dim3(64), lds, 0, t);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3f;
}
int main() {
    const int n = 240, mk = 120;
    for (int nact : {131, 60, 26}) {
        std::vector<double> D(n, 0.), E(n, 0.);
        srand(3);
        for (int i = 0; i < nact; ++i) { D[i] = 1.0 + rand() / (double)RAND_MAX; if (i < nact - 1) E[i] = 0.3 * (rand() / (double)RAND_MAX + 0.2); }
        double *dD, *dE, *dW, *dZ, *dS;
        HC(hipMalloc(&dD, 8 * n)); HC(hipMalloc(&dE, 8 * n)); HC(hipMalloc(&dW, 8 * (n + 8))); HC(hipMalloc(&dZ, 8 * (size_t)n * mk)); HC(hipMalloc(&dS, 8 * 2048));
        HC(hipMemcpy(dD, D.data(), 8 * n, hipMemcpyHostToDevice)); HC(hipMemcpy(dE, E.data(), 8 * n, hipMemcpyHostToDevice));
        TeigArgs t{dD, dE, n, dW, mk, dZ, n, dS, dS + 256, (int*)(dS + 512), (int*)(dS + 512) + 256, (int*)(dS + 512) + 512};
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        float ts = 1e9f, te = 1e9f, tr = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_tridiag_split, dim3(1), dim3(256), 0, 0, t); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ts = fminf(ts, ms);
            hipEventRecord(e0); hipLaunchKernelGGL(k_tridiag_eigvals, dim3(n), dim3(64), 0, 0, t); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); te = fminf(te, ms);
            hipEventRecord(e0); hipLaunchKernelGGL(k_tridiag_rank, dim3(1), dim3(256), 0, 0, t); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tr = fminf(tr, ms);
        }
        HC(hipGetLastError());
        const size_t lds = sizeof(double) * (512 + (size_t)5 * n * IV_L);
        printf("active block %3d of n=240, mk=120: split %.1f us  eigvals %.1f us  rank %.1f us | invit full %.1f us, LU only %.1f, LU+reciprocals %.1f, one sweep %.1f, no zero fill %.1f\n",
               nact, ts * 1e3f, te * 1e3f, tr * 1e3f, run<0>(t, mk, lds), run<1>(t, mk, lds), run<2>(t, mk, lds), run<3>(t, mk, lds), run<4>(t, mk, lds));
        HC(hipGetLastError());
    }
    return 0;
}
