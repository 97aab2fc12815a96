// per-phase cycle profile of the one-workgroup tridiagonalisation (tnml_amd/csrc/eigh.hip)
#define TNML_EIGH_PROF 1
#include "../../tnml_amd/csrc/eigh.hip"
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*) {}
void prof_end(tnml_ctx*, int, hipEvent_t) {}
int main() {
    const int n = 240;
    std::vector<double> A((size_t)n * n);
    srand(1);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = rand() / (double)RAND_MAX - 0.5; A[i + (size_t)n * j] = v; A[j + (size_t)n * i] = v; }
    double *dA, *dD, *dE, *dT, *dV; long long* dbg;
    hipMalloc(&dA, 8 * n * n); hipMalloc(&dV, 8 * n * n); hipMalloc(&dD, 8 * n); hipMalloc(&dE, 8 * n); hipMalloc(&dT, 8 * n); hipMalloc(&dbg, 64);
    hipMemcpy(dA, A.data(), 8 * n * n, hipMemcpyHostToDevice);
    TriArgs t{dA, n, n, dD, dE, dT, dV, n, dbg};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_sytrd_onewg, dim3(1), dim3(512), 0, 0, t);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost);
        long long tot = 0; for (int i = 0; i < 6; ++i) tot += h[i];
        printf("rep %d: %.3f ms; cycles extract+bar %lld  householder %lld  symv+bar %lld  reduce %lld  K+w+2bar %lld  update %lld  | total %lld cyc, wall %.1f us (100MHz ticks %lld) => %.2f GHz\n",
               rep, ms, h[0], h[1], h[2], h[3], h[4], h[5], tot, h[6] / 100.0, h[6], tot / (h[6] * 10.0));
    }
    return 0;
}
