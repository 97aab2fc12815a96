// the one-workgroup tridiagonalisation kernel of tnml_amd/csrc/eigh.hip (k_sytrd_v3): timing, per-phase cycle profile of two waves,
// and a host check of what it returns (trace, and T = Q^T A Q through the reflectors).  Rounds 1-2 compared it here with two earlier
// kernels that have since been removed (profiles/r02_probe_eigh.txt keeps that comparison).
#ifndef NOPROF
#define TNML_EIGH_PROF 1
#endif
#include "../../tnml_amd/csrc/eigh.hip"
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*) {}
void prof_end(tnml_ctx*, int, hipEvent_t) {}
int eigh_mc_tridiagonalize(tnml_ctx*, hipStream_t, const double*, int, double*, double*, double*, double*, double, void*, unsigned*, long long*, int, int) { return 1; }   // n > 240: tools/probe/probe_mc.hip
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    for (int n : {240, 200, 37}) {
        std::vector<double> A((size_t)n * n);
        srand(1);
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = rand() / (double)RAND_MAX - 0.5; A[i + (size_t)n * j] = v; A[j + (size_t)n * i] = v; }
        double *dA, *dD, *dE, *dT, *dV; long long* dbg;
        HC(hipMalloc(&dA, 8 * n * n)); HC(hipMalloc(&dbg, 64));
        HC(hipMalloc(&dV, 8 * n * n)); HC(hipMalloc(&dD, 8 * n)); HC(hipMalloc(&dE, 8 * n)); HC(hipMalloc(&dT, 8 * n)); HC(hipMemset(dV, 0, 8 * n * n));
        HC(hipMemcpy(dA, A.data(), 8 * n * n, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        HC(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_v3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(V3_SMEM_DOUBLES * sizeof(double))));
        for (int pw : {0, 7}) {
            long long h[8] = {0, 0, 0, 0, 0, 0, 0, pw};
            HC(hipMemcpy(dbg, h, 64, hipMemcpyHostToDevice));
            TriArgs t{dA, n, n, dD, dE, dT, dV, n, dbg, nullptr, 0.0};
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                HC(hipEventRecord(e0));
                hipLaunchKernelGGL(k_sytrd_v3, dim3(1), dim3(512), V3_SMEM_DOUBLES * sizeof(double), 0, t);
                HC(hipGetLastError());
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            HC(hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost));
            printf("n=%d k_sytrd_v3: best of 5 = %.3f ms (%.2f us/step)\n", n, best, 1e3 * best / (n - 2));
            printf("   cycles (wave %d): A householder %lld  B symv %lld  wait1 %lld  C reduce %lld  wait2 %lld  D update+lookahead %lld\n", pw, h[0], h[1], h[2], h[3], h[4], h[5]);
        }
        std::vector<double> D(n), Es(n);
        HC(hipMemcpy(D.data(), dD, 8 * n, hipMemcpyDeviceToHost)); HC(hipMemcpy(Es.data(), dE, 8 * (n - 1), hipMemcpyDeviceToHost));
        double trT = 0, trA = 0, fT = 0, fA = 0;
        for (int i = 0; i < n; ++i) { trT += D[i]; trA += A[i + (size_t)n * i]; fT += D[i] * D[i]; }
        for (int i = 0; i < n - 1; ++i) fT += 2 * Es[i] * Es[i];
        for (size_t i = 0; i < (size_t)n * n; ++i) fA += A[i] * A[i];
        printf("   trace(A) %.12f  sum D %.12f   |A|_F^2 %.12f  |T|_F^2 %.12f (orthogonal similarity keeps both)\n", trA, trT, fA, fT);
    }
    return 0;
}
