// one-workgroup tridiagonalisation kernels of tnml_amd/csrc/eigh.hip: timing of k_sytrd_onewg (with its per-phase
// cycle profile) and k_sytrd_v2, and agreement of their outputs (same Householder convention -> same D, E, tau, V)
#ifndef NOPROF
#define TNML_EIGH_PROF 1
#endif
#include "../../tnml_amd/csrc/eigh.hip"
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*) {}
void prof_end(tnml_ctx*, int, hipEvent_t) {}
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    for (int n : {240, 200, 37}) {
        std::vector<double> A((size_t)n * n);
        srand(1);
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = rand() / (double)RAND_MAX - 0.5; A[i + (size_t)n * j] = v; A[j + (size_t)n * i] = v; }
        double *dA, *dD[3], *dE[3], *dT[3], *dV[3]; long long* dbg;
        HC(hipMalloc(&dA, 8 * n * n)); HC(hipMalloc(&dbg, 64));
        for (int v = 0; v < 3; ++v) { HC(hipMalloc(&dV[v], 8 * n * n)); HC(hipMalloc(&dD[v], 8 * n)); HC(hipMalloc(&dE[v], 8 * n)); HC(hipMalloc(&dT[v], 8 * n)); HC(hipMemset(dV[v], 0, 8 * n * n)); }
        HC(hipMemcpy(dA, A.data(), 8 * n * n, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        const int nb = (n + TB - 1) / TB;
        int threads = TU * nb * (nb + 1) / 2; if (threads < nb * TB) threads = nb * TB; threads = (threads + 63) / 64 * 64;
        HC(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_v3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(V3_SMEM_DOUBLES * sizeof(double))));
        for (int vv = 0; vv < 4; ++vv) {
            const int ver = vv < 2 ? vv : 2;
            long long pw[8] = {0, 0, 0, 0, 0, 0, 0, vv == 3 ? 7 : 0};
            HC(hipMemcpy(dbg, pw, 64, hipMemcpyHostToDevice));
            TriArgs t{dA, n, n, dD[ver], dE[ver], dT[ver], dV[ver], n, dbg};
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                HC(hipEventRecord(e0));
                if (ver == 0) hipLaunchKernelGGL(k_sytrd_onewg, dim3(1), dim3(threads), 0, 0, t);
                else if (ver == 2) hipLaunchKernelGGL(k_sytrd_v3, dim3(1), dim3(512), V3_SMEM_DOUBLES * sizeof(double), 0, t);
                else          hipLaunchKernelGGL(k_sytrd_v2, dim3(1), dim3(512), 0, 0, t);
                HC(hipGetLastError());
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("n=%d %s: best of 5 = %.3f ms\n", n, ver == 0 ? "k_sytrd_onewg (16x16 blocks, 4 lanes each, profiled build)" : ver == 1 ? "k_sytrd_v2 (8x8 blocks, 1 lane each)" : (vv == 2 ? "k_sytrd_v3 (profile: wave 0)" : "k_sytrd_v3 (profile: wave 7)"), best);
            {
                long long h[8]; HC(hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost));
                if (ver == 0) printf("   cycles (wave 0): extract+bar %lld  householder %lld  symv+bar %lld  reduce %lld  K+w+2bar %lld  update %lld\n", h[0], h[1], h[2], h[3], h[4], h[5]);
                else          printf("   cycles (wave 0): A householder %lld  B symv %lld  wait1 %lld  C reduce %lld  wait2 %lld  D update+lookahead %lld\n", h[0], h[1], h[2], h[3], h[4], h[5]);
            }
        }
        std::vector<double> D0(n), D1(n), E0(n), E1(n), V0((size_t)n * n), V1((size_t)n * n);
        HC(hipMemcpy(D0.data(), dD[1], 8 * n, hipMemcpyDeviceToHost)); HC(hipMemcpy(D1.data(), dD[2], 8 * n, hipMemcpyDeviceToHost));
        HC(hipMemcpy(E0.data(), dE[1], 8 * (n - 1), hipMemcpyDeviceToHost)); HC(hipMemcpy(E1.data(), dE[2], 8 * (n - 1), hipMemcpyDeviceToHost));
        HC(hipMemcpy(V0.data(), dV[1], 8 * n * n, hipMemcpyDeviceToHost)); HC(hipMemcpy(V1.data(), dV[2], 8 * n * n, hipMemcpyDeviceToHost));
        double dd = 0, de = 0, dv = 0, tr0 = 0, tr1 = 0, trA = 0;
        for (int i = 0; i < n; ++i) { dd = fmax(dd, fabs(D0[i] - D1[i])); tr0 += D0[i]; tr1 += D1[i]; trA += A[i + (size_t)n * i]; }
        for (int i = 0; i < n - 1; ++i) de = fmax(de, fabs(E0[i] - E1[i]));
        for (size_t i = 0; i < (size_t)n * (n - 1); ++i) dv = fmax(dv, fabs(V0[i] - V1[i]));
        printf("   v2 vs v3: max|D-D'| %.2e  max|E-E'| %.2e  max|V-V'| %.2e   trace(A) %.12f  sum D %.12f / %.12f\n", dd, de, dv, trA, tr0, tr1);
    }
    return 0;
}
