// k_chol_rinv_blocked (tnml_amd/csrc/eigh.hip): S = L L^T, Rinv = L^-T of a 120 x 120 Gram matrix S = Q^T Q -- host check of
// Rinv^T S Rinv = I and of the triangle, time of one launch with every panel factored and with the adaptive panel count.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Itnml_amd/csrc -Iinclude tools/probe/probe_cholqr.hip -o tools/probe/probe_cholqr
#include "../../tnml_amd/csrc/eigh.hip"
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*) {}
void prof_end(tnml_ctx*, int, hipEvent_t) {}
int eigh_mc_tridiagonalize(tnml_ctx*, hipStream_t, const double*, int, double*, double*, double*, double*, double, void*, unsigned*, long long*, int, int) { return 1; }
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    struct Case { int m, nbad; double eps; const char* what; };
    const Case cases[] = {{120, 120, 0.3, "m = 120, every column perturbed by 0.3"}, {120, 56, 0.3, "m = 120, the first 56 columns perturbed (7 panels)"},
                          {120, 120, 1e-3, "m = 120, perturbed by 1e-3"}, {117, 117, 0.3, "m = 117 (ragged last tile)"}, {8, 8, 0.3, "m = 8"}, {128, 128, 0.5, "m = 128"}};
    for (const Case& cs : cases) {
        const int m = cs.m, n = 2 * m;
        std::vector<double> Q((size_t)n * m, 0.), S((size_t)m * m);
        srand(5);
        for (int j = 0; j < m; ++j) {
            Q[j + (size_t)n * j] = 1.;
            if (j < cs.nbad) for (int i = 0; i < n; ++i) Q[i + (size_t)n * j] += cs.eps * (rand() / (double)RAND_MAX - 0.5) / std::sqrt((double)n) * 4.;
        }
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double s = 0.; for (int k = 0; k < n; ++k) s += Q[k + (size_t)n * i] * Q[k + (size_t)n * j]; S[i + (size_t)m * j] = s; }
        double *dS, *dR, *dF;
        HC(hipMalloc(&dS, 8 * m * m)); HC(hipMalloc(&dR, 8 * m * m)); HC(hipMalloc(&dF, 8 * 8));
        HC(hipMemcpy(dS, S.data(), 8 * m * m, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        for (int all = 1; all >= 0; --all) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                HC(hipMemset(dR, 0xff, 8 * m * m));
                HC(hipEventRecord(e0));
                hipLaunchKernelGGL(k_chol_rinv_blocked, dim3(1), dim3(256), 0, 0, dS, m, dR, dF + 1, all, 0);
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
            }
            HC(hipGetLastError());
            std::vector<double> R((size_t)m * m); double F[8];
            HC(hipMemcpy(R.data(), dR, 8 * m * m, hipMemcpyDeviceToHost)); HC(hipMemcpy(F, dF, 64, hipMemcpyDeviceToHost));
            // T = Rinv^T S Rinv, lower part of Rinv must be zero
            double dev = 0., low = 0.;
            std::vector<double> SR((size_t)m * m);
            for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double s = 0.; for (int k = 0; k < m; ++k) s += S[i + (size_t)m * k] * R[k + (size_t)m * j]; SR[i + (size_t)m * j] = s; if (i > j) low = std::fmax(low, std::fabs(R[i + (size_t)m * j])); }
            for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double s = 0.; for (int k = 0; k < m; ++k) s += R[k + (size_t)m * i] * SR[k + (size_t)m * j]; dev = std::fmax(dev, std::fabs(s - (i == j ? 1. : 0.))); }
            printf("%-52s %s | %6.1f us | max |Rinv^T S Rinv - I| %.1e  below the diagonal %.1e  flags fail %.0f factored %.0f  max|S - I| in %.1e\n",
                   cs.what, all ? "all panels     " : "adaptive panels", best * 1e3f, dev, low, F[1], F[2], F[3]);
        }
        (void)hipFree(dS); (void)hipFree(dR); (void)hipFree(dF);
    }
    return 0;
}
