// The tridiagonal eigen stage of the split (eigh_tri.hip: k_teig_values / k_teig_vectors, two launches) on tridiagonal problems of the
// shapes the split meets, with a host check: residuals |T z - lambda z| / |T|, norms, and (for information) max |Z^T Z - I|.
// Built with -DTEIG_AB and the round-4 kernels (git show 72b01de:tnml_amd/csrc/eigh.hip) it ran the A/B of
// profiles/r05_probe_tridiagonal_eigensolver_r4_vs_r5.txt; the old kernels are gone from the tree.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Itnml_amd/csrc -Iinclude tools/probe/probe_teig.hip -o tools/probe/probe_teig
#include "../../tnml_amd/csrc/eigh_tri.hip"
#include <cmath>
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*, hipStream_t) {}
void prof_end(tnml_ctx*, int, hipEvent_t, hipStream_t) {}
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Check { double res, nrm, orth; };
static Check check(int n, int mk, const std::vector<double>& D, const std::vector<double>& E, const std::vector<double>& W, const std::vector<double>& Z) {
    double tn = 0.;
    for (int i = 0; i < n; ++i) tn = std::fmax(tn, std::fabs(D[i]) + (i ? std::fabs(E[i - 1]) : 0.) + (i < n - 1 ? std::fabs(E[i]) : 0.));
    if (!(tn > 0.)) tn = 1.;
    Check c{0., 0., 0.};
    for (int g = 0; g < mk; ++g) {
        const double lam = W[n - 1 - g];
        const double* z = &Z[(size_t)n * g];
        double r2 = 0., z2 = 0.;
        for (int i = 0; i < n; ++i) {
            const double t = D[i] * z[i] + (i ? E[i - 1] * z[i - 1] : 0.) + (i < n - 1 ? E[i] * z[i + 1] : 0.) - lam * z[i];
            r2 += t * t; z2 += z[i] * z[i];
        }
        c.res = std::fmax(c.res, std::sqrt(r2) / tn);
        c.nrm = std::fmax(c.nrm, std::fabs(std::sqrt(z2) - 1.));
        if (!(z2 == z2)) c.nrm = 1e300;
    }
    for (int g = 0; g < mk; ++g)
        for (int h = 0; h < g; ++h) {
            double d = 0.;
            for (int i = 0; i < n; ++i) d += Z[(size_t)n * g + i] * Z[(size_t)n * h + i];
            c.orth = std::fmax(c.orth, std::fabs(d));
        }
    return c;
}

int main() {
    tnml_ctx ctx;
    struct Case { int n, mk, nact, kind; const char* what; };
    const Case cases[] = {
        {240, 120, 131, 0, "random block 131 + decoupled rows (the driver's window)"},
        {240, 120, 240, 0, "random, unreduced"},
        {240, 120, 60, 0, "random block 60"},
        {240, 120, 26, 0, "random block 26"},
        {240, 120, 240, 1, "graded over 14 decades (a Gram matrix)"},
        {240, 120, 200, 2, "clusters: d = 1, couplings 1e-7"},
        {240, 120, 240, 3, "two blocks + singles"},
        {37, 20, 37, 0, "n = 37"},
        {3, 3, 3, 0, "n = 3"},
        {300, 150, 170, 0, "n = 300 (8 vectors per workgroup)"},
        {600, 300, 330, 0, "n = 600 (4 vectors per workgroup; config 5)"},
        {600, 300, 600, 1, "n = 600 graded"},
    };
    for (const Case& cs : cases) {
        const int n = cs.n, mk = cs.mk;
        std::vector<double> D(n, 0.), E(n, 0.);
        srand(3);
        auto rnd = []() { return rand() / (double)RAND_MAX; };
        for (int i = 0; i < cs.nact; ++i) {
            if (cs.kind == 0) { D[i] = 1.0 + rnd(); if (i < cs.nact - 1) E[i] = 0.3 * (rnd() + 0.2); }
            else if (cs.kind == 1) { D[i] = std::pow(10., -14. * i / n) * (1. + 0.3 * rnd()); if (i < cs.nact - 1) E[i] = 0.4 * std::pow(10., -14. * (i + 0.5) / n) * (rnd() - 0.5); }
            else if (cs.kind == 2) { D[i] = 1.0; if (i < cs.nact - 1) E[i] = 1e-7 * (rnd() + 0.5); }
            else { D[i] = 1.0 + rnd(); if (i < cs.nact - 1) E[i] = (i == 99 || (i > 200 && i % 3 == 0)) ? 0. : 0.3 * (rnd() + 0.2); }
        }
        double *dD, *dE, *dW[2], *dZ[2], *dS;
        HC(hipMalloc(&dD, 8 * n)); HC(hipMalloc(&dE, 8 * n)); HC(hipMalloc(&dS, 8 * TEIG_SCRATCH_DOUBLES));
        for (int v = 0; v < 2; ++v) { HC(hipMalloc(&dW[v], 8 * (n + 8))); HC(hipMalloc(&dZ[v], 8 * (size_t)n * mk)); HC(hipMemset(dZ[v], 0xff, 8 * (size_t)n * mk)); }
        HC(hipMemcpy(dD, D.data(), 8 * n, hipMemcpyHostToDevice)); HC(hipMemcpy(dE, E.data(), 8 * n, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        float best[2] = {1e9f, 1e9f};
        for (int v = 1; v < 2; ++v)
            for (int rep = 0; rep < 6; ++rep) {
                HC(hipEventRecord(e0));
                const int rc = eigh_tridiag_eig(&ctx, dD, dE, n, dW[1], mk, dZ[1], n, dS);
                if (rc) return 1;
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best[v]) best[v] = ms;
            }
        // the two new kernels one by one
        float tv[2] = {1e9f, 1e9f};
        {
            int* is = (int*)(dS + 3 * TEIG_MAXN);
            Teig2Args t{dD, dE, n, dW[1], mk, dZ[1], n, dS, dS + TEIG_MAXN, dS + 2 * TEIG_MAXN, is, is + TEIG_MAXN, is + 2 * TEIG_MAXN, nullptr};
            const int ns = (n + 63) & ~63, ivl = n <= 248 ? 16 : (n <= 448 ? 8 : 4);
            const size_t lds = sizeof(double) * (3 * (size_t)ns + (size_t)4 * n * ivl);
            for (int rep = 0; rep < 5; ++rep) {
                float ms;
                HC(hipEventRecord(e0)); hipLaunchKernelGGL(k_teig_values, dim3(n), dim3(64), 0, 0, t); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipEventElapsedTime(&ms, e0, e1)); tv[0] = fminf(tv[0], ms);
                HC(hipEventRecord(e0));
                if (ivl == 16) hipLaunchKernelGGL((k_teig_vectors<16>), dim3((mk + 15) / 16), dim3(256), lds, 0, t);
                else if (ivl == 8) hipLaunchKernelGGL((k_teig_vectors<8>), dim3((mk + 7) / 8), dim3(256), lds, 0, t);
                else hipLaunchKernelGGL((k_teig_vectors<4>), dim3((mk + 3) / 4), dim3(256), lds, 0, t);
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipEventElapsedTime(&ms, e0, e1)); tv[1] = fminf(tv[1], ms);
            }
            HC(hipGetLastError());
        }
        std::vector<double> W[2], Z[2];
        for (int v = 1; v < 2; ++v) {
            W[v].resize(n); Z[v].resize((size_t)n * mk);
            HC(hipMemcpy(W[v].data(), dW[v], 8 * n, hipMemcpyDeviceToHost)); HC(hipMemcpy(Z[v].data(), dZ[v], 8 * (size_t)n * mk, hipMemcpyDeviceToHost));
        }
        const Check c1 = check(n, mk, D, E, W[1], Z[1]);
        printf("%-58s n=%3d mk=%3d | %6.1f us (values %.1f + vectors %.1f) | residual %.1e | norm-1 %.1e | max|z_g.z_h| %.1e\n",
               cs.what, n, mk, best[1] * 1e3f, tv[0] * 1e3f, tv[1] * 1e3f, c1.res, c1.nrm, c1.orth);
        (void)hipFree(dD); (void)hipFree(dE); (void)hipFree(dS);
        for (int v = 0; v < 2; ++v) { (void)hipFree(dW[v]); (void)hipFree(dZ[v]); }
    }
    return 0;
}
