// k_sytrd_mc (tnml_amd/csrc/eigh_mc.hip): the multi-workgroup tridiagonalisation against a host application of its own reflectors
// (H_{nref-1} ... H_0 A H_0 ... H_{nref-1} must be tridiagonal with the returned D, E), for full-rank symmetric matrices and for
// rank-deficient Gram matrices with the rank-adaptive exit; timing, and repeat runs must be bit-identical.
#define MC_PROF 1
#include "../../tnml_amd/csrc/eigh_mc.hip"
#include <cstdarg>
#include <cstring>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double urand() { return rand() / (double)RAND_MAX - 0.5; }

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : 0;
    const int xp = argc > 2 ? atoi(argv[2]) : 0;                      // 99 / 98 / 97: timing experiments without validation / publishes / polls (results are garbage)
    void* xbuf; HC(hipMalloc(&xbuf, eigh_mc_xbuf_bytes())); HC(hipMemset(xbuf, 0, eigh_mc_xbuf_bytes()));
    unsigned epoch = 0;
    long long* dbg; HC(hipMalloc(&dbg, 128)); HC(hipMemset(dbg, 0, 128));
    struct Case { int n, rank; double tol; };
    const Case cases[] = {{600, 0, 0.}, {600, 320, 1e-15}, {597, 0, 0.}, {640, 330, 1e-15}, {320, 0, 0.}, {250, 130, 1e-15}, {480, 250, 1e-15}, {241, 0, 0.}};
    for (const Case& cs : cases) {
        const int n = cs.n;
        if (only && n != only) continue;
        std::vector<double> A((size_t)n * n, 0.);
        srand(7 + n);
        if (cs.rank == 0) {
            for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { const double v = urand(); A[i + (size_t)n * j] = v; A[j + (size_t)n * i] = v; }
        } else {                                                     // Gram matrix of an n x rank factor with a graded spectrum
            std::vector<double> F((size_t)n * cs.rank);
            for (int r = 0; r < cs.rank; ++r) { const double s = pow(10., -6. * r / cs.rank); for (int i = 0; i < n; ++i) F[i + (size_t)n * r] = s * urand(); }
            for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
                double t = 0.; for (int r = 0; r < cs.rank; ++r) t += F[i + (size_t)n * r] * F[j + (size_t)n * r];
                A[i + (size_t)n * j] = t; A[j + (size_t)n * i] = t;
            }
        }
        double *dA, *dD, *dE, *dT, *dV;
        HC(hipMalloc(&dA, 8 * (size_t)n * n)); HC(hipMalloc(&dV, 8 * (size_t)n * n)); HC(hipMalloc(&dD, 8 * n)); HC(hipMalloc(&dE, 8 * n)); HC(hipMalloc(&dT, 8 * n));
        HC(hipMemcpy(dA, A.data(), 8 * (size_t)n * n, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        std::vector<double> D(n), E(n), tau(n), V((size_t)n * n), D2(n), E2(n), V2((size_t)n * n);
        float best = 1e9f;
        bool same = true;
        for (int rep = 0; rep < 4; ++rep) {
            HC(hipMemset(dV, 0, 8 * (size_t)n * n)); HC(hipMemset(dD, 0, 8 * n)); HC(hipMemset(dE, 0, 8 * n)); HC(hipMemset(dT, 0, 8 * n));
            HC(hipEventRecord(e0));
            { long long pwv = rep == 3 ? 6 : 0; HC(hipMemcpy(dbg + 15, &pwv, 8, hipMemcpyHostToDevice)); }
            if (eigh_mc_tridiagonalize(nullptr, 0, dA, n, dD, dE, dT, dV, cs.tol, xbuf, &epoch, dbg, xp)) return 1;
            HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            if (rep >= 2) {
                long long h[15]; HC(hipMemcpy(h, dbg, 120, hipMemcpyDeviceToHost));
                printf("   cycles wg0 wave %d: A scalars %lld  B block products %lld  bar1 %lld  C1 row sums + publish %lld  C2 poll + next column %lld  bar2 %lld  D rank-2 update %lld | failed polls lane 0 (scalars + row) %lld, lane 1 %lld, lane 20 (row only) %lld\n",
                       rep == 3 ? 6 : 0, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
            }
            unsigned long long status = 0; HC(hipMemcpy(&status, eigh_mc_status_ptr(xbuf), 8, hipMemcpyDeviceToHost));
            if (status) { printf("n=%d: kernel aborted (status %llu)\n", n, status); return 1; }
            HC(hipMemcpy(rep ? D2.data() : D.data(), dD, 8 * n, hipMemcpyDeviceToHost));
            HC(hipMemcpy(rep ? E2.data() : E.data(), dE, 8 * n, hipMemcpyDeviceToHost));
            HC(hipMemcpy(rep ? V2.data() : V.data(), dV, 8 * (size_t)n * n, hipMemcpyDeviceToHost));
            if (rep) same = same && !memcmp(D.data(), D2.data(), 8 * n) && !memcmp(E.data(), E2.data(), 8 * (n - 1)) && !memcmp(V.data(), V2.data(), 8 * (size_t)n * (n - 1));
        }
        HC(hipMemcpy(tau.data(), dT, 8 * n, hipMemcpyDeviceToHost));
        const int nref = (int)tau[n - 1];
        // host: B = H ... A ... H
        std::vector<double> B = A, pv(n), qv(n);
        double anorm = 0.; for (double t : A) anorm = fmax(anorm, fabs(t));
        for (int k = 0; k < nref && k < n - 1; ++k) {
            const double* v = &V[(size_t)n * k];
            const double t = tau[k];
            for (int i = 0; i < n; ++i) { double s = 0.; for (int j = 0; j < n; ++j) s += B[i + (size_t)n * j] * v[j]; pv[i] = t * s; }
            double pvv = 0.; for (int i = 0; i < n; ++i) pvv += pv[i] * v[i];
            for (int i = 0; i < n; ++i) qv[i] = pv[i] - 0.5 * t * pvv * v[i];
            for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) B[i + (size_t)n * j] -= v[i] * qv[j] + qv[i] * v[j];
        }
        double ed = 0., ee = 0., eo = 0., tail = 0.;
        for (int i = 0; i < n; ++i) {
            if (i <= nref) ed = fmax(ed, fabs(B[i + (size_t)n * i] - D[i]));
            if (i < nref && i < n - 1) ee = fmax(ee, fabs(B[i + 1 + (size_t)n * i] - E[i]));
            for (int j = 0; j < n; ++j) {
                if (abs(i - j) <= 1 && i <= nref && j <= nref) continue;
                if (i > nref && j > nref) tail = fmax(tail, fabs(B[i + (size_t)n * j])); else if (abs(i - j) > 1) eo = fmax(eo, fabs(B[i + (size_t)n * j]));
                else if (i > nref || j > nref) eo = fmax(eo, fabs(B[i + (size_t)n * j]));
            }
        }
        double tr = 0.; for (int i = 0; i < n; ++i) tr += A[i + (size_t)n * i];
        printf("n=%d rank=%d tol=%.0e: %d workgroups, %.3f ms (%.2f us/step), reflectors %d | max|A| %.2e  |diag-D| %.2e  |sub-E| %.2e  off-tridiagonal %.2e  dropped block %.2e (tol*trace %.2e)  repeat runs %s\n",
               n, cs.rank, cs.tol, mc_workgroups(n), best, best * 1000. / (nref > 0 ? nref : 1), nref, anorm, ed, ee, eo, tail, cs.tol * tr, same ? "bit-identical" : "DIFFER");
        fflush(stdout);
        hipFree(dA); hipFree(dV); hipFree(dD); hipFree(dE); hipFree(dT);
    }
    return 0;
}
