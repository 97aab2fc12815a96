// standalone timing / ablation harness for k_fgemm64 (tnml_amd/csrc/kernels_gemm.hip)
#include "../../tnml_amd/csrc/kernels_gemm.hip"
#include <cstdarg>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*) {}
void prof_end(tnml_ctx*, int, hipEvent_t) {}
template <int RT, int CT, int WR, int WC, int KT, int DB, int ABL>
static void run(const char* tag, Fgemm64Args a) {
    constexpr int BM = 16 * RT * WR, BN = 16 * CT * WC;
    dim3 grid(a.NTp / BM, (a.Np + BN - 1) / BN, a.L);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_fgemm64<RT, CT, WR, WC, KT, DB, ABL>), grid, dim3(64 * WR * WC), 0, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("%-44s %8.1f us  %6.1f TF\n", tag, best * 1e3, 2.0 * a.NTp * a.Kp * a.Np / best / 1e9);
}
int main(int argc, char** argv) {
    const bool zero = argc > 1;
    const int NTp = 60160, m = 120, Kp = 240, Np = 240;
    float *E, *phi; double *M, *out;
    hipMalloc(&E, sizeof(float) * (size_t)m * NTp); hipMalloc(&phi, sizeof(float) * 4 * NTp);
    hipMalloc(&M, sizeof(double) * Kp * Np); hipMalloc(&out, sizeof(double) * (size_t)m * NTp);
    std::vector<float> h((size_t)m * NTp); srand(1); for (auto& x : h) x = zero ? 0.f : rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(E, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
    hipMemcpy(phi, h.data(), sizeof(float) * 4 * NTp, hipMemcpyHostToDevice);
    std::vector<double> hm(Kp * Np); for (auto& x : hm) x = zero ? 0. : rand() / (double)RAND_MAX - 0.5;
    hipMemcpy(M, hm.data(), sizeof(double) * hm.size(), hipMemcpyHostToDevice);
    Fgemm64Args a{E, 0, m, phi, M, 0, Kp, Np, phi + 2 * NTp, out, 0, m, NTp, 1};
    run<2, 5, 4, 3, 16, 0, 0>("128x240 12w KT16 full", a);
    run<2, 5, 4, 3, 16, 0, 2>("128x240 12w KT16 MFMA-only (no staging)", a);
    run<2, 5, 4, 3, 16, 0, 3>("128x240 12w KT16 MFMA only, no LDS fragment loads", a);
    { Fgemm64Args b = a; b.NTp = 32768; run<2, 5, 4, 3, 16, 0, 0>("   same, 256 tiles (one round): full", b); run<2, 5, 4, 3, 16, 0, 2>("   same, 256 tiles: MFMA-only (no staging)", b); run<2, 5, 4, 3, 16, 0, 3>("   same, 256 tiles: MFMA only, no LDS loads", b); }
    run<2, 5, 4, 3, 16, 1, 0>("128x240 12w KT16 dbuf full", a);
    run<2, 5, 4, 3, 16, 2, 0>("128x240 12w KT16 3buf full", a);
    run<2, 5, 4, 3, 16, 2, 2>("128x240 12w KT16 3buf MFMA-only", a);
    run<1, 5, 4, 3, 16, 2, 0>("64x240 12w KT16 3buf full", a);
    run<1, 5, 5, 3, 16, 2, 0>("80x240 15w KT16 3buf full", a);
    run<1, 5, 5, 3, 16, 2, 2>("80x240 15w KT16 3buf MFMA-only", a);
    run<1, 5, 5, 3, 16, 0, 0>("80x240 15w KT16 full", a);
    run<2, 5, 4, 3, 8, 2, 0>("128x240 12w KT8 3buf full", a);
    return 0;
}
