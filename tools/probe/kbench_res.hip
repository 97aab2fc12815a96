// standalone check + timing of the resident-operand kernels (tnml_amd/csrc/kernels_res.hip) against the fused forward kernel
// (kernels_fused.hip) and the tiled gradient GEMM (kernels_gemm.hip) they replace, at BASELINE config 3 shape.
//   kbench_res [NT] [grid]
#include "../../tnml_amd/csrc/kernels_res.hip"
#include "../../tnml_amd/csrc/kernels_fused.hip"
#include "../../tnml_amd/csrc/kernels_stream.hip"
#include "../../tnml_amd/csrc/kernels_gemm.hip"
#include "k_grad_res_q.inc"   // the two resident-accumulator gradient forms of round 4 (out of the library since round 5)
#include "k_grad_h.inc"       // the fourth gradient form (not in the library): tools/probe/attic/k_grad_h_attempt.hip.txt + ablation switches
#include <cstdarg>
#include <cstdlib>
#include <vector>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*, hipStream_t) {}
void prof_end(tnml_ctx*, int, hipEvent_t, hipStream_t) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc((void**)&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static double maxrel(const std::vector<double>& a, const std::vector<double>& b, size_t n) {
    double mx = 0., sc = 0.;
    for (size_t i = 0; i < n; ++i) { sc = fmax(sc, fabs(b[i])); mx = fmax(mx, fabs(a[i] - b[i])); }
    return mx / (sc > 0. ? sc : 1.);
}
static hipStream_t g_st;
// mean time of one call of f (asynchronous launches on g_st), 10 calls back to back between two events
template <typename F> static float time_it(F f, int reps = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipStreamSynchronize(g_st);
    hipEventRecord(e0, g_st); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1, g_st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main(int argc, char** argv) {
    const int NT = argc > 1 ? atoi(argv[1]) : 60000;
    const int NTp = (NT + 255) / 256 * 256, m = 120, Kp = 240, Np = 240;
    tnml_ctx ctx; tnml_ctx* c = &ctx;
    c->cfg.device = 0; c->cfg.dtype = TNML_F64; c->cfg.mode = TNML_MODE_FIXEDL; c->NTp = NTp; c->NT = NT;
    if (argc > 2) c->res_grid = atoi(argv[2]);
    c->partial_cap = NTp / 64;
    c->slab_bytes = (size_t)128 * Kp * Kp * 8;
    double* EI = dalloc<double>((size_t)m * NTp); double* EL = dalloc<double>((size_t)10 * m * NTp);
    double* phiI = dalloc<double>(2 * (size_t)NTp); double* phiO = dalloc<double>(2 * (size_t)NTp);
    double* M = dalloc<double>((size_t)Kp * Np);
    int* label = dalloc<int>(NTp);
    double *P0 = dalloc<double>(10 * (size_t)NTp), *dP0 = dalloc<double>(10 * (size_t)NTp), *P1 = dalloc<double>(10 * (size_t)NTp), *dP1 = dalloc<double>(10 * (size_t)NTp);
    double* Ppart = dalloc<double>(20 * (size_t)NTp);
    c->partials = dalloc<double>((size_t)c->partial_cap * 12);
    c->slab = dalloc<char>(c->slab_bytes);
    c->counters = dalloc<unsigned>(16);
    double* tail0 = dalloc<double>(64); double* tail1 = dalloc<double>(64);
    double* G0 = dalloc<double>((size_t)Kp * Np); double* G1 = dalloc<double>((size_t)Kp * Np);
    {
        srand(7);
        auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
        std::vector<double> h((size_t)10 * m * NTp);
        for (int l = 0; l < 10; ++l) for (int q = 0; q < m; ++q) for (int n = 0; n < NTp; ++n) h[((size_t)l * m + q) * NTp + n] = n < NT ? rnd() : 0.;
        CK(hipMemcpy(EL, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        for (int q = 0; q < m; ++q) for (int n = 0; n < NTp; ++n) h[(size_t)q * NTp + n] = n < NT ? rnd() : 0.;
        CK(hipMemcpy(EI, h.data(), (size_t)m * NTp * 8, hipMemcpyHostToDevice));
        for (int s = 0; s < 2; ++s) for (int n = 0; n < NTp; ++n) h[(size_t)s * NTp + n] = n < NT ? (s ? 0.3 * rnd() : 1.0) : 0.;
        CK(hipMemcpy(phiI, h.data(), 2 * (size_t)NTp * 8, hipMemcpyHostToDevice));
        for (int s = 0; s < 2; ++s) for (int n = 0; n < NTp; ++n) h[(size_t)s * NTp + n] = n < NT ? (s ? 0.3 * rnd() : 1.0) : 0.;
        CK(hipMemcpy(phiO, h.data(), 2 * (size_t)NTp * 8, hipMemcpyHostToDevice));
        for (int i = 0; i < Kp * Np; ++i) h[i] = 0.05 * rnd();
        CK(hipMemcpy(M, h.data(), (size_t)Kp * Np * 8, hipMemcpyHostToDevice));
        std::vector<int> lab(NTp); for (int n = 0; n < NTp; ++n) lab[n] = n < NT ? rand() % 10 : -1;
        CK(hipMemcpy(label, lab.data(), NTp * 4, hipMemcpyHostToDevice));
    }
    CK(hipStreamCreate(&c->stream)); g_st = c->stream;
    for (int mode : {LD_MODE_COST, LD_MODE_PAP}) {
        FwdFusedArgs ff; ff.EI = EI; ff.mI = m; ff.phiI = phiI; ff.M = M; ff.Kp = Kp; ff.Np = Np; ff.phiO = phiO; ff.EL = EL; ff.EL_lstride = (size_t)m * NTp;
        ff.mO = m; ff.NTp = NTp; ff.ntiles = NTp / 64; ff.label = label; ff.P = P0; ff.dP = mode == LD_MODE_PAP ? nullptr : dP0; ff.mode = mode; ff.partials = c->partials;
        auto run_old = [&]() { launch_fwd_fused(c, ff); launch_labeldot_reduce(c, ff.ntiles, tail0, mode == LD_MODE_PAP ? 1 : 0); };
        FwdResArgs fr{EI, phiI, M, phiO, EL, (size_t)m * NTp, NTp, NTp / 32, Ppart};
        PfinishArgs pf{2, Ppart, nullptr, nullptr, nullptr, nullptr, label, NTp, P1, mode == LD_MODE_PAP ? nullptr : dP1, mode, c->partials, c->counters, tail1, mode == LD_MODE_PAP ? 1 : 0};
        auto run_new = [&]() { launch_fwd_res(c, fr); launch_pfinish(c, pf); };
        CK(hipMemset(tail0, 0, 64 * 8)); CK(hipMemset(tail1, 0, 64 * 8));
        run_old(); CK(hipStreamSynchronize(c->stream));
        run_new(); CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
        std::vector<double> a(10 * (size_t)NTp), b(10 * (size_t)NTp), t0(12), t1(12);
        CK(hipMemcpy(a.data(), P1, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), P0, b.size() * 8, hipMemcpyDeviceToHost));
        printf("mode %d: P max rel diff %.3e", mode, maxrel(a, b, a.size()));
        if (mode == LD_MODE_COST) { CK(hipMemcpy(a.data(), dP1, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dP0, b.size() * 8, hipMemcpyDeviceToHost)); printf("  dP %.3e", maxrel(a, b, a.size())); }
        CK(hipMemcpy(t0.data(), tail0, 96, hipMemcpyDeviceToHost)); CK(hipMemcpy(t1.data(), tail1, 96, hipMemcpyDeviceToHost));
        double tr = 0.; for (int i = 0; i < 12; ++i) tr = fmax(tr, fabs(t0[i] - t1[i]) / fmax(1., fabs(t0[i])));
        printf("  tail rel %.3e (cost0 %.6f vs %.6f, ncorrect %.0f vs %.0f, sum %.6f vs %.6f)\n", tr, t1[0], t0[0], t1[10], t0[10], t1[11], t0[11]);
        run_new(); CK(hipStreamSynchronize(c->stream));       // bit-identical repeat?
        std::vector<double> a2(10 * (size_t)NTp); CK(hipMemcpy(a2.data(), P1, a2.size() * 8, hipMemcpyDeviceToHost));
        size_t nd = 0; for (size_t i = 0; i < a2.size(); ++i) nd += a2[i] != a[i] && mode == LD_MODE_PAP;
        if (mode == LD_MODE_PAP) printf("         repeat run: %zu differing outputs\n", nd);
        const float t_old = time_it([&]() { run_old(); });
        const float t_new = time_it([&]() { run_new(); });
        const float t_k = time_it([&]() { launch_fwd_res(c, fr); });
        {
            const size_t lds = sizeof(double) * FR_LDS_DOUBLES;
#define TRY(PS, PK, ABL, what) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_res<PS, PK, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                const float t_ = time_it([&]() { hipLaunchKernelGGL((k_fwd_res<PS, PK, ABL>), dim3(256), dim3(768), lds, c->stream, fr); }); printf("         %-64s %.1f us\n", what, t_ * 1e3); }
            TRY(0, 1, 0, "GEMM waves never pause:") TRY(6, 2, 0, "pause 384 cycles every 8 MFMAs (default):") TRY(4, 2, 0, "256 cycles every 8 MFMAs:") TRY(6, 3, 0, "384 cycles every 12 MFMAs:")
            TRY(4, 1, 0, "256 cycles every 4 MFMAs:") TRY(7, 2, 0, "448 cycles every 8 MFMAs:") TRY(5, 2, 0, "320 cycles every 8 MFMAs:") TRY(8, 3, 0, "512 cycles every 12 MFMAs:")
            TRY(0, 1, 1, "GEMM role alone (no environment loads):") TRY(0, 1, 2, "streaming role alone (no MFMAs):")
            if (mode == LD_MODE_PAP) {
                long long* dbg = dalloc<long long>(16 * 12 * 4);
                FwdResArgs fd = fr; fd.dbg = dbg;
                hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_res<6, 2, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((k_fwd_res<6, 2, 5>), dim3(256), dim3(768), lds, c->stream, fd); hipStreamSynchronize(c->stream);
                std::vector<long long> hd(16 * 12 * 4); hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
                printf("         per-wave clock64 ticks of the paced kernel (GEMM waves: MFMA loop / epilogue / barrier wait; streaming waves: round work / - / barrier wait):\n");
                for (int b : {0, 8}) for (int w = 0; w < 12; ++w) { const long long* d = &hd[(b * 12 + w) * 4]; printf("           wg %d wave %2d: %lld  %lld  %lld\n", b, w, d[0], d[1], d[2]); }
            }
        }
        const double gf = 2.0 * NTp * Kp * Np / 1e9, mb = ((double)11 * m * NTp * 8) / 1e6;
        printf("         k_fwd_fused + reduce %.1f us | k_fwd_res + k_pfinish %.1f us (kernel alone %.1f us = %.1f TF = %.1f %% of 78.6, %.2f TB/s)\n",
               t_old * 1e3, t_new * 1e3, t_k * 1e3, gf / t_k, 100. * gf / t_k / 78.6, mb / t_k / 1e3);
    }
    {   // ---- gradient: k_bgemm64 (tiled, fused Z build) against k_grad_res, weights = the residuals dP of the cost run above
        Bgemm64Args g;
        g.EI = EI; g.mI = m; g.phiI = phiI; g.Zq64 = nullptr; g.Zq32 = nullptr; g.mO = m; g.phiO = phiO; g.EL = EL; g.EL_lstride = (size_t)m * NTp; g.dPz = dP1;
        g.w = nullptr; g.w_lstride = 0; g.Kp = Kp; g.Np = Np; g.NTp = NTp; g.L = 1; g.env64 = 1;
        GradResArgs gr{EI, phiI, phiO, EL, (size_t)m * NTp, dP1, NTp, NTp / 32};
        launch_bgemm64(c, g, G0); CK(hipStreamSynchronize(c->stream));
        launch_grad_res(c, gr, G1); CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
        std::vector<double> a((size_t)Kp * Np), b2((size_t)Kp * Np);
        CK(hipMemcpy(a.data(), G1, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), G0, b2.size() * 8, hipMemcpyDeviceToHost));
        printf("gradient: G max rel diff %.3e (|G|max %.4f)\n", maxrel(a, b2, a.size()), [&]() { double mx = 0; for (double x : b2) mx = fmax(mx, fabs(x)); return mx; }());
        launch_grad_res(c, gr, G1); CK(hipStreamSynchronize(c->stream));
        std::vector<double> a2((size_t)Kp * Np); CK(hipMemcpy(a2.data(), G1, a2.size() * 8, hipMemcpyDeviceToHost));
        size_t nd = 0; for (size_t i = 0; i < a2.size(); ++i) nd += a2[i] != a[i];
        printf("         repeat run: %zu differing entries\n", nd);
        const float t_old = time_it([&]() { launch_bgemm64(c, g, G0); });
        const float t_new = time_it([&]() { launch_grad_res(c, gr, G1); });
        const double gf = 2.0 * NTp * Kp * Np / 1e9;
        printf("         k_bgemm64 + slab reduce %.1f us | k_grad_res + slab reduce %.1f us (%.1f TF on the algorithmic flops)\n", t_old * 1e3, t_new * 1e3, gf / t_new);
        const size_t lds = sizeof(double) * GR_LDS_DOUBLES;
        gr.slab = (double*)c->slab;
#define TRYG(PS, PK, ABL, what) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_res<PS, PK, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            const float t_ = time_it([&]() { hipLaunchKernelGGL((k_grad_res<PS, PK, ABL>), dim3(256), dim3(768), lds, c->stream, gr); }); printf("         %-64s %.1f us\n", what, t_ * 1e3); }
        TRYG(0, 1, 0, "kernel alone, GEMM waves never pause:") TRYG(6, 2, 0, "pause 384 cycles every 8 MFMAs (default):") TRYG(4, 2, 0, "256 cycles every 8 MFMAs:") TRYG(6, 3, 0, "384 cycles every 12 MFMAs:")
        TRYG(4, 1, 0, "256 cycles every 4 MFMAs:") TRYG(8, 2, 0, "512 cycles every 8 MFMAs:") TRYG(8, 4, 0, "512 cycles every 16 MFMAs:")
        TRYG(0, 1, 1, "GEMM role alone (no environment loads):") TRYG(0, 1, 2, "streaming role alone (no MFMAs):")
        // ---- k_grad_q (uniform waves, groups of four workgroups)
        CK(hipMemset(G1, 0, (size_t)Kp * Np * 8));
        if (launch_grad_q(c, gr, G1)) return 1;
        CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
        CK(hipMemcpy(a.data(), G1, a.size() * 8, hipMemcpyDeviceToHost));
        printf("k_grad_q: G max rel diff %.3e against k_bgemm64\n", maxrel(a, b2, a.size()));
        launch_grad_q(c, gr, G1); CK(hipStreamSynchronize(c->stream));
        CK(hipMemcpy(a2.data(), G1, a2.size() * 8, hipMemcpyDeviceToHost));
        nd = 0; for (size_t i = 0; i < a2.size(); ++i) nd += a2[i] != a[i];
        printf("         repeat run: %zu differing entries\n", nd);
        const float t_q = time_it([&]() { launch_grad_q(c, gr, G1); });
        printf("         k_grad_q + slab reduce %.1f us (%.1f TF on the algorithmic flops)\n", t_q * 1e3, gf / t_q);
        {
            const size_t ldsq = GQ_LDS_BYTES;
            GradResArgs gq = gr; gq.ntiles = NTp / GQ_T;
#define TRYQ(ABL, what) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_q<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq); \
            const float t_ = time_it([&]() { hipLaunchKernelGGL((k_grad_q<ABL>), dim3(c->res_grid > 0 ? c->res_grid : 256), dim3(512), ldsq, c->stream, gq); }); printf("         %-64s %.1f us\n", what, t_ * 1e3); }
            TRYQ(0, "k_grad_q alone:") TRYQ(1, "no DMA (LDS + MFMA only):") TRYQ(2, "no MFMAs (DMA + build only):") TRYQ(3, "no MFMAs, no Label-free pieces:") TRYQ(4, "no MFMAs, no Label-carrying pieces:") TRYQ(5, "DMA only (no build reads):")
        }
        {   // ---- k_grad_h (probe only): k_grad_q's GEMM waves without a single VMEM instruction + four mover waves
            const size_t ldsh = GH_LDS_BYTES;
            GradResArgs gh = gr; gh.ntiles = NTp / 32; gh.slab = (double*)c->slab;
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_h<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh);
            auto run_h = [&]() { hipLaunchKernelGGL((k_grad_h<0>), dim3(256), dim3(768), ldsh, c->stream, gh); launch_slab_reduce64(c, (const double*)c->slab, G1, (size_t)Kp * Np, 64); };
            CK(hipMemset(G1, 0, (size_t)Kp * Np * 8));
            run_h(); CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
            CK(hipMemcpy(a.data(), G1, a.size() * 8, hipMemcpyDeviceToHost));
            printf("k_grad_h: G max rel diff %.3e against k_bgemm64\n", maxrel(a, b2, a.size()));
            run_h(); CK(hipStreamSynchronize(c->stream));
            CK(hipMemcpy(a2.data(), G1, a2.size() * 8, hipMemcpyDeviceToHost));
            nd = 0; for (size_t i = 0; i < a2.size(); ++i) nd += a2[i] != a[i];
            printf("         repeat run: %zu differing entries\n", nd);
            printf("         k_grad_h + slab reduce %.1f us\n", time_it(run_h) * 1e3);
#define TRYH(ABL, what) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_h<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh); \
            const float t_ = time_it([&]() { hipLaunchKernelGGL((k_grad_h<ABL>), dim3(c->res_grid > 0 ? c->res_grid : 256), dim3(768), ldsh, c->stream, gh); }); printf("         %-64s %.1f us\n", what, t_ * 1e3); }
            TRYH(0, "k_grad_h alone:") TRYH(1, "no loads (LDS + MFMA only; the operands are zeros):") TRYH(10, "no loads, LDS filled with full-mantissa numbers:") TRYH(2, "no MFMAs (movers + DMA + build):")
            TRYH(6, "movers do not write to LDS:") TRYH(7, "no DMA pieces (movers' ring on):") TRYH(8, "ring off (DMA pieces on):")
        }
    }
    return 0;
}
