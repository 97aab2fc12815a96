// Where does the FIXED part of a big launch go?  k_grad_quad takes 14 us + 2.4 us per 1 000 images (rocprof: 32.5 us at 7 500 images, 159.7 at
// 60 000), k_fwd_res 12 + 2.15, k_shift_res 18 + 9.1.  This probe stamps every workgroup of k_grad_quad (ABL 5: entry, prologue done, loop
// done, stores done; 100 MHz wall clock) and times an empty kernel of the same launch shape beside it.
//   probe_fixed [NT]
#include "../../tnml_amd/csrc/kernels_grad.hip"
#include "../../tnml_amd/csrc/kernels_gemm.hip"
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include <algorithm>
int tnml_fail(tnml_ctx*, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return 1; }
void prof_begin(tnml_ctx*, int, hipEvent_t*, hipStream_t) {}
void prof_end(tnml_ctx*, int, hipEvent_t, hipStream_t) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc((void**)&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static hipStream_t g_st;
template <typename F> static float time_it(F f, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipStreamSynchronize(g_st);
    hipEventRecord(e0, g_st); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1, g_st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / reps;
}
__global__ __launch_bounds__(1024) void k_empty(double* p) {
    extern __shared__ double e_lds[];
    if (p && threadIdx.x == 0 && blockIdx.x == 100000) p[0] = e_lds[0];
}
// the epilogue alone: every workgroup writes its 128 x 64 share of a slab (what k_grad_quad's last lines do), nothing else
__global__ __launch_bounds__(1024) void k_store_only(double* slab, int Kp, int Np, int ngroups) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
    const int h = (b >> 3) & 3, grp = (b & 7) + 8 * (b >> 5);
    if (grp >= ngroups) return;
    const int rgp = w >> 2, J = w & 3, li = lane & 15, g = lane >> 4, sI = rgp >> 1, a0 = 64 * (rgp & 1), cc = 16 * J + li;
    double* out = slab + (size_t)grp * Kp * Np;
    const int col = 64 * h + cc;
    if (col < Np)
        for (int r = 0; r < 4; ++r)
            for (int e4 = 0; e4 < 4; ++e4) {
                const int row = 2 * (a0 + 16 * r + g + 4 * e4) + sI;
                if (row < Kp) out[(size_t)row * Np + col] = (double)(row + col);
            }
}
int main(int argc, char** argv) {
    const int NT = argc > 1 ? atoi(argv[1]) : 7500;
    const int NTp = (NT + 255) / 256 * 256, m = 120, Kp = 240, Np = 240;
    tnml_ctx ctx; tnml_ctx* c = &ctx;
    c->cfg.device = 0; c->cfg.dtype = TNML_F64; c->cfg.mode = TNML_MODE_FIXEDL; c->NTp = NTp; c->NT = NT;
    c->slab_bytes = (size_t)128 * Kp * Kp * 8;
    c->defer_slab = true;
    double* EI = dalloc<double>((size_t)m * NTp); double* EL = dalloc<double>((size_t)10 * m * NTp);
    double* phi = dalloc<double>(4 * (size_t)NTp);
    double* dP = dalloc<double>(10 * (size_t)NTp);
    c->slab = dalloc<char>(c->slab_bytes);
    double* G = dalloc<double>((size_t)Kp * Np);
    long long* dbg = dalloc<long long>(4 * 512);
    {
        srand(7);
        auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
        std::vector<double> h((size_t)10 * m * NTp);
        for (auto& x : h) x = rnd();
        CK(hipMemcpy(EL, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(EI, h.data(), (size_t)m * NTp * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(phi, h.data(), 4 * (size_t)NTp * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dP, h.data(), 10 * (size_t)NTp * 8, hipMemcpyHostToDevice));
    }
    CK(hipStreamCreate(&c->stream)); g_st = c->stream;
    Bgemm64Args a{};
    a.EI = EI; a.mI = m; a.phiI = phi; a.mO = m; a.phiO = phi + 2 * (size_t)NTp; a.EL = EL; a.EL_lstride = (size_t)m * NTp; a.dPz = dP;
    a.Kp = Kp; a.Np = Np; a.NTp = NTp; a.L = 1; a.env64 = 1;
    c->grad_quad = 2;
    if (!grad_quad_applies(c, a)) { printf("grad_quad does not apply\n"); return 1; }
    const float t_full = time_it([&]() { launch_grad_quad(c, a, G); });
    // the same launch by hand (ABL 5: stamps)
    GradQuadArgs K;
    K.EI = EI; K.phiI = phi; K.phiO = phi + 2 * (size_t)NTp; K.EL = EL; K.EL_lstride = a.EL_lstride; K.dP = dP; K.NTp = NTp;
    K.mI = m; K.mO = m; K.Kp = Kp; K.Np = Np; K.slab = (double*)c->slab; K.nchunks = NTp / GQ_TI;
    int quads = 64; if (quads > K.nchunks) quads = K.nchunks;
    K.per = (K.nchunks + quads - 1) / quads; K.ngroups = (K.nchunks + K.per - 1) / K.per; K.dbg = dbg;
    const size_t lds = sizeof(double) * GQ_LDS_DOUBLES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 32 * ((K.ngroups + 7) / 8);
    const float t_st = time_it([&]() { hipLaunchKernelGGL(k_grad_quad<5>, dim3(grid), dim3(1024), lds, g_st, K); });
    CK(hipStreamSynchronize(g_st)); CK(hipGetLastError());
    std::vector<long long> h(4 * (size_t)grid);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    printf("images %d (NTp %d): %d quads x %d chunks, grid %d\n", NT, NTp, K.ngroups, K.per, grid);
    printf("  k_grad_quad<0> through launch_grad_quad %.2f us per launch (events, 20 back to back); <5> with stamps %.2f us\n", t_full, t_st);
    {
        long long e0 = 1ll << 62, e1 = 0, x1 = 0, x0 = 1ll << 62; double pro = 0, loop = 0, epi = 0; int nw = 0;
        std::vector<double> ent, fin;
        for (int b = 0; b < grid; ++b) {
            const int grp = (b & 7) + 8 * (b >> 5);
            if (grp >= K.ngroups) continue;
            const long long* d = &h[4 * (size_t)b];
            e0 = std::min(e0, d[0]); e1 = std::max(e1, d[0]); x1 = std::max(x1, d[3]); x0 = std::min(x0, d[3]);
            pro += d[1] - d[0]; loop += d[2] - d[1]; epi += d[3] - d[2]; ++nw;
        }
        printf("  stamps (10 ns ticks -> us): first entry .. last entry %.2f us; first entry .. last exit %.2f us; first exit .. last exit %.2f us\n",
               0.01 * (e1 - e0), 0.01 * (x1 - e0), 0.01 * (x1 - x0));
        printf("  mean per workgroup: prologue %.2f us, loop %.2f us (%.2f per chunk), stores + wait %.2f us\n", 0.01 * pro / nw, 0.01 * loop / nw, 0.01 * loop / nw / K.per, 0.01 * epi / nw);
        for (int b : {0, 8, 16, 24, 1, 33, grid - 32, grid - 1}) {
            const long long* d = &h[4 * (size_t)b];
            printf("    wg %3d: entry +%.2f, prologue %.2f, loop %.2f, stores %.2f, exit +%.2f\n", b, 0.01 * (d[0] - e0), 0.01 * (d[1] - d[0]), 0.01 * (d[2] - d[1]), 0.01 * (d[3] - d[2]), 0.01 * (d[3] - e0));
        }
    }
    {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const float t6 = time_it([&]() { hipLaunchKernelGGL(k_grad_quad<6>, dim3(grid), dim3(1024), lds, g_st, K); });
        CK(hipStreamSynchronize(g_st));
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double pro = 0, loop = 0, epi = 0; int nw = 0;
        for (int b = 0; b < grid; ++b) {
            if ((b & 7) + 8 * (b >> 5) >= K.ngroups) continue;
            const long long* d = &h[4 * (size_t)b];
            pro += d[1] - d[0]; loop += d[2] - d[1]; epi += d[3] - d[2]; ++nw;
        }
        printf("  second chunk requested AFTER the first has landed: %.2f us per launch; prologue %.2f us, loop %.2f us, stores %.2f us\n", t6, 0.01 * pro / nw, 0.01 * loop / nw, 0.01 * epi / nw);
    }
    const float t_e = time_it([&]() { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(1024), lds, g_st, (double*)nullptr); });
    const float t_e0 = time_it([&]() { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, g_st, (double*)nullptr); });
    const float t_s = time_it([&]() { hipLaunchKernelGGL(k_store_only, dim3(grid), dim3(1024), 0, g_st, (double*)c->slab, Kp, Np, K.ngroups); });
    printf("  empty kernel, same shape (1024 lanes, %zu B LDS) %.2f us per launch; 64 lanes, no LDS %.2f us; the slab stores alone (%.1f MB) %.2f us\n",
           lds, t_e, t_e0, 1e-6 * K.ngroups * Kp * Np * 8, t_s);
    // one chunk per quad at most: what is left of the launch when the loop is one stage long
    {
        GradQuadArgs K1 = K; K1.nchunks = std::min(K.nchunks, K.ngroups); K1.per = 1; K1.ngroups = K1.nchunks;
        const float t1 = time_it([&]() { hipLaunchKernelGGL(k_grad_quad<0>, dim3(grid), dim3(1024), lds, g_st, K1); });
        printf("  one chunk per quad: %.2f us per launch\n", t1);
    }
    return 0;
}
