// kernels_grad.hip -- k_grad_quad: the gradient GEMM dP*dag(t.v) (fixedL.cc:379,418) at m = 120, fp64 storage, Label index on an environment:
//
//   G[2a + s][2q + t] = sum_n  E_n[a] phiI_n[s]  *  phiO_n[t] Z_n[q],      Z_n[q] = sum_l EL_n[l][q] dP_n[l]
//
// (E the Label-free, EL the Label-carrying environment; the dense t.v of the reference is never formed).  The reduction runs over the
// IMAGES, so both MFMA operands change every k-step and neither can live in registers the way the bond matrix does in k_fwd_res; what can
// live in registers is the RESULT.  The shape follows the counters of round 5 (profiles/r05_grad_pmc_by_variant.txt): every earlier form
// kept two MFMA-issuing waves per SIMD (128 accumulator registers per wave, or 157 KB of LDS) and sat at 3/4 of the matrix pipe with
// nothing else running; a wave that only loads or only does VALU work is starved by the MFMA waves of its SIMD.  Here
//   * a QUAD of workgroups (one XCD: blocks b, b + 8, b + 16, b + 24) owns the 240 x 240 accumulators for one slab of images; workgroup h
//     of the quad owns the output links q in [30 h, 30 h + 30), i.e. 60 of the 240 columns (4 column tiles, the last one 3/4 full), and
//     therefore streams only ITS quarter of the Label-carrying environment -- the 577 MB stream is read exactly once;
//   * 12 UNIFORM waves per workgroup = THREE MFMA-issuing waves per SIMD, 5 row tiles x 1 column tile = 40 accumulator registers each;
//     every wave also does 1/12 of the staging (loads into registers one stage ahead, LDS writes, the ten FMAs per element of Z), and the
//     three waves of a SIMD do that part after DIFFERENT MFMA blocks of a stage, so that at any time two of them feed the matrix pipe;
//   * the site features are applied to the FRAGMENTS (one multiply per fragment element), so the Label-free rows go to LDS raw and are
//     staged once per workgroup for both values of s: 46 KB of LDS per 32-image stage, two stages, ONE barrier per stage;
//   * row stride 36 doubles: the 16-byte fragment reads of 16 consecutive rows fall on 16 different bank quads (stride 34, as in
//     k_bgemm64, makes two lanes of each ds_read_b128 group collide -- its 12 % bank-conflict time).
// Deterministic: every element is accumulated over its slab's images in a fixed order; the slabs are summed in slab order by the
// consumer (the CG vector kernel, or k_slab_reduce64), exactly like k_bgemm64's.
#include "tnml_internal.h"

typedef double f64x4g __attribute__((ext_vector_type(4)));

#define GQ_TI 32                       // images per stage
#define GQ_RS 36                       // doubles between staged rows
#define GQ_Q 30                        // output links per workgroup
#define GQ_E_D (120 * GQ_RS)           // doubles per stage buffer: Label-free rows
#define GQ_Z_D (31 * GQ_RS)            // Z rows of this workgroup + one row of zeros (the padding columns of the last column tile)
#define GQ_P_D (4 * GQ_TI)             // phiI[0..1], phiO[0..1]
#define GQ_D_D (TNML_NL * GQ_TI)       // dP rows
#define GQ_LDS_DOUBLES (2 * (GQ_E_D + GQ_Z_D + GQ_P_D + GQ_D_D))

struct GradQuadArgs {
    const double* EI; const double* phiI; const double* phiO;      // [120][NTp], [2][NTp], [2][NTp]
    const double* EL; size_t EL_lstride;                          // [10][120][NTp]
    const double* dP;                                             // [10][NTp]
    int NTp;
    double* slab;                                                 // [ngroups][240][240], M-layout
    int ngroups, per, nchunks;                                    // quads, 32-image chunks per quad, chunks in all
};

// workgroup barrier without the vmcnt(0) of __syncthreads(): the loads for the stage after next stay in flight across it
static __device__ __forceinline__ void gq_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ABL (probe builds): 1 = no loads of the Label-carrying environment (compute side alone), 2 = no MFMAs (stream side alone)
template <int ABL>
__global__ __launch_bounds__(768) void k_grad_quad(GradQuadArgs A) {
    extern __shared__ __attribute__((aligned(16))) double gq_lds[];
    double* const Es = gq_lds;                         // [2][120][36]
    double* const Zs = Es + 2 * GQ_E_D;                // [2][31][36]
    double* const Ph = Zs + 2 * GQ_Z_D;                // [2][4][32]
    double* const Ds = Ph + 2 * GQ_P_D;                // [2][10][32]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int h = (b >> 3) & 3, grp = (b & 7) + 8 * (b >> 5);
    if (grp >= A.ngroups) return;
    const int NTp = A.NTp;
    const int c0 = grp * A.per;
    const int nch = min(A.per, A.nchunks - c0);        // >= 1 (the launcher sizes ngroups)
    auto n_of = [&](int k) { return (size_t)(c0 + (k < nch ? k : nch - 1)) * GQ_TI; };      // (beyond the slab: the last chunk again, never consumed)

    // ---- staging pieces: 16 bytes per lane, 4 rows x 32 images per wave instruction.  Every wave moves the Label-free rows 4w .. 4w + 3
    //      and 4w + 48 .. 4w + 51 of the NEXT stage; its third piece is, by wave: 0..5 the rows 4w + 96 .., 6 the four feature rows (both
    //      of the next stage), 7..9 the dP rows 4 (w - 7) .. of the stage AFTER next, 10..11 nothing
    const int rho = lane >> 4, x2 = 2 * (lane & 15);
    double2 pc[3];
    // (addresses as uniform base + 32-bit lane offset: one SGPR pair per request instead of a VGPR pair)
    const unsigned pvoff = (unsigned)(((size_t)rho * NTp + x2) * sizeof(double));
    auto ld16 = [&](const double* ubase, unsigned voff) { return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(ubase) + voff); };
    // third piece: uniform base (at image 0) + lane offset, its LDS place (doubles) = dst2 + (buffer or slot) * st2
    const double* ub2; unsigned voff2 = pvoff;
    int dst2, st2;
    bool ok2 = true, dp2 = false;
    if (w < 6)       { ub2 = A.EI + (size_t)(4 * (w + 24)) * NTp; dst2 = (4 * (w + 24) + rho) * GQ_RS + x2; st2 = GQ_E_D; }
    else if (w == 6) {                                 // rows phiI[0], phiI[1], phiO[0], phiO[1] (two sites of one feature array: the second base as an offset from the first)
        const double* f1 = A.phiO - 2 * (size_t)NTp;
        ub2 = A.phiI < f1 ? A.phiI : f1;
        voff2 = pvoff + (unsigned)(((rho < 2 ? A.phiI : f1) - ub2) * (ptrdiff_t)sizeof(double));
        dst2 = 2 * (GQ_E_D + GQ_Z_D) + rho * GQ_TI + x2; st2 = GQ_P_D;
    } else {
        const int row = 4 * (w - 7) + rho;
        ok2 = w < 10 && row < TNML_NL; dp2 = true;
        ub2 = A.dP + (size_t)(w < 10 ? 4 * (w - 7) : 0) * NTp;
        dst2 = 2 * (GQ_E_D + GQ_Z_D + GQ_P_D) + row * GQ_TI + x2; st2 = GQ_D_D;
    }
    const int dst0 = (4 * w + rho) * GQ_RS + x2;
    auto piece_load = [&](size_t nE, size_t nD) {
        pc[0] = ld16(A.EI + (size_t)(4 * w) * NTp + nE, pvoff);
        pc[1] = ld16(A.EI + (size_t)(4 * (w + 12)) * NTp + nE, pvoff);
        pc[2] = ok2 ? ld16(ub2 + (dp2 ? nD : nE), voff2) : make_double2(0., 0.);
    };
    auto piece_store = [&](int bufE, int slotD) {
        *reinterpret_cast<double2*>(Es + bufE * GQ_E_D + dst0) = pc[0];
        *reinterpret_cast<double2*>(Es + bufE * GQ_E_D + dst0 + 48 * GQ_RS) = pc[1];
        if (ok2) *reinterpret_cast<double2*>(gq_lds + dst2 + (dp2 ? slotD : bufE) * st2) = pc[2];
    };
    // ---- Z units: one output link x 32 images per wave instruction, the ten labels dealt to the two lane halves (labels 5 qs .. 5 qs + 4:
    //      five loads per lane and unit, the halves added by one cross-half exchange); unit ids w, w + 12, w + 24 (< 30)
    const int img = lane & 31, qs = lane >> 5;
    double el[3][5];
    const unsigned elvoff = (unsigned)(((size_t)(5 * qs) * A.EL_lstride + img) * sizeof(double));      // < 4 GB (checked by the launcher)
    const bool u3 = w < 6;                              // this wave has a third unit
    auto el_load = [&](size_t n0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double* ub = A.EL + (size_t)(GQ_Q * h + w + 12 * j) * NTp + n0;       // uniform
#pragma unroll
            for (int l = 0; l < 5; ++l)
                el[j][l] = ((j < 2 || u3) && ABL != 1) ? __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(ub + (size_t)l * A.EL_lstride) + elvoff)) : 1.0;
        }
    };
    auto z_build = [&](int bufZ, int slotD) {
        const double* dp = Ds + slotD * GQ_D_D + 5 * qs * GQ_TI + img;
        double d[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) d[l] = dp[l * GQ_TI];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double z = 0.;
#pragma unroll
            for (int l = 0; l < 5; ++l) z = fma(el[j][l], d[l], z);
            const double zo = __shfl_xor(z, 32);
            if (qs == 0 && (j < 2 || u3)) Zs[bufZ * GQ_Z_D + (w + 12 * j) * GQ_RS + img] = z + zo;       // (labels 0..4) + (labels 5..9)
        }
    };

    // ---- MFMA roles: wave (rg, J) = row tiles 5 rg .. 5 rg + 4 of the M-layout rows 2a + s, column tile J of this workgroup's 60 columns
    //      2q + t.  Rows 2a, 2a + 1 share the staged row a (two lanes read the same 16 bytes: a broadcast), so one feature fragment
    //      serves all five row tiles and a row tile is 8 consecutive staged rows: offsets differ by compile-time constants.
    const int rg = w >> 2, J = w & 3;
    const int li = lane & 15, g = lane >> 4;
    const int eoff = (8 * 5 * rg + (li >> 1)) * GQ_RS + 2 * g;             // + 8 r GQ_RS for row tile r
    const int poff = (li & 1) * GQ_TI + 2 * g;
    const int cc = 16 * J + li;                                             // column 60 h + cc
    const int zoff = (cc < 2 * GQ_Q ? (cc >> 1) : GQ_Q) * GQ_RS + 2 * g;    // padding columns read the row of zeros
    const int ooff = (2 + (cc & 1)) * GQ_TI + 2 * g;
    f64x4g acc[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) acc[r] = f64x4g{0., 0., 0., 0.};

    // ---- prologue
    if (tid < 2 * GQ_RS) Zs[(tid / GQ_RS) * GQ_Z_D + GQ_Q * GQ_RS + (tid % GQ_RS)] = 0.;
    piece_load(n_of(0), n_of(0));
    piece_store(0, 0);
    piece_load(n_of(0), n_of(1));                      // (the first stage's rows once more, harmlessly: what is wanted is dP of the second stage)
    piece_store(0, 1);
    el_load(n_of(0));
    gq_barrier();
    z_build(0, 0);
    piece_load(n_of(1), n_of(2));
    el_load(n_of(1));
    gq_barrier();

    for (int k = 0; k < nch; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
        const double* Eb = Es + cur * GQ_E_D;
        const double* Zb = Zs + cur * GQ_Z_D;
        const double* Pb = Ph + cur * GQ_P_D;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int kk = 8 * blk;
            double a0[5], a1[5];
            const double2 p = *reinterpret_cast<const double2*>(Pb + poff + kk);
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const double2 e = *reinterpret_cast<const double2*>(Eb + eoff + 8 * r * GQ_RS + kk);
                a0[r] = e.x * p.x; a1[r] = e.y * p.y;
            }
            const double2 z = *reinterpret_cast<const double2*>(Zb + zoff + kk);
            const double2 o = *reinterpret_cast<const double2*>(Pb + ooff + kk);
            const double b0 = z.x * o.x, b1 = z.y * o.y;
            if (ABL != 2) {
#pragma unroll
                for (int r = 0; r < 5; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[r], b0, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 5; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[r], b1, acc[r], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 5; ++r) acc[r][0] += a0[r] * b0 + a1[r] * b1;
            }
            // the staging share of this wave, after the block whose number is its place on the SIMD (waves w, w + 4, w + 8 share one):
            // what was requested one stage ago goes to the other buffers, the next requests go out
            __builtin_amdgcn_sched_barrier(0);       // (no fragment reads of later blocks hoisted over this one's MFMAs: they would cost their registers for the whole stage)
            if (blk == rg) {
                piece_store(nxt, cur);                 // E / features of chunk k + 1 -> buffers nxt; dP of chunk k + 2 -> slot (k + 2) & 1 = cur
                z_build(nxt, nxt);                     // Z of chunk k + 1 from dP slot (k + 1) & 1
                piece_load(n_of(k + 2), n_of(k + 3));
                el_load(n_of(k + 2));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        gq_barrier();
    }

    // ---- epilogue: the quad's partial G, M-layout rows 2a + s, columns 2q + t
    double* out = A.slab + (size_t)grp * 240 * 240;
    if (cc < 2 * GQ_Q) {
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[(size_t)(16 * (5 * rg + r) + g + 4 * e) * 240 + 2 * GQ_Q * h + cc] = acc[r][e];
    }
}

bool grad_quad_applies(const tnml_ctx* c, const Bgemm64Args& a) {
    if (!c->grad_quad || !a.EL || !a.env64 || a.L != 1 || a.w) return false;
    if (a.Kp != 240 || a.Np != 240 || a.mI != 120 || a.mO != 120 || a.NTp % GQ_TI) return false;
    if (c->grad_quad == 1 && a.NTp < 4096) return false;
    if ((size_t)TNML_NL * a.EL_lstride * sizeof(double) >= ((size_t)1 << 32)) return false;      // 32-bit lane offsets
    return c->slab_bytes >= (size_t)64 * 240 * 240 * sizeof(double);
}

int launch_grad_quad(tnml_ctx* c, const Bgemm64Args& a, double* G) {
    if (!c->cu_count) { hipDeviceProp_t pr; c->cu_count = hipGetDeviceProperties(&pr, c->cfg.device) == hipSuccess ? pr.multiProcessorCount : 256; }
    GradQuadArgs K;
    K.EI = static_cast<const double*>(a.EI); K.phiI = static_cast<const double*>(a.phiI); K.phiO = static_cast<const double*>(a.phiO);
    K.EL = static_cast<const double*>(a.EL); K.EL_lstride = a.EL_lstride; K.dP = a.dPz; K.NTp = a.NTp;
    K.slab = static_cast<double*>(c->slab);
    K.nchunks = a.NTp / GQ_TI;
    int quads = c->cu_count / 4;                                   // one workgroup per CU
    const int cap = (int)(c->slab_bytes / ((size_t)240 * 240 * sizeof(double)));
    if (quads > cap) quads = cap;
    if (c->bgemm_wgs > 0 && c->bgemm_wgs / 4 < quads) quads = c->bgemm_wgs / 4 > 0 ? c->bgemm_wgs / 4 : 1;      // test knob: fewer quads -> more stages each
    if (quads > K.nchunks) quads = K.nchunks;
    K.per = (K.nchunks + quads - 1) / quads;
    K.ngroups = (K.nchunks + K.per - 1) / K.per;
    const size_t lds = sizeof(double) * GQ_LDS_DOUBLES;
    if (!c->attr_gq) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return tnml_fail(c, "grad_quad: cannot reserve %zu bytes of LDS", lds);
        c->attr_gq = true;
    }
    const int grid = 32 * ((K.ngroups + 7) / 8);                 // blocks b, b + 8, b + 16, b + 24 of a run of 32 = one quad on one XCD
    {
        ProfScope ps(c, KC_BGEMM);
        // (grad_quad = 3 / 4: the ablations of the record under profiles/ -- compute side alone / stream side alone; wrong results by construction)
        if (c->grad_quad == 3)      hipLaunchKernelGGL(k_grad_quad<1>, dim3(grid), dim3(768), lds, c->stream, K);
        else if (c->grad_quad == 4) hipLaunchKernelGGL(k_grad_quad<2>, dim3(grid), dim3(768), lds, c->stream, K);
        else                        hipLaunchKernelGGL(k_grad_quad<0>, dim3(grid), dim3(768), lds, c->stream, K);
    }
    const size_t n = (size_t)240 * 240;
    if (c->defer_slab) c->slab_pending = K.ngroups;            // the CG vector kernel that consumes G sums the slabs itself (slab order: the same bits)
    else {
        ProfScope ps(c, KC_SLABRED);
        launch_slab_reduce64(c, static_cast<const double*>(c->slab), G, n, K.ngroups);
    }
    HIPCK(c, hipGetLastError());
    return 0;
}
