// kernels_grad.hip -- k_grad_quad: the gradient GEMM dP*dag(t.v) (fixedL.cc:379,418) for bond dimensions up to 128 (BASELINE config 3: m = 120), fp64
// storage, Label index on an environment:
//
//   G[2a + s][2q + t] = sum_n  E_n[a] * ( phiI_n[s] phiO_n[t] Z_n[q] ),      Z_n[q] = sum_l EL_n[l][q] dP_n[l]
//
// (E the Label-free, EL the Label-carrying environment; the dense t.v of the reference is never formed).  The reduction runs over the
// IMAGES, so both MFMA operands change every k-step and neither can live in registers the way the bond matrix does in k_fwd_res; what can
// live in registers is the RESULT.  The shape follows two measurements:
//   (1) profiles/r05_grad_pmc_by_variant.txt: every earlier form kept two MFMA-issuing waves per SIMD (128 accumulator registers per
//       wave, or 157 KB of LDS) and sat at 3/4 of the matrix pipe with nothing else running; a wave that only loads or only does VALU work
//       is starved by the MFMA waves of its SIMD;
//   (2) profiles/r06_grad_quad_v1_first_run.txt: with three MFMA waves per SIMD and BOTH site features applied to the fragments (12
//       multiplies per 10 MFMAs) the compute side alone still reached only 0.61 of the pipe -- a VALU instruction is issued only while
//       no MFMA of any wave waits at the SIMD's issue port, so the waves fall into step: MFMA phases, then VALU phases with the pipe idle.
// Hence:
//   * a QUAD of workgroups (one XCD: blocks b, b + 8, b + 16, b + 24) owns the accumulators of the whole gradient (up to 256 x 256; 240 x 240 at
//     m = 120) for one slab of images; workgroup h of the quad owns the output links q in [32 h, 32 h + 32), i.e. 64 columns 2q + t = 4 column
//     tiles (at m = 120 the last workgroup holds 24 links), and therefore streams only ITS share of the Label-carrying environment -- the
//     577 MB stream is read exactly once;
//   * 16 UNIFORM waves per workgroup = FOUR MFMA-issuing waves per SIMD, 4 row tiles x 1 column tile = 32 accumulator registers each;
//   * the A operand is the RAW Label-free environment: rows are taken s-major (8 row tiles of a = 0..127 per value of s, rows from the bond
//     dimension on zero), so a wave's row tiles share one s and both site features move to the B side as ONE weight w_st[n] = phiI_n[s] phiO_n[t],
//     tabulated per stage: 2 multiplies per 8 MFMAs, and the Label-free rows go to LDS untouched, once for both values of s.  Price: 16 row
//     tiles instead of 15 (6.7 % more MFMAs);
//   * every wave does 1/16 of the staging (loads into registers one stage ahead, LDS writes, the ten FMAs per element of Z), and the four
//     waves of a SIMD do that part after DIFFERENT MFMA blocks of a stage; 46 KB of LDS per 32-image stage, two stages, ONE barrier per stage;
//   * fragments are read one k-step (4 images, ds_read_b64) at a time and ONE K-STEP AHEAD of the MFMAs that use them: the four waves
//     of a SIMD share the pipe round-robin and therefore stay in step -- with the reads of a block at its top (the 16-byte form) all four
//     waited for LDS at the same moments and the pipe idled ~20 % of every block (profiles/r06_grad_quad_steps.txt, "no staging");
//   * row stride 34 doubles: the 8-byte fragment reads of 16 consecutive rows x 2 k-slots fall on 32 different bank pairs.
// Deterministic: every element is accumulated over its slab's images in a fixed order; the slabs are summed in slab order by the
// consumer (the CG vector kernel, or k_slab_reduce64), exactly like k_bgemm64's.
#include "tnml_internal.h"

typedef double f64x4g __attribute__((ext_vector_type(4)));

#define GQ_TI 32                       // images per stage
#ifndef GQ_SLOT
#define GQ_SLOT(rgp) (2 * (rgp))       // the k-step before which the waves of row group rgp do their staging share (tuning builds override it)
#endif
#ifndef GQ_PRIO
#define GQ_PRIO 3                      // issue priority of a wave while it stages
#endif
#define GQ_RS 34                       // doubles between staged rows
#define GQ_Q 32                        // output links per workgroup
#define GQ_E_D_N(NR) (32 * (NR) * GQ_RS) // doubles per stage buffer: Label-free rows 0..mI-1, zeros up to 32 NR - 1 (NR = row tiles per wave: 4 quad form, 2 pair form)
#define GQ_Z_D (GQ_Q * GQ_RS)          // Z rows of this workgroup (zeros for links beyond the bond dimension)
#define GQ_W_D (4 * GQ_RS)             // w[2 s + t][n] = phiI[s][n] phiO[t][n]
#define GQ_LDS_DOUBLES_N(NR) (2 * (GQ_E_D_N(NR) + GQ_Z_D + GQ_W_D))
#define GQ_LDS_DOUBLES GQ_LDS_DOUBLES_N(4)

struct GradQuadArgs {
    const double* EI; const double* phiI; const double* phiO;      // [mI][NTp], [2][NTp], [2][NTp]
    const double* EL; size_t EL_lstride;                          // [10][mO][NTp]
    const double* dP;                                             // [10][NTp]
    int NTp;
    int mI, mO, Kp, Np;                                           // bond dimensions (<= 128) and the padded M-layout extents of G
    double* slab;                                                 // [ngroups][Kp][Np], M-layout
    int ngroups, per, nchunks;                                    // quads, 32-image chunks per quad, chunks in all
    long long* dbg = nullptr;                                     // ABL 5 (tools/probe/probe_fixed.hip): four 100 MHz stamps per workgroup
};

// workgroup barrier without the vmcnt(0) of __syncthreads(): the loads for the stage after next stay in flight across it
static __device__ __forceinline__ void gq_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ABL (the ablations of the record under profiles/): 1 = no loads of the Label-carrying environment (compute side alone),
// 3 = no staging at all inside the loop (the MFMA loop with its fragment reads and the barrier per stage), 5 = the kernel as shipped + time
// stamps of every workgroup (entry, prologue done, loop done, stores done)
// NR = row tiles per wave.  4: the QUAD form above (bond dimensions up to 128 x 128).  2: the PAIR form for bonds up to 64 x 64 (the trained bonds of
// a network with minm = maxm / 2, fixedL.cc:593: m = 59-60 at config 3) -- a pair of workgroups (blocks b, b + 8 of a run of 16) owns the 128 x 128
// tile grid, workgroup h the output links [32 h, 32 h + 32), a wave 2 row tiles x 1 column tile; 64 staged rows instead of 128, everything else
// as in the quad form.  At these sizes the launch is bound by the 320 MB it streams, not by the matrix pipe: half the MFMAs per staged byte.
template <int ABL, int NR = 4>
__global__ __launch_bounds__(1024) void k_grad_quad(GradQuadArgs A) {
    static_assert(NR == 4 || NR == 2, "quad form or pair form");
    constexpr int GQ_E_D = GQ_E_D_N(NR);
    extern __shared__ __attribute__((aligned(16))) double gq_lds[];
    double* const Es = gq_lds;                         // [2][32 NR][34]
    double* const Zs = Es + 2 * GQ_E_D;                // [2][32][34]
    double* const Ws = Zs + 2 * GQ_Z_D;                // [2][4][32]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const long long ts0 = (ABL == 5 || ABL == 6) ? (long long)wall_clock64() : 0;
    const int h = (b >> 3) & (NR - 1), grp = (b & 7) + 8 * (NR == 4 ? b >> 5 : b >> 4);
    if (grp >= A.ngroups) return;
    const int NTp = A.NTp;
    const int c0 = grp * A.per;
    const int nch = min(A.per, A.nchunks - c0);        // >= 1 (the launcher sizes ngroups)
    auto n_of = [&](int k) { return (size_t)(c0 + (k < nch ? k : nch - 1)) * GQ_TI; };      // (beyond the slab: the last chunk again, never consumed)

    // ---- staging pieces: 16 bytes per lane, 4 rows x 32 images per wave instruction (addresses as uniform base + 32-bit lane offset: one
    //      SGPR pair per request instead of a VGPR pair), requested one stage ahead.  Slot 0: the Label-free rows 4w .. 4w + 3; slot 1: the
    //      rows 4w + 64 .. (rows from mI on are written as zeros); slot 2, wave 15 alone: the four feature rows, from which it tabulates the
    //      weights w[2 s + t][n]
    const int rho = lane >> 4, x2 = 2 * (lane & 15);
    double2 pc[3];
    const unsigned pvoff = (unsigned)(((size_t)rho * NTp + x2) * sizeof(double));
    auto ld16 = [&](const double* ubase, unsigned voff) { return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(ubase) + voff); };
    const bool v0 = 4 * w + rho < A.mI, v1 = NR == 4 && 64 + 4 * w + rho < A.mI;     // this lane's rows exist
    // (wave 15: rows phiI[0], phiI[1], phiO[0], phiO[1] -- two sites of one feature array: the second base as an offset from the first)
    const double* fO = A.phiO - 2 * (size_t)NTp;
    const double* fb = A.phiI < fO ? A.phiI : fO;
    const unsigned fvoff = pvoff + (unsigned)(((rho < 2 ? A.phiI : fO) - fb) * (ptrdiff_t)sizeof(double));
    // (the lane offsets pass through an empty asm where they are used: their zero extension then sits next to the load and folds into its
    // address mode -- hoisted out of the loop it becomes a 64-bit VGPR add per request, 30 registers of addresses alive across the MFMAs)
    auto piece_load = [&](size_t nE, double2 (&pc)[3]) {
        unsigned vo = pvoff, vf = fvoff;
        asm volatile("" : "+v"(vo), "+v"(vf));
        pc[0] = v0 ? ld16(A.EI + (size_t)(4 * w) * NTp + nE, vo) : make_double2(0., 0.);
        if (NR == 4) pc[1] = v1 ? ld16(A.EI + (size_t)(4 * (w + 16)) * NTp + nE, vo) : make_double2(0., 0.);
        if (w == 15) pc[2] = ld16(fb + nE, vf);
    };
    const int dst0 = (4 * w + rho) * GQ_RS + x2;
    auto piece_store = [&](int bufE, const double2 (&pc)[3]) {
        *reinterpret_cast<double2*>(Es + bufE * GQ_E_D + dst0) = pc[0];
        if (NR == 4) *reinterpret_cast<double2*>(Es + bufE * GQ_E_D + dst0 + 64 * GQ_RS) = pc[1];
        if (w == 15) {                                 // lane (2 s + t, image pair): phiI[s] from lane row s, phiO[t] from lane row 2 + t
            const int lI = 16 * (rho >> 1) + (lane & 15), lO = 16 * (2 + (rho & 1)) + (lane & 15);
            const double ix = __shfl(pc[2].x, lI), iy = __shfl(pc[2].y, lI), ox = __shfl(pc[2].x, lO), oy = __shfl(pc[2].y, lO);
            *reinterpret_cast<double2*>(Ws + bufE * GQ_W_D + rho * GQ_RS + x2) = make_double2(ix * ox, iy * oy);
        }
    };
    // ---- Z units: one output link x 32 images per wave instruction, the ten labels dealt to the two lane halves (labels 5 qs .. 5 qs + 4:
    //      five loads per lane and unit + the five dP values of the lane's image, the halves added by one cross-half exchange); unit ids
    //      w, w + 16 = the links 32 h + w, 32 h + w + 16 (links from mO on: zeros, nothing loaded)
    const int img = lane & 31, qs = lane >> 5;
    double el[2][5], dpv[5];
    const unsigned elvoff = (unsigned)(((size_t)(5 * qs) * A.EL_lstride + img) * sizeof(double));      // < 4 GB (checked by the launcher)
    const unsigned dpvoff = (unsigned)(((size_t)(5 * qs) * NTp + img) * sizeof(double));
    const bool uv[2] = {GQ_Q * h + w < A.mO, GQ_Q * h + w + 16 < A.mO};                                // uniform
    auto el_load = [&](size_t n0, double (&el)[2][5], double (&dpv)[5]) {
        unsigned elvo = elvoff, dpvo = dpvoff;
        asm volatile("" : "+v"(elvo), "+v"(dpvo));
#pragma unroll
        for (int l = 0; l < 5; ++l) dpv[l] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.dP + (size_t)l * NTp + n0) + dpvo);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double* ub = A.EL + (size_t)(GQ_Q * h + w + 16 * j) * NTp + n0;       // uniform
#pragma unroll
            for (int l = 0; l < 5; ++l)
                el[j][l] = !uv[j] ? 0. : (ABL == 1 ? 1.0 : *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ub + (size_t)l * A.EL_lstride) + elvo));      // (default cache policy: non-temporal loads measured 4-7 % slower here)
        }
    };
    auto z_build = [&](int bufZ, const double (&el)[2][5], const double (&dpv)[5]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double z = 0.;
#pragma unroll
            for (int l = 0; l < 5; ++l) z = fma(el[j][l], dpv[l], z);
            const double zo = __shfl_xor(z, 32);
            if (qs == 0) Zs[bufZ * GQ_Z_D + (w + 16 * j) * GQ_RS + img] = z + zo;       // (labels 0..4) + (labels 5..9)
        }
    };

    // ---- MFMA roles: wave (rgp, J): rgp = 2 s + half -> the row tiles a = 64 half + 16 r + i (r = 0..3) of site-index value s; J = column
    //      tile of this workgroup's 64 columns 2q' + t (lane i: q' = c >> 1, t = c & 1)
    const int rgp = w >> 2, J = w & 3;
    const int li = lane & 15, g = lane >> 4;
    const int sI = rgp >> 1, a0 = 16 * NR * (rgp & 1);
    const int eoff = (a0 + li) * GQ_RS + g;                               // + 16 r GQ_RS for row tile r, + 4 ks for k-step ks (lane group g: image 4 ks + g)
    const int cc = 16 * J + li;                                            // column 64 h + cc
    const int zoff = (cc >> 1) * GQ_RS + g;
    const int woff = (2 * sI + (cc & 1)) * GQ_RS + g;
    // ---- prologue
    {   // the first TWO chunks are requested together (the accumulators are not live yet: their registers hold the first chunk's requests)
        double2 pc0[3]; double el0[2][5], dpv0[5];
        piece_load(n_of(0), pc0);
        el_load(n_of(0), el0, dpv0);
        if (ABL != 6) { piece_load(n_of(1), pc); el_load(n_of(1), el, dpv); }
        piece_store(0, pc0);
        z_build(0, el0, dpv0);
        if (ABL == 6) { piece_load(n_of(1), pc); el_load(n_of(1), el, dpv); }      // (probe: the second chunk requested when the first has landed)
    }
    gq_barrier();
    const long long ts1 = (ABL == 5 || ABL == 6) ? (long long)wall_clock64() : 0;

    f64x4g acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = f64x4g{0., 0., 0., 0.};
    double en[NR], zn, on;                             // the fragments of the NEXT k-step
    auto frag_load = [&](int buf, int ks) {
        const double* Eb = Es + buf * GQ_E_D + eoff + 4 * ks;
#pragma unroll
        for (int r = 0; r < NR; ++r) en[r] = Eb[16 * r * GQ_RS];
        zn = Zs[buf * GQ_Z_D + zoff + 4 * ks];
        on = Ws[buf * GQ_W_D + woff + 4 * ks];
    };
    frag_load(0, 0);
    for (int k = 0; k < nch; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // the staging share of this wave, at the place in the stage that is its place on the SIMD (waves w, w + 4, w + 8, w + 12 share
            // one): before k-step 0, 2, 4 or 6 -- never after the last one, where no other wave would have MFMAs left to run beside it and
            // the whole workgroup would wait for it at the barrier.  What was requested one stage ago goes to the other buffers, the next
            // requests go out.
            if (ABL != 3 && ks == GQ_SLOT(rgp)) {
                __builtin_amdgcn_s_setprio(GQ_PRIO);         // few instructions beside the MFMAs of three other waves: issue them ahead (2 % of the launch)
                piece_store(nxt, pc);                  // E / weights of chunk k + 1
                z_build(nxt, el, dpv);                 // Z of chunk k + 1
                piece_load(n_of(k + 2), pc);
                el_load(n_of(k + 2), el, dpv);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            double ec[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) ec[r] = en[r];
            const double bc = zn * on;
            __builtin_amdgcn_sched_barrier(0);
            if (ks < 7) frag_load(cur, ks + 1);        // (the first k-step of the next stage is read behind the barrier)
            __builtin_amdgcn_sched_barrier(0);       // reads of the next k-step FIRST, then this one's MFMAs: left alone the scheduler issues the MFMAs first
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(ec[r], bc, acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);       // (keeps the order stated: reads of the next k-step, then this one's MFMAs)
        }
        gq_barrier();
        frag_load(nxt, 0);
    }

    // ---- epilogue: the quad's partial G, M-layout rows 2a + s, columns 2q + t, out to the padded extents (rows and columns beyond the bond
    //      dimensions come out as the zeros the consumer expects there)
    const long long ts2 = (ABL == 5 || ABL == 6) ? (long long)wall_clock64() : 0;
    double* out = A.slab + (size_t)grp * A.Kp * A.Np;
    const int col = 2 * GQ_Q * h + cc;
    if (col < A.Np) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int row = 2 * (a0 + 16 * r + g + 4 * e4) + sI;
                if (row < A.Kp) out[(size_t)row * A.Np + col] = acc[r][e4];
            }
    }
    if ((ABL == 5 || ABL == 6) && A.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) { long long* d = A.dbg + 4 * (size_t)b; d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = (long long)wall_clock64(); }
    }
}

static inline bool gq_pair_form(const Bgemm64Args& a) { return a.mI <= 64 && a.mO <= 64 && a.Kp <= 128 && a.Np <= 128; }

bool grad_quad_applies(const tnml_ctx* c, const Bgemm64Args& a) {
    if (!c->grad_quad || !a.EL || !a.env64 || a.L != 1 || a.w) return false;
    if (a.mI < 1 || a.mO < 1 || a.mI > 128 || a.mO > 128 || a.Kp < 2 * a.mI || a.Np < 2 * a.mO || a.Kp > 256 || a.Np > 256 || a.NTp % GQ_TI) return false;
    // unforced: from 4 096 images per rank on and from mI mO >= 72^2 on -- its 256 x 256 tile grid is fixed (155-160 us at 60 000 images whatever
    // the bond), but the tiles k_bgemm64 has for other sizes than 120 are far from it: 192 / 203 / 210 / 376 / 387 / 273 us at m = 72 / 88 / 96 /
    // 104 / 112 / 128 (profiles/r06_grad_quad_by_bond_dimension.txt).  Bonds up to 64 x 64 have the PAIR form (128 x 128 tile grid): unforced
    // for 33^2 <= mI mO <= 56^2 from 15 360 images per rank on -- 59 / 60 / 63 us at m = 33 / 40 / 48 against 73 / 79 / 70 of k_bgemm64's 128 x 64
    // tiles; at m = 60-64 both stream their 320 MB at ~5 TB/s (67-76 against 71-74 us, box by box) and at 7 500 images k_bgemm64 is ahead
    // (19.3 against 20.3 us): those stay where they were (profiles/r06_grad_pair_form.txt)
    if (c->grad_quad == 1) {
        if (gq_pair_form(a)) {
            const int mm = a.mI * a.mO;
            if (!c->grad_pair || a.NTp < 15360 || mm < c->grad_pair_min * c->grad_pair_min || mm > c->grad_pair_max * c->grad_pair_max) return false;
        } else if (a.NTp < 4096 || a.mI * a.mO < 72 * 72) return false;
    }
    if ((size_t)TNML_NL * a.EL_lstride * sizeof(double) >= ((size_t)1 << 32)) return false;      // 32-bit lane offsets
    return c->slab_bytes >= (size_t)64 * a.Kp * a.Np * sizeof(double);
}

template <int NR>
static int grad_quad_go(tnml_ctx* c, const GradQuadArgs& K, int grid) {
    const size_t lds = sizeof(double) * GQ_LDS_DOUBLES_N(NR);
    bool& done = NR == 4 ? c->attr_gq : c->attr_gp;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<0, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<1, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grad_quad<3, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return tnml_fail(c, "grad_quad: cannot reserve %zu bytes of LDS", lds);
        done = true;
    }
    ProfScope ps(c, KC_GRAD_QUAD);
    // (grad_quad = 3 / 5: the ablations of the record under profiles/ -- compute side alone / the MFMA loop alone; wrong results by construction)
    if (c->grad_quad == 3)      hipLaunchKernelGGL((k_grad_quad<1, NR>), dim3(grid), dim3(1024), lds, c->stream, K);
    else if (c->grad_quad == 5) hipLaunchKernelGGL((k_grad_quad<3, NR>), dim3(grid), dim3(1024), lds, c->stream, K);
    else                        hipLaunchKernelGGL((k_grad_quad<0, NR>), dim3(grid), dim3(1024), lds, c->stream, K);
    return 0;
}

int launch_grad_quad(tnml_ctx* c, const Bgemm64Args& a, double* G) {
    if (!c->cu_count) { hipDeviceProp_t pr; c->cu_count = hipGetDeviceProperties(&pr, c->cfg.device) == hipSuccess ? pr.multiProcessorCount : 256; }
    const bool pair = gq_pair_form(a) && c->grad_pair != 0;          // (grad_pair = 0: the quad form for every bond -- tests, A/B)
    const int nwg = pair ? 2 : 4;                                  // workgroups per group
    GradQuadArgs K;
    K.EI = static_cast<const double*>(a.EI); K.phiI = static_cast<const double*>(a.phiI); K.phiO = static_cast<const double*>(a.phiO);
    K.EL = static_cast<const double*>(a.EL); K.EL_lstride = a.EL_lstride; K.dP = a.dPz; K.NTp = a.NTp;
    K.mI = a.mI; K.mO = a.mO; K.Kp = a.Kp; K.Np = a.Np;
    K.slab = static_cast<double*>(c->slab);
    K.nchunks = a.NTp / GQ_TI;
    int groups = c->cu_count / nwg;                                // one workgroup per CU
    const int cap = (int)(c->slab_bytes / ((size_t)a.Kp * a.Np * sizeof(double)));
    if (groups > cap) groups = cap;
    if (c->bgemm_wgs > 0 && c->bgemm_wgs / nwg < groups) groups = c->bgemm_wgs / nwg > 0 ? c->bgemm_wgs / nwg : 1;      // test knob: fewer groups -> more stages each
    if (groups > K.nchunks) groups = K.nchunks;
    K.per = (K.nchunks + groups - 1) / groups;
    K.ngroups = (K.nchunks + K.per - 1) / K.per;
    const int grid = 8 * nwg * ((K.ngroups + 7) / 8);             // blocks b, b + 8, (b + 16, b + 24) of a run of 8 nwg = one group on one XCD
    if (pair) TCK(grad_quad_go<2>(c, K, grid));
    else      TCK(grad_quad_go<4>(c, K, grid));
    const size_t n = (size_t)a.Kp * a.Np;
    if (c->defer_slab) c->slab_pending = K.ngroups;            // the CG vector kernel that consumes G sums the slabs itself (slab order: the same bits)
    else {
        ProfScope ps(c, KC_SLABRED);
        launch_slab_reduce64(c, static_cast<const double*>(c->slab), G, n, K.ngroups);
    }
    HIPCK(c, hipGetLastError());
    return 0;
}
