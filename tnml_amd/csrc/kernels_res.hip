// kernels_res.hip -- the two image-proportional contractions of a CG pass with the SMALL operand resident in registers
// (m = 120, fp64 storage, Label index on an environment; BASELINE config 3):
//
//   k_fwd_res  : B*t.v (fixedL.cc:318,377,399,416).  The bond matrix M (240 x 240 doubles = 460 KB) is the operand every image
//                tile shares, so it should be fetched ONCE per workgroup and not once per tile (k_fwd_fused re-stages it from L2
//                through LDS for each of its 938 tiles: 15 chunks, 30 barriers per tile).  A CU's register file holds 512 KB, so
//                one CU cannot keep all of M beside anything else -- two can: the 15 column tiles of M are dealt to a PAIR of
//                workgroups (8 + 7), each GEMM wave keeps ONE 16-column tile for the full reduction length in 120 VGPRs for the
//                whole launch, and the pair walks the same 32-image tiles.  Per tile the only LDS traffic is the image operand
//                (the Label-free environment, 30 KB, double buffered), one ds_read_b64 per MFMA, and there are two barriers per tile
//                instead of thirty.  Column tiles of M are output links q: workgroup `half` owns q in [64 half, 64 half + nq) and
//                therefore streams only ITS rows of the Label-carrying environment -- the 577 MB stream is still read exactly once.
//                What the split costs: each workgroup holds the label dot over its own q only, so the launch leaves two partial
//                output vectors Ppart[half][l][n]; k_pfinish adds them (fixed order) and does the per-image epilogue.
//   k_pfinish  : P = Ppart[0] + Ppart[1], then dP = delta - P, per-label cost partials, argmax count or |P|^2 (the tail of
//                k_labeldot) per 64-image wave; k_reduce_partials sums them.  (A "last workgroup to arrive reduces" form of this kernel
//                cost 26 us per launch instead of 8 + 5: its __threadfence() is an L2 write-back on this multi-XCD part.)
//
// Wave roles of k_fwd_res (768 lanes, one workgroup per CU): waves 0..7 GEMM (wave w: column tile 8 half + w), waves 8..11 stream the
// Label-carrying environment for the tile the GEMM waves finished one round earlier (4 rows of 10 labels in flight per lane) and
// stage the next tile's image operand.  Deterministic: fixed summation order everywhere.
#include "tnml_internal.h"

typedef double f64x4r __attribute__((ext_vector_type(4)));

#define FR_TI 32                       // images per tile
// NKA = MFMA k-steps over the rows a of the input environment (4 each; 30 at m = 120); rows of the staged image operand: 4 NKA environment
// rows, phiI[0..1], phiO[0..1]
#define FR_ROWS(NKA) (4 * (NKA) + 4)
#define FR_EIS(NKA) (FR_ROWS(NKA) * FR_TI)       // doubles per buffer
#define FR_LDS_DOUBLES_N(NKA) (2 * FR_EIS(NKA) + 2 * 64 * FR_TI + 2 * 4 * TNML_NL * FR_TI)
#define FR_LDS_DOUBLES FR_LDS_DOUBLES_N(30)

// workgroup barrier WITHOUT the vmcnt(0) that __syncthreads() carries: the streaming waves keep four rows of the Label-carrying
// environment in flight across it.  LDS writes of this wave are complete (lgkmcnt(0)); what else has to have landed is stated at the call.
static __device__ __forceinline__ void fr_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One barrier per 32-image tile.  Round `it` of a workgroup (tiles pair, pair + npairs, ... = its tiles 0 .. niter-1):
//   GEMM waves      : tile it          -> U[it & 1] (LDS), reading the image operand X[it & 1]
//   streaming waves : stage tile it+1  -> X[(it+1) & 1] (LDS-DMA); label dot of tile it-1 from U[(it-1) & 1] -> red[it & 1];
//                     wave 8 first adds the four waves' parts of tile it-2 from red[(it-1) & 1] -> Ppart
// Every buffer a round writes was last read one round earlier, so the single barrier at the end of a round orders everything.
// PS, PK: the GEMM waves sleep 64 PS cycles after every PK-th k-step (4 MFMAs each).  Why: while the two GEMM waves of a SIMD have an
// MFMA ready, NO other wave of that SIMD gets a VALU instruction issued, whatever its priority or age
// (tools/probe/probe_valu_mfma.hip: ten FMAs of a third wave take 8 300 cycles instead of 80, fp64, fp32 and integer alike) -- the
// streaming wave's FMAs, and with them its next loads, would wait for stalls of the GEMM waves.  The pauses are those stalls, made on purpose.
// ABL (probe builds only): 1 = no label-dot loads (GEMM role alone), 2 = no MFMAs (streaming role alone), 5 = per-wave cycle counters
// Other bond dimensions than 120 x 120 (trained bonds shrink towards minm = maxm/2, fixedL.cc:593): NKA = ceil(mI / 4) rounded up to the next
// instantiation built (rows from mI on are staged from valid rows and meet zero rows of M); the T = Np / 16 column tiles of M (8 output links
// each; links from mO on: zero columns of M, rows of the Label-carrying environment taken from valid ones) are dealt ceil(T / 2) + floor(T / 2)
// to the two workgroups of a pair; the streaming waves always run their eight steps -- a step beyond the half's tiles re-reads the
// environment's first rows (cache hits) and weighs them zero, so that the hand-counted waits hold on every path.
// GEN = false: the 120 x 120 bond of the benchmark with its extents as constants (NKA = 30).
// NST = steps of the streaming waves per tile: 8, or 4 where no half has more than four column tiles (mO <= 64).
template <int PS, int PK, int ABL, int NKA = 30, bool GEN = false, int NST = 8>
__global__ __launch_bounds__(768) void k_fwd_res(FwdResArgs A) {
    static_assert(GEN || NKA == 30, "the constant-extent form is the 120 x 120 bond");
    static_assert(NST == 8 || (NST == 4 && GEN), "four or eight streaming steps per tile");
    constexpr int EIS = FR_EIS(NKA);
    const int mI = GEN ? A.mI : 120, mO = GEN ? A.mO : 120, Kp = GEN ? A.Kp : 240, Np = GEN ? A.Np : 240;
    extern __shared__ __attribute__((aligned(16))) double fr_lds[];
    double* EIs = fr_lds;                        // [2][4 NKA + 4][32]
    double* Us = EIs + 2 * EIS;                  // [2][64][32]: U[q - first link of the half][image]
    double* red = Us + 2 * 64 * FR_TI;           // [2][4][10][32]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    // workgroups b and b + 8 land on the same XCD (round-robin dispatch over 8 XCDs): the pair shares its image operand in one L2
    const int half = (b >> 3) & 1, pair = (b & 7) + 8 * (b >> 4), npairs = gridDim.x >> 1;
    const int NTp = A.NTp;
    const int T = Np >> 4, T0 = (T + 1) >> 1;
    const int Th = half ? T - T0 : T0;           // column tiles (of 8 output links) of this half: 8 + 7 at m = 120
    const int qb = half ? 8 * T0 : 0;            // its first output link
    const int niter = (A.ntiles - pair + npairs - 1) / npairs;      // >= 1 (launch_fwd_res sizes the grid)

    if (wid < 8) {
        // ---------------- GEMM role ----------------
        // X_n[2a + s] = E_n[a] phi_n[s], so T = phi[0] (E M_even) + phi[1] (E M_odd) with M_even / M_odd the rows 2a / 2a + 1 of M: two
        // accumulators per image group share ONE environment fragment per k-step (half the LDS reads) and the site feature is applied
        // once, in the epilogue, instead of once per MFMA.
        const int w = wid;
        const bool act = w < Th;
        const int ct = (half ? T0 : 0) + w;
        double me[NKA], mo[NKA];                 // M[2 (4 ks + g) + s][16 ct + i], s = 0 / 1
        {
            // MFMA row i of this wave's tile is column j = 2 q + t of M with q = 8 ct + (i & 7), t = i >> 3: a lane's accumulator rows
            // g, g + 4 (t = 0) and g + 8, g + 12 (t = 1) are then BOTH site-index values of the output links 8 ct + g, 8 ct + g + 4,
            // and the fold over t in the epilogue needs no cross-lane exchange
            const int g = lane >> 4, i = lane & 15;
            const int col = 2 * (8 * ct + (i & 7)) + (i >> 3);
#pragma unroll
            for (int ks = 0; ks < NKA; ++ks) {
                const bool in = act && 2 * (4 * ks + g) + 1 < Kp;            // (Kp is even: both rows or neither)
                me[ks] = in ? A.M[(size_t)(2 * (4 * ks + g)) * Np + col] : 0.;
                mo[ks] = in ? A.M[(size_t)(2 * (4 * ks + g) + 1) * Np + col] : 0.;
            }
        }
        fr_barrier();                            // prologue: the first tile's image operand is in X[0]
        long long t_mma = 0, t_epi = 0, t_bar = 0;
        const long long wc0 = ABL == 5 ? (long long)wall_clock64() : 0;
        for (int it = 0; it < niter + 2; ++it) {
            const long long c0 = ABL == 5 ? clock64() : 0;
            long long c1 = c0, c2 = c0;
            if (it < niter && act) {
                const double* Eb = EIs + (it & 1) * EIS;
                double* Ub = Us + (it & 1) * 64 * FR_TI;
                // (the lane index passes through an empty asm before the MFMA loop and again before the epilogue: everything derived
                // from it is then recomputed where it is used instead of living -- spilled -- across the 120 MFMAs)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                int g = ln >> 4, i = ln & 15;
                // Four accumulators per wave (2 image groups x even / odd rows of M): a dependent fp64 MFMA can only issue ~500 cycles
                // after its predecessor (measured: 2 chains per wave, 2 waves per SIMD = 52 % of the matrix pipe whatever else the kernel
                // does), so a SIMD needs >= 8 independent accumulator chains to keep its pipe busy.
                // (odd rows of the staged operand hold their two 16-image halves swapped: the rows one fragment read touches then fall on
                // different LDS banks)
                const double* ep0 = Eb + g * FR_TI + 16 * (g & 1) + i;            // E[4 ks + g][i]        (image group 0)
                const double* ep1 = Eb + g * FR_TI + 16 * ((g & 1) ^ 1) + i;      // E[4 ks + g][16 + i]   (image group 1)
                f64x4r ce0 = {0., 0., 0., 0.}, co0 = {0., 0., 0., 0.}, ce1 = {0., 0., 0., 0.}, co1 = {0., 0., 0., 0.};
                // environment fragments two k-steps ahead of their MFMAs; the group barriers pin the issue order the source states
                // (two LDS reads, four MFMAs) -- left alone the scheduler sinks every read to just before its use
                double a0 = ep0[0], b0 = ep1[0], a1 = ep0[4 * FR_TI], b1 = ep1[4 * FR_TI];
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int ks = 0; ks < (ABL == 2 ? 1 : NKA); ++ks) {
                    const double xa = a0, xb = b0;
                    a0 = a1; b0 = b1;
                    if (ks + 2 < NKA) { a1 = ep0[4 * (ks + 2) * FR_TI]; b1 = ep1[4 * (ks + 2) * FR_TI]; __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
                    ce0 = __builtin_amdgcn_mfma_f64_16x16x4f64(me[ks], xa, ce0, 0, 0, 0);
                    co0 = __builtin_amdgcn_mfma_f64_16x16x4f64(mo[ks], xa, co0, 0, 0, 0);
                    ce1 = __builtin_amdgcn_mfma_f64_16x16x4f64(me[ks], xb, ce1, 0, 0, 0);
                    co1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mo[ks], xb, co1, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    if (PS > 0 && ks % PK == PK - 1) { __builtin_amdgcn_s_sleep(PS); __builtin_amdgcn_sched_barrier(0); }
                }
                if (ABL == 5) c1 = clock64();
                // lane (g, i) holds rows g + 4e of its tile for image i: output links q = 8 ct + g (e = 0, 2: t = 0, 1) and q + 4 (e = 1, 3)
                ln = lane;
                asm volatile("" : "+v"(ln));
                g = ln >> 4; i = ln & 15;
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    const double pI0 = Eb[(4 * NKA) * FR_TI + 16 * grp + i], pI1 = Eb[(4 * NKA + 1) * FR_TI + 16 * (grp ^ 1) + i];
                    const double pO0 = Eb[(4 * NKA + 2) * FR_TI + 16 * grp + i], pO1 = Eb[(4 * NKA + 3) * FR_TI + 16 * (grp ^ 1) + i];
                    const f64x4r ce = grp ? ce1 : ce0, co = grp ? co1 : co0;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double t0 = fma(pI1, co[e], pI0 * ce[e]), t1 = fma(pI1, co[e + 2], pI0 * ce[e + 2]);
                        const double v = fma(pO1, t1, pO0 * t0);
                        Ub[(8 * w + g + 4 * e) * FR_TI + 16 * grp + i] = v;
                    }
                }
                if (ABL == 5) c2 = clock64();
            }
            fr_barrier();
            if (ABL == 5) { const long long c3 = clock64(); t_mma += c1 - c0; t_epi += c2 - c1; t_bar += c3 - c2; }
        }
        if (ABL == 5 && A.dbg && blockIdx.x < 16 && lane == 0) { long long* d = A.dbg + (blockIdx.x * 12 + wid) * 4; d[0] = t_mma; d[1] = t_epi; d[2] = t_bar; d[3] = (long long)wall_clock64() - wc0; }
    } else {
        // ---------------- streaming role ----------------
        // All global loads of this role are inline asm with hand-counted s_waitcnt vmcnt(N): the ring of rows below has to stay in
        // flight across rounds and barriers, and the compiler's own bookkeeping answers a loop-carried load with vmcnt(0).  Loads
        // return in order, so "row r has landed" = "at most (operations issued after row r) are outstanding".
        __builtin_amdgcn_s_setprio(3);          // few instructions, all of them latency critical: issue ahead of the GEMM waves of this SIMD
        const int sw = wid - 8;
        const int img = lane & 31, qs = lane >> 5;
        const int nk = Th;                                       // 8 rows of q per step over the 4 waves x 2 lane halves: one step per column tile
        // staging of a tile's image operand straight into LDS (no staging registers): one global_load_lds_dwordx4 moves 4 rows of
        // 32 images (1 KB; lane -> row lane >> 4, images 2 (lane & 15) ...): NKA groups of environment rows 4 grp .. and the four feature
        // rows (phiI[0..1], phiO[0..1]) as group NKA; every wave issues exactly 8 pieces (groups sw, sw + 4, ...; slots beyond the last group
        // repeat groups -- same bytes to the same place) so that the counts below hold on every path.  Rows from mI on only have to be finite
        // (they meet zero rows of M): the group that straddles mI repeats its last valid row, the groups beyond it take rows 0..3.
        const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)fr_lds);
        const int xcol = 2 * ((lane & 15) ^ (((lane >> 4) & 1) << 3));      // source images of this lane's 16 bytes: odd rows swap their halves
        const double* const phr = ((lane >> 4) < 2 ? A.phiI : A.phiO) + (size_t)((lane >> 4) & 1) * NTp + xcol;
        const double* const eir = A.EI + (size_t)(lane >> 4) * NTp + xcol;
        const int gb = GEN ? (mI - 1) >> 2 : NKA;                           // the last group with a valid row (constant extents: every group is whole)
        const int rb = 4 * gb + (lane >> 4) < mI ? 4 * gb + (lane >> 4) : mI - 1;
        const double* const eib = A.EI + (size_t)rb * NTp + xcol;
        auto x_stage = [&](int tile, int buf) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int grp = sw + 4 * r;                                               // uniform
                const double* src;
                if constexpr (GEN) {
                    if (grp > NKA) grp -= NKA + 1;
                    if (grp > NKA) grp -= NKA + 1;
                    src = (grp == NKA ? phr : (grp < gb ? eir + (size_t)(4 * grp) * NTp : (grp == gb ? eib : eir))) + (size_t)tile * FR_TI;
                } else {
                    grp = r < 7 ? sw + 4 * r : (sw < 2 ? sw + 28 : (sw == 2 ? 30 : 3));
                    src = (grp == 30 ? phr : eir + (size_t)(4 * grp) * NTp) + (size_t)tile * FR_TI;
                }
                const unsigned dst = lds0 + (unsigned)((buf * EIS + 4 * grp * FR_TI) * sizeof(double));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        };
        const double* ELw = A.EL + (size_t)(qb + 2 * sw) * NTp;                   // uniform part of this wave's rows
        const unsigned eoff = (unsigned)(((size_t)qs * NTp + img) * sizeof(double));  // lane part (bytes)
        const unsigned uoff = (unsigned)((2 * sw + qs) * FR_TI + img);                // lane part of a U read (doubles)
        // row offset (doubles, from ELw) and lane part of the eight steps: step k < nk reads the links qb + 8 k + 2 sw + (0, 1); a step beyond
        // the half's tiles reads the links 2 sw + (0, 1) of the environment's first rows and is weighed zero; links from mO on (the last tile
        // of all, mO not a multiple of 8) meet U = 0 (zero columns of M): a pair that straddles mO takes its valid row twice, a pair beyond
        // it the first rows again
        size_t roff[8]; bool same[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q0 = qb + 8 * k + 2 * sw;
            const bool real = k < nk && q0 < mO;
            roff[k] = real ? (size_t)(8 * k) * NTp : (size_t)0 - (size_t)qb * NTp;
            same[k] = GEN && real && q0 + 1 >= mO;
        }
        const unsigned eoff0 = (unsigned)(img * sizeof(double));
        // four rows (of 10 labels) of the Label-carrying environment in flight per lane, as a ring that runs on across the tiles
        // and across the barriers: while row k of a tile is consumed, row k + 4 (of this tile or the next) is requested
        double ea[TNML_NL], eb[TNML_NL], ec[TNML_NL], ed[TNML_NL];
        auto s_load = [&](int tile, int k, double (&e)[TNML_NL]) {
            const unsigned vo = same[k] ? eoff0 : eoff;
#pragma unroll
            for (int l = 0; l < TNML_NL; ++l) {
                const double* bp = ELw + (size_t)tile * FR_TI + (size_t)l * A.EL_lstride + roff[k];      // uniform: an SGPR pair
                if (ABL == 1) e[l] = 1.0;
                else asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(e[l]) : "v"(vo), "s"(bp) : "memory");
            }
        };
#define FR_WAIT(N, e) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(e[8]), "+v"(e[9]) :: "memory")
        // label dot of a tile (its rows 0..3 are in the ring) over this half's output links; requests rows 0..3 of tile `ntile`;
        // mid(): the round's other memory operations (8 LDS-DMA pieces, wave 8 also 5 stores), issued behind rows 4..7
        auto label_dot = [&](int tile, int ntile, const double* Ub, double* rb, auto&& mid) {
            double px[TNML_NL];
#pragma unroll
            for (int l = 0; l < TNML_NL; ++l) px[l] = 0.;
            auto s_use = [&](int k, const double (&e)[TNML_NL], bool on) {
                double u = Ub[8 * k * FR_TI + uoff];
                if (!on) u = 0.;
#pragma unroll
                for (int l = 0; l < TNML_NL; ++l) px[l] = fma(e[l], u, px[l]);
            };
            // (a fence pins the FMAs above it: px passes through it -- left alone the compiler sinks them to the end of the round)
#define FR_FENCE() do { asm volatile("" : "+v"(px[0]), "+v"(px[1]), "+v"(px[2]), "+v"(px[3]), "+v"(px[4]), "+v"(px[5]), "+v"(px[6]), "+v"(px[7]), "+v"(px[8]), "+v"(px[9]) :: "memory"); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
            // outstanding behind row 0: rows 1..3 = 30 loads, and so on down the ring
            if constexpr (NST == 8) {
                FR_WAIT(30, ea); s_use(0, ea, nk > 0); FR_FENCE(); s_load(tile, 4, ea);
                FR_WAIT(30, eb); s_use(1, eb, nk > 1); FR_FENCE(); s_load(tile, 5, eb);
                FR_WAIT(30, ec); s_use(2, ec, nk > 2); FR_FENCE(); s_load(tile, 6, ec);
                FR_WAIT(30, ed); s_use(3, ed, nk > 3); FR_FENCE(); s_load(tile, 7, ed);
                mid();
                // behind row 4: rows 5..7 (30) + the 8 pieces (+ 5 stores on wave 8: the wait is then stricter than needed, never weaker)
                FR_WAIT(38, ea); s_use(4, ea, nk > 4); FR_FENCE(); s_load(ntile, 0, ea);
                FR_WAIT(38, eb); s_use(5, eb, nk > 5); FR_FENCE(); s_load(ntile, 1, eb);
                FR_WAIT(38, ec); s_use(6, ec, nk > 6); FR_FENCE(); s_load(ntile, 2, ec);
                FR_WAIT(38, ed); s_use(7, ed, nk > 7); FR_FENCE(); s_load(ntile, 3, ed);
            } else {
                // four steps per tile: every step requests its row of the NEXT tile.  Behind row 0 in the first round: rows 1..3 (30); later:
                // row 1, the 8 pieces, rows 2, 3 (38) -- 30 is never weaker; behind rows 2, 3: 38 in every round
                FR_WAIT(30, ea); s_use(0, ea, nk > 0); FR_FENCE(); s_load(ntile, 0, ea);
                FR_WAIT(30, eb); s_use(1, eb, nk > 1); FR_FENCE(); s_load(ntile, 1, eb);
                mid();
                FR_WAIT(38, ec); s_use(2, ec, nk > 2); FR_FENCE(); s_load(ntile, 2, ec);
                FR_WAIT(38, ed); s_use(3, ed, nk > 3); FR_FENCE(); s_load(ntile, 3, ed);
            }
#undef FR_FENCE
#pragma unroll
            for (int l = 0; l < TNML_NL; ++l) {
                px[l] += __shfl_xor(px[l], 32);
                if (qs == 0) rb[(sw * TNML_NL + l) * FR_TI + img] = px[l];
            }
        };
        // wave 8 adds the four waves' parts in a fixed order -> Ppart[half][l][image]
        auto finalize = [&](int tile, const double* rb) {
            if (sw == 0) {
#pragma unroll
                for (int l2 = 0; l2 < TNML_NL / 2; ++l2) {           // 64 lanes: image lane & 31, labels of parity lane >> 5
                    const int l = 2 * l2 + qs;
                    const double s = ((rb[(0 * TNML_NL + l) * FR_TI + img] + rb[(1 * TNML_NL + l) * FR_TI + img]) + rb[(2 * TNML_NL + l) * FR_TI + img]) + rb[(3 * TNML_NL + l) * FR_TI + img];
                    A.Ppart[((size_t)half * TNML_NL + l) * NTp + (size_t)tile * FR_TI + img] = s;
                }
            }
        };
        x_stage(pair, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fr_barrier();                                            // prologue: the first tile's image operand is in X[0]
        // round 0: nothing to contract yet; open the ring with rows 0..3 of the first tile
        x_stage(niter > 1 ? pair + npairs : pair, 1);
        s_load(pair, 0, ea); s_load(pair, 1, eb); s_load(pair, 2, ec); s_load(pair, 3, ed);
        asm volatile("s_waitcnt vmcnt(40)" ::: "memory");        // the staged tile has landed (issued before the 40 row loads)
        fr_barrier();
        long long t_dot = 0, t_bar = 0;
        for (int it = 1; it <= niter; ++it) {
            const long long c0 = ABL == 5 ? clock64() : 0;
            const int tile = pair + (it - 1) * npairs;
            label_dot(tile, it < niter ? tile + npairs : tile, Us + ((it - 1) & 1) * 64 * FR_TI, red + (it & 1) * 4 * TNML_NL * FR_TI, [&]() {
                x_stage(it + 1 < niter ? pair + (it + 1) * npairs : tile, (it + 1) & 1);      // (no next tile: this one again, into the idle buffer)
                if (it >= 2) finalize(pair + (it - 2) * npairs, red + ((it - 1) & 1) * 4 * TNML_NL * FR_TI);
            });
            // all but the row loads requested behind the pieces (rows 0..3 / rows 2, 3 of the next tile): the staged tile has landed
            if constexpr (NST == 8) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            else                    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            const long long c1 = ABL == 5 ? clock64() : 0;
            fr_barrier();
            if (ABL == 5) { const long long c2 = clock64(); t_dot += c1 - c0; t_bar += c2 - c1; }
        }
        if (ABL == 5 && A.dbg && blockIdx.x < 16 && lane == 0) { long long* d = A.dbg + (blockIdx.x * 12 + wid) * 4; d[0] = t_dot; d[1] = 0; d[2] = t_bar; d[3] = niter; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the ring's last requests are never consumed)
        finalize(pair + (niter - 1) * npairs, red + (niter & 1) * 4 * TNML_NL * FR_TI);
        fr_barrier();
#undef FR_WAIT
    }
}

// ==========================================================================================================================
// k_shift_res -- the Label-carrying environment shift (TrainStates::shiftE / init, fixedL.cc:142-149,221-228), bond dimensions up to 120 x 128:
//   E'[l][y][n] = sum_{a,s} E[l][a][n] phi[s][n] A[a,s,y]  =  phi[0] (E M_even) + phi[1] (E M_odd),  M = the packed site matrix [Kp][Np].
// M (245 KB at m = 120) fits the registers of ONE workgroup: 8 waves, a wave keeps ONE 16-column tile for the whole reduction length (4 NKS
// rows of the input environment, NKS = the template parameter: 2 NKS doubles per lane) for the whole launch and the workgroup walks 64-image
// tiles of the 10 x NTp rows.  No second role: every wave issues its share of the next tile's LDS-DMA pieces at the top of a round, so nothing
// but MFMAs, LDS fragment reads and stores runs beside the matrix pipe.  The second half of a tile's outputs is stored at the top of the NEXT
// round: the wait that lands the DMA pieces at the end of a round (vmcnt counts loads and stores alike) then finds only stores that are half a
// round old.  One barrier per tile.
// Other bond dimensions than 120 (trained bonds shrink towards minm = maxm/2, fixedL.cc:593): the input dimension picks the instantiation
// (NKS = ceil(mI / 4) rounded up to the next one built; rows from mI on are staged from a valid row and meet zero rows of M); with at most four
// column tiles (mO <= 64) the eight waves take one 32-image half of the tile each, so all of them keep issuing MFMAs.
// ==========================================================================================================================
#define SR_TI 64

template <int NKS>
__global__ __launch_bounds__(512) void k_shift_res(ShiftResArgs A) {
    constexpr int ROWS = 4 * NKS + 2;                  // 4 NKS environment rows, phi[0], phi[1]
    constexpr int NP = 2 * NKS + 1;                    // DMA pieces of 2 rows per tile (the last one: the two feature rows)
    extern __shared__ __attribute__((aligned(16))) double sr_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTp = A.NTp, tpl = NTp / SR_TI;
    const int G = gridDim.x;
    const int niter = (A.ntiles - (int)blockIdx.x + G - 1) / G;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)sr_lds));
    const unsigned doff = (unsigned)(((size_t)(lane >> 5) * NTp + 2 * (lane & 31)) * sizeof(double));     // lane part of a piece's source: 2 rows x 64 images
    auto dma16 = [&](const double* base, unsigned voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
    };
    // tile t = rows n0 .. n0 + 63 of label l; NP pieces of 2 rows; 8 per wave, the last slots repeat pieces (same bytes to the same place).
    // Rows from mI on (mI odd, or below the instantiation's 4 NKS) meet zero rows of M and only have to be finite: a piece that straddles
    // mI takes its last valid row twice (lane offset without the row part), a piece beyond it takes rows 0 and 1
    const unsigned doff0 = (unsigned)((2 * (lane & 31)) * sizeof(double));
    auto stage = [&](int t, int buf) {
        const int l = t / tpl, n0 = (t - l * tpl) * SR_TI;
        const double* eb = A.EI + (size_t)l * A.EI_lstride + n0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int p = w + 8 * r;
            if (p >= NP) p -= NP;
            if (p >= NP) p -= NP;
            const int r0 = 2 * p < A.mI ? 2 * p : 0;
            dma16(p == NP - 1 ? A.phiI + n0 : eb + (size_t)r0 * NTp, (p != NP - 1 && 2 * p == A.mI - 1) ? doff0 : doff, lds0 + (unsigned)((buf * ROWS * SR_TI + 2 * p * SR_TI) * sizeof(double)));
        }
    };
    // wave -> column tile and image halves: more than four column tiles: wave w keeps tile w and runs both 32-image halves of a tile;
    // up to four: waves w and w + 4 keep tile w & 3 and run one half each
    const int CT = A.Np >> 4;
    const bool split = CT <= 4;
    const int ct = split ? (w & 3) : w;
    const bool act = ct < CT;
    const bool do0 = !split || w < 4, do1 = !split || w >= 4;
    double me[NKS], mo[NKS];
    {
        const int g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool in = act && 2 * (4 * ks + g) + 1 < A.Kp;          // (Kp is even: both rows or neither)
            me[ks] = in ? A.M[(size_t)(2 * (4 * ks + g)) * A.Np + 16 * ct + i] : 0.;
            mo[ks] = in ? A.M[(size_t)(2 * (4 * ks + g) + 1) * A.Np + 16 * ct + i] : 0.;
        }
    }
    if (niter > 0) stage(blockIdx.x, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fr_barrier();
    double ob[8];                                   // second-half outputs of the previous tile
    double* obp = nullptr;
    for (int it = 0; it < niter; ++it) {
        const int t = blockIdx.x + it * G;
        const int l = t / tpl, n0 = (t - l * tpl) * SR_TI;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int g = ln >> 4, i = ln & 15;
        if (obp) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int y = 16 * ct + g + 4 * (q & 3); if (y < A.mO) obp[(size_t)(4 * (q & 3)) * NTp + 16 * (q >> 2)] = ob[q]; }
        }
        if (it + 1 < niter) stage(t + G, (it + 1) & 1);
        const double* Eb = sr_lds + (it & 1) * ROWS * SR_TI;
        double* op = A.out + (size_t)l * A.out_lstride + (size_t)(16 * ct + g) * NTp + n0 + i;
        if (act) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 0 ? !do0 : !do1) continue;                         // uniform
                const double* ep0 = Eb + g * SR_TI + 32 * pass + i;            // E[4 ks + g][32 pass + i], [.. + 16 + i]
                const double* ep1 = ep0 + 16;
                f64x4r ce0 = {0., 0., 0., 0.}, co0 = {0., 0., 0., 0.}, ce1 = {0., 0., 0., 0.}, co1 = {0., 0., 0., 0.};
                double a0 = ep0[0], b0 = ep1[0], a1 = ep0[4 * SR_TI], b1 = ep1[4 * SR_TI];
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const double xa = a0, xb = b0;
                    a0 = a1; b0 = b1;
                    if (ks + 2 < NKS) { a1 = ep0[4 * (ks + 2) * SR_TI]; b1 = ep1[4 * (ks + 2) * SR_TI]; __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
                    ce0 = __builtin_amdgcn_mfma_f64_16x16x4f64(me[ks], xa, ce0, 0, 0, 0);
                    co0 = __builtin_amdgcn_mfma_f64_16x16x4f64(mo[ks], xa, co0, 0, 0, 0);
                    ce1 = __builtin_amdgcn_mfma_f64_16x16x4f64(me[ks], xb, ce1, 0, 0, 0);
                    co1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mo[ks], xb, co1, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                // lane (g, i) holds output links y = 16 ct + g + 4 e of images 32 pass + 16 grp + i
                ln = lane;
                asm volatile("" : "+v"(ln));
                g = ln >> 4; i = ln & 15;
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    const double pI0 = Eb[(4 * NKS) * SR_TI + 32 * pass + 16 * grp + i], pI1 = Eb[(4 * NKS + 1) * SR_TI + 32 * pass + 16 * grp + i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double v = fma(pI1, grp ? co1[e] : co0[e], pI0 * (grp ? ce1[e] : ce0[e]));
                        if (pass == 0) { if (16 * ct + g + 4 * e < A.mO) op[(size_t)(4 * e) * NTp + 16 * grp] = v; }
                        else ob[4 * grp + e] = v;
                    }
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (do1) obp = op + 32;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's pieces have landed (the stores still counted are half a round old)
        fr_barrier();
    }
    if (obp) {
        const int g = lane >> 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int y = 16 * ct + g + 4 * (q & 3); if (y < A.mO) obp[(size_t)(4 * (q & 3)) * NTp + 16 * (q >> 2)] = ob[q]; }
    }
}

// the instantiations built: input dimensions up to 4 NKS
static const int sr_nks_list[] = {12, 16, 18, 20, 22, 24, 26, 28, 30};
static int sr_nks_for(int mI) {
    for (int v : sr_nks_list) if (4 * v >= mI) return v;
    return 0;
}
bool shift_res_applies(int mI, int mO) { return mI >= 33 && sr_nks_for(mI) != 0 && mO >= 1 && mO <= 128; }      // (mI >= 2: a piece is two rows)

template <int NKS>
static int shift_res_go(tnml_ctx* c, const ShiftResArgs& a, int grid) {
    const size_t lds = sizeof(double) * 2 * (4 * NKS + 2) * SR_TI;
    if (!c->attr_sr[NKS]) {                           // (function attributes are per device: remembered per context)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_shift_res<NKS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return tnml_fail(c, "shift_res: cannot reserve %zu bytes of LDS", lds);
        c->attr_sr[NKS] = true;
    }
    ProfScope ps(c, KC_FGEMM_SHIFT);
    hipLaunchKernelGGL(k_shift_res<NKS>, dim3(grid), dim3(512), lds, c->stream, a);
    return 0;
}
int launch_shift_res(tnml_ctx* c, const ShiftResArgs& a_in) {
    ShiftResArgs a = a_in;
    if (a.NTp % SR_TI) return tnml_fail(c, "shift_res: image count not a multiple of %d", SR_TI);
    if (((size_t)a.L * a.EI_lstride) * sizeof(double) >= ((size_t)1 << 32)) return tnml_fail(c, "shift_res: environment larger than 4 GB (32-bit lane offsets)");
    if (!shift_res_applies(a.mI, a.mO) || a.Np % 16 || a.Np < a.mO || a.Np > 128) return tnml_fail(c, "shift_res: bond dimensions %d x %d (packed %d) not served", a.mI, a.mO, a.Np);
    if (!c->cu_count) { hipDeviceProp_t pr; c->cu_count = hipGetDeviceProperties(&pr, c->cfg.device) == hipSuccess ? pr.multiProcessorCount : 256; }
    a.ntiles = a.L * (a.NTp / SR_TI);
    int grid = c->cu_count;
    if (c->res_grid > 0 && c->res_grid < grid) grid = c->res_grid;
    if (grid > a.ntiles) grid = a.ntiles;
    switch (sr_nks_for(a.mI)) {
        case 12: TCK((shift_res_go<12>(c, a, grid))); break;
        case 16: TCK((shift_res_go<16>(c, a, grid))); break;
        case 18: TCK((shift_res_go<18>(c, a, grid))); break;
        case 20: TCK((shift_res_go<20>(c, a, grid))); break;
        case 22: TCK((shift_res_go<22>(c, a, grid))); break;
        case 24: TCK((shift_res_go<24>(c, a, grid))); break;
        case 26: TCK((shift_res_go<26>(c, a, grid))); break;
        case 28: TCK((shift_res_go<28>(c, a, grid))); break;
        default: TCK((shift_res_go<30>(c, a, grid))); break;
    }
    HIPCK(c, hipGetLastError());
    return 0;
}

// wave-level cost buckets of 64 images -> out[12] (the epilogue of k_labeldot / k_fwd_fused)
static __device__ __forceinline__ void res_wave_partials(double val, int lab, int cor, bool pap, double* out, int lane) {
    if (pap) {
        const double s = wave_sum(val);
        if (lane < 12) out[lane] = lane == 11 ? s : 0.;
        return;
    }
    double mine = 0.;
#pragma unroll
    for (int t = 0; t < TNML_NL; ++t) {
        const double s = wave_sum(lab == t ? val : 0.);
        if (lane == t) mine = s;
    }
    const double sc = wave_sum((double)cor);
    if (lane == 10) mine = sc;
    if (lane < 12) out[lane] = mine;
}

// P = Ppart[0] + Ppart[1] (npart = 2) or an update P += alpha Pp (npart = 0: the fast CG's output update, k_pupdate), then the
// per-image epilogue; partial sums per 64-image wave -> partials[NTp / 64][12].
__global__ __launch_bounds__(256) void k_pfinish(PfinishArgs A) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int NTp = A.NTp;
    const int ni = blockIdx.x * 256 + tid;
    if (A.conv && A.conv[0] != 0.) return;                    // CG already converged: everything stays as it is (uniform over the grid)
    const int lab = A.label[ni];
    double P[TNML_NL];
    if (A.npart == 2) {
#pragma unroll
        for (int l = 0; l < TNML_NL; ++l) P[l] = A.Ppart[(size_t)l * NTp + ni] + A.Ppart[((size_t)TNML_NL + l) * NTp + ni];
    } else {
        const double a = A.alpha[0];
#pragma unroll
        for (int l = 0; l < TNML_NL; ++l) P[l] = fma(a, A.Pp[(size_t)l * NTp + ni], A.P[(size_t)l * NTp + ni]);
    }
    double val = 0.; int cor = 0;
    if (A.mode == LD_MODE_PAP) {
#pragma unroll
        for (int l = 0; l < TNML_NL; ++l) {
            val = fma(P[l], P[l], val);                        // sqr(norm(pv)), fixedL.cc:400
            if (A.Pout) A.Pout[(size_t)l * NTp + ni] = P[l];
        }
        if (lab < 0) val = 0.;
    } else {
        double best = fabs(P[0]); int arg = 0;
#pragma unroll
        for (int l = 0; l < TNML_NL; ++l) {
            const double tgt = l == lab ? 1. : 0.;
            const double d = lab >= 0 ? tgt - P[l] : 0.;       // deltas[t.l] - P
            val = fma(d, d, val);
            if (A.dP) A.dP[(size_t)l * NTp + ni] = d;
            if (A.Pout) A.Pout[(size_t)l * NTp + ni] = P[l];
            const double wgt = fabs(P[l]);
            if (wgt > best) { best = wgt; arg = l; }           // first maximum (util.h:42-57)
        }
        cor = (lab >= 0 && arg == lab) ? 1 : 0;
    }
    res_wave_partials(val, lab, cor, A.mode == LD_MODE_PAP, A.partials + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 12, lane);
}

template <int PS, int PK, int NKA, bool GEN, int NST = 8>
static int fwd_res_go(tnml_ctx* c, const FwdResArgs& a, int grid) {
    const size_t lds = sizeof(double) * FR_LDS_DOUBLES_N(NKA);
    bool& done = c->attr_fr[2 * (GEN ? 1 : 0) + (NST == 4 ? 1 : 0)][NKA];
    if (!done || !GEN) {                              // (the constant-extent form has one slot for its five pacings: set every time, as before)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_res<PS, PK, 0, NKA, GEN, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return tnml_fail(c, "fwd_res: cannot reserve %zu bytes of LDS", lds);
        done = true;
    }
    ProfScope ps(c, KC_FWD_RES);
    hipLaunchKernelGGL((k_fwd_res<PS, PK, 0, NKA, GEN, NST>), dim3(grid), dim3(768), lds, c->stream, a);
    return 0;
}
// the instantiations built for other bonds than 120 x 120: input dimensions up to 4 NKA
static const int fr_nka_list[] = {12, 16, 18, 20, 22, 24, 26, 28, 30};
static int fr_nka_for(int mI) {
    for (int v : fr_nka_list) if (4 * v >= mI) return v;
    return 0;
}
// bond dimensions k_fwd_res serves: input 33..120 (one instantiation per reduction length), output 16..120 (2..15 column tiles of 8 links on a
// pair of workgroups)
bool fwd_res_applies(int mI, int mO) { return mI >= 33 && fr_nka_for(mI) != 0 && mO >= 16 && mO <= 120; }
int launch_fwd_res(tnml_ctx* c, const FwdResArgs& a) {
    if (a.NTp % 256) return tnml_fail(c, "fwd_res: image count not a multiple of 256");
    if ((size_t)TNML_NL * a.EL_lstride * sizeof(double) >= ((size_t)1 << 32)) return tnml_fail(c, "fwd_res: environment larger than 4 GB (32-bit lane offsets)");
    if (!fwd_res_applies(a.mI, a.mO) || a.Np % 16 || a.Np < 2 * a.mO || a.Np > 240 || a.Kp < 2 * a.mI) return tnml_fail(c, "fwd_res: bond %d x %d (packed %d x %d) not served", a.mI, a.mO, a.Kp, a.Np);
    if (!c->cu_count) { hipDeviceProp_t pr; c->cu_count = hipGetDeviceProperties(&pr, c->cfg.device) == hipSuccess ? pr.multiProcessorCount : 256; }
    int grid = c->cu_count / 16 * 16;
    if (c->res_grid > 0 && c->res_grid < grid) grid = c->res_grid / 16 * 16;      // test knob: fewer workgroups -> more rounds each
    if (grid < 16) grid = 16;
    while (grid > 16 && (grid / 2) > a.ntiles) grid -= 16;                        // every pair of workgroups has at least one tile
    // pacing of the GEMM waves (see k_fwd_res): 384 cycles every 8 MFMAs measured best (tools/probe/kbench_res.hip,
    // profiles/r04_probe_fwd_res.txt); option "res_pace" selects the others (120 x 120 bonds only)
    if (a.mI == 120 && a.mO == 120 && a.Kp == 240 && a.Np == 240 && c->fwd_res != 3) {
        switch (c->res_pace) {
            case 1:  TCK((fwd_res_go<0, 1, 30, false>(c, a, grid))); break;      // no pauses
            case 2:  TCK((fwd_res_go<4, 2, 30, false>(c, a, grid))); break;
            case 3:  TCK((fwd_res_go<6, 3, 30, false>(c, a, grid))); break;
            case 4:  TCK((fwd_res_go<4, 1, 30, false>(c, a, grid))); break;
            default: TCK((fwd_res_go<6, 2, 30, false>(c, a, grid))); break;
        }
    } else if (a.Np <= 128) {                                                       // mO <= 64: at most four column tiles per half, four streaming steps per tile
        switch (fr_nka_for(a.mI)) {
            case 12: TCK((fwd_res_go<6, 2, 12, true, 4>(c, a, grid))); break;
            case 16: TCK((fwd_res_go<6, 2, 16, true, 4>(c, a, grid))); break;
            case 18: TCK((fwd_res_go<6, 2, 18, true, 4>(c, a, grid))); break;
            case 20: TCK((fwd_res_go<6, 2, 20, true, 4>(c, a, grid))); break;
            case 22: TCK((fwd_res_go<6, 2, 22, true, 4>(c, a, grid))); break;
            case 24: TCK((fwd_res_go<6, 2, 24, true, 4>(c, a, grid))); break;
            case 26: TCK((fwd_res_go<6, 2, 26, true, 4>(c, a, grid))); break;
            case 28: TCK((fwd_res_go<6, 2, 28, true, 4>(c, a, grid))); break;
            default: TCK((fwd_res_go<6, 2, 30, true, 4>(c, a, grid))); break;
        }
    } else {                                                                        // (fwd_res = 3: the general form on a 120 x 120 bond too -- tests)
        switch (fr_nka_for(a.mI)) {
            case 12: TCK((fwd_res_go<6, 2, 12, true>(c, a, grid))); break;
            case 16: TCK((fwd_res_go<6, 2, 16, true>(c, a, grid))); break;
            case 18: TCK((fwd_res_go<6, 2, 18, true>(c, a, grid))); break;
            case 20: TCK((fwd_res_go<6, 2, 20, true>(c, a, grid))); break;
            case 22: TCK((fwd_res_go<6, 2, 22, true>(c, a, grid))); break;
            case 24: TCK((fwd_res_go<6, 2, 24, true>(c, a, grid))); break;
            case 26: TCK((fwd_res_go<6, 2, 26, true>(c, a, grid))); break;
            case 28: TCK((fwd_res_go<6, 2, 28, true>(c, a, grid))); break;
            default: TCK((fwd_res_go<6, 2, 30, true>(c, a, grid))); break;
        }
    }
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_pfinish(tnml_ctx* c, const PfinishArgs& a) {
    if (a.NTp % 256) return tnml_fail(c, "pfinish: image count not a multiple of 256");
    ProfScope ps(c, KC_PUPDATE);
    hipLaunchKernelGGL(k_pfinish, dim3(a.NTp / 256), dim3(256), 0, c->stream, a);
    HIPCK(c, hipGetLastError());
    return 0;
}
