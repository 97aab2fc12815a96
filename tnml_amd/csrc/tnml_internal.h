// tnml_internal.h -- context and kernel-launch declarations shared by the libtnml.so sources.
//
// Device data layout (all per rank, "image-fastest" structure of arrays, fp32):
//   phi    [N][2][NTp]            local features of every image, site-major
//   label  [NTp]                  int32, -1 for padding images
//   env_j  [L][m_j][NTp]          environment of site j (L = 10 when it carries the Label index)
//   U      [L][mO][NTp]           workspace: feature-GEMM output
//   P, dP  [10][NTp]
//   Zp     [mO][NTp]
// NTp = NT_local rounded up to 256; padding images have zero features, so they contribute
// nothing to any contraction.
//
// Bond tensors / CG vectors live in "M-layout" (fp64 master, fp32 GEMM operand):
//   M[l][k][j], k = 2*x + s  (x = link of the label-free "input" env, s = its site index)
//               j = 2*y + t  (y = link of the "output" side, t = its site index)
//   zero padded to Kp x Np (multiples of 16).
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tnml.h"

#define TNML_NTPAD 256

enum KClass {
    KC_FGEMM_FWD = 0, KC_FGEMM_SHIFT, KC_LABELDOT, KC_ZPRIME, KC_BGEMM, KC_SLABRED, KC_PACK, KC_VEC,
    KC_SMALLGEMM, KC_SVD, KC_ALLREDUCE, KC_PUPDATE, KC_FWD_FUSED, KC_FWD_RES, KC_GRAD_QUAD, KC_COUNT
};
static const char* const kclass_names[KC_COUNT] = {
    "fgemm_fwd", "fgemm_shift", "labeldot", "zprime", "bgemm", "slab_reduce", "pack", "cg_vec",
    "small_gemm", "svd", "allreduce", "p_update", "fwd_fused", "fwd_res", "grad_quad"};

struct EnvSlot {
    void* ptr = nullptr;    // [L][m][NTp], fp32 or fp64 elements (tnml_ctx::env64); null while the environment is spilled to the host
    int m = 0, L = 0;
    int slab = -1, unit = -1;   // where it lives (unit -1: the whole slab, a Label-carrying environment)
    // host tier (option env_budget_mb; the reference's Nbatch / proj_images spill, fixedL.cc:115-120,153,177-178,216,231, with host memory
    // in the place of its disk files): an environment that has been evicted lives in `host` until something needs it again
    char* host = nullptr; size_t host_cap = 0; bool host_pinned = false;
    bool on_host = false;
    hipEvent_t ev = nullptr; bool ev_pending = false;   // a copy back from the host is in flight on the copy stream: the compute stream waits for ev before it touches ptr
    bool built() const { return ptr != nullptr || on_host; }
};
// Environment memory: slabs of one Label-carrying environment ([10][maxm][NTp]); a Label-free environment takes
// one tenth of a slab.  During a sweep the mix of the two kinds changes from ~half/half at the chain ends to
// all Label-free at the centre, at constant total need, so units are recycled inside slabs and hipMalloc is
// off the hot path (DESIGN.md "HBM layout").
struct EnvSlab {
    char* base = nullptr;
    unsigned mask = 0;      // bit k: unit k in use
    hipEvent_t ev = nullptr; bool ev_pending = false;   // a former tenant is still being copied to the host (copy stream): whoever takes a unit of this slab waits for ev first
};

struct SiteT {
    double* a = nullptr;    // ITensor layout [ml][2][mr]([10]), capacity fixed at create
    int ml = 0, mr = 0, L = 1;
    bool set = false;
};

// scalar slots of the device-side CG state (doubles)
enum { SC_COST0 = 0, /* ..9 */ SC_NCORR = 10, SC_PP = 11, SC_RR = 12, /* 13: second |r|^2 slot */ SC_ALPHA = 14, SC_BETA = 15,
       SC_PNORM2 = 16, SC_BNORM2 = 17, SC_PAP = 18, SC_RNORM = 19, SC_COST = 20, SC_CONV = 21 /* host copy of the flag */, SC_CONV_NEXT = 22, SC_NPASS = 23, SC_CONVP = 24 /* ,25: the flag by pass parity */,
       SC_NORMS = 28 /* |newB|^2, |B-newB|^2 */, SC_N = 32 };
// The all-reduce buffer of a rank is [tail (TNML_TAILN) | G (Kp*Np*LB)]: the scalars ride IN FRONT of the gradient, at a fixed place.
//   tail[0..9]  per-label cost partials, [10] #correct, [11] sum |p.v_n|^2 (the pAp pass), [12..14] local (not summed) scalars of
//               tnml_quadcost: |B|^2, |newB|^2, |B - newB|^2 -- written AFTER the reduction
//   tail[16..26] the same cost partials of the "after SVD" quadcost of a bond update, [32..39] the replica fingerprint of the two site
//               tensors the split wrote, as exact integer pieces (see fingerprint_pieces): both are produced at the END of a bond
//               update and ride in the FIRST all-reduce of the next one (or in one small all-reduce when nothing follows)
#define TNML_NSCAL_AR 16   /* live CG scalars at the head of the tail */
#define TNML_TAILN 48
#define TNML_CARRY 16      /* first carried slot */
#define TNML_FPSLOT 32     /* first fingerprint slot (8 doubles) */
#define TNML_CARRYN 24     /* carried doubles: slots 16..39 */
#define TNML_SPECSLOT 28   /* carried: 1 when the deferred check of a speculative split failed (summed over the ranks: every rank rolls back together) */

struct BondPlan {
    int b = -1;
    int kind = 0;              // 0: Label on RE, 1: Label on LE, 2: Label on B
    int mL = 0, mR = 0;
    int mI = 0, mO = 0;        // link dims of input (label-free GEMM side) and output side
    int Kp = 0, Np = 0, LB = 1;
    const void* EI = nullptr; const void* phiI = nullptr;     // input env [mI][NTp], its features (env-typed)
    const void* EX = nullptr; const void* phiO = nullptr;     // other env: [10][mO][NTp] (kind 0/1) or [mO][NTp] (kind 2)
    size_t msize() const { return (size_t)LB * Kp * Np; }
};

struct ProfPending { hipEvent_t e0, e1; int kc; };
// a site tensor the split of a bond update in flight has replaced: its former buffer and bond dimensions stay until the split is verified
struct SiteUndo { int j = 0; double* old = nullptr; int ml = 0, mr = 0; };
struct PendingReport {
    tnml_bond_report rep; double lambda_cost = 0.; hipEvent_t ev = nullptr, ev2 = nullptr; bool fp = false;
    // speculative split (no host synchronisation inside tnml_bond_update_begin): what tnml_bond_update_end needs to finish the report
    // and, when the deferred orthogonality check fails, to roll the bond update back and run it again with the synchronous split
    int b = 0, ha = 0; tnml_sweep_params sp{};
    bool spec = false; int split_n = 0, split_mk = 0; int nundo = 0; SiteUndo undo[2];
    int dn_pairs = 0, cost_rows = 0; bool trace_mirrored = false, carry_direct = false;
};

struct tnml_ctx {
    tnml_config cfg;
    int N = 0, NT = 0, NTp = 0, c0 = 0, maxm = 0;
    hipStream_t stream = nullptr;
    int fused_fwd = 1;               // forward pass as one persistent kernel (kernels_fused.hip): 1 = from 14 336 images per rank on, 0 never, 2 always; env TNML_FUSED_FWD / option "fused_fwd"
    int cg_method = 0;               // per-label variant: 0 = conj (single.h:162-288), 1 = fast_conj (single.h:290-398), 2 = exact (single.h:117-160); option "cg_method"
    double pcut = 1e-8;              // PCut of the exact solver (single.cc:50); tnml_set_option_real "pcut"
    double noise = 0.;               // per-label variant: sweeps.noise() (single.cc:25,222); >= 1e-14 selects the density-matrix split of single.h:648-672; tnml_set_option_real "noise"
    double* noise_ws = nullptr;      // its workspace, allocated with the first such split
    int sytrd_exit = 1;              // rank-adaptive exit of the tridiagonalisation of the split's Gram matrix (eigh.hip); option "sytrd_exit", env TNML_SYTRD_TOL=0 disables
    rocblas_handle blas = nullptr;
    ncclComm_t comm = nullptr;
    struct LocalComm* local = nullptr;   // in-process communicator of ranks sharing one device (local_comm.hip)
    struct IpcComm* ipc = nullptr;       // cross-process one-shot all-reduce over IPC-mapped receive regions (ipc_comm.hip)
    bool multi() const { return comm != nullptr || local != nullptr || ipc != nullptr; }   // the sum over images is also a sum over ranks
    std::string err;
    std::string warn;               // last non-fatal notice (tnml_last_warning): e.g. maxm / minm clamped to the context's maxm by the split
    int64_t bytes = 0;

    void* phi = nullptr;       // [N][2][NTp], env-typed
    int* label = nullptr;      // [NTp]
    void* ones = nullptr;      // [NTp] of 1: the "environment" beyond the chain ends, env-typed
    bool data_set = false;

    std::vector<SiteT> W;      // 1..N
    std::vector<EnvSlot> env;  // 1..N
    std::vector<EnvSlab> slabs;
    size_t small_elems = 0, big_elems = 0;

    // workspaces
    // element type of U/P/dP/Zp/slab follows cfg.dtype (double for TNML_F64, float for TNML_F32)
    void* U = nullptr;         // [10][maxm][NTp]
    void* P = nullptr;         // [10][NTp]
    void* dP = nullptr;        // [10][NTp]
    void* Pp = nullptr;        // [10][NTp]  p*t.v of the last pAp pass (fast CG)
    bool fuse_z = true;        // gradient GEMM builds Z from EL and dP itself instead of a k_zprime pass (env TNML_FUSE_Z=0 disables)
    bool fast_cg = true;       // P <- P + a (p*t.v) instead of re-running the forward GEMM (env TNML_FAST_CG=0 disables)
    // The per-image outputs P_n = W.Phi(x_n) belong to the network, not to the bond they are evaluated at: the "after SVD"
    // quadcost of one bond update leaves in P/dP exactly what the first gradient evaluation of the next bond update would
    // recompute with its own forward GEMM + label dot.  p_valid marks P/dP as current; anything that changes W, the data or
    // P itself clears it (env TNML_REUSE_P=0 disables the shortcut).
    bool reuse_p = true, p_valid = false;
    long env_budget_bytes = 0;           // option env_budget_mb: cap on the environment slabs held on the device (0: none); beyond it environments spill to host memory
    int env_protect[4] = {0, 0, 0, 0};   // sites whose environments must stay on the device (the operands of the operation in flight)
    long env_spills = 0, env_fetches = 0, env_prefetches = 0;
    hipStream_t copy_stream = nullptr;   // host tier: evictions and prefetches run beside the compute stream
    hipEvent_t ev_compute = nullptr;     // "everything enqueued on the compute stream so far" (recorded before an eviction starts)
    int env_async = 1;                   // option env_async: 0 = every copy of the host tier on the compute stream (the simple form)
    int bf16_once = 1;                   // option bf16_once: the bf16 modes' forward feature GEMM with its operands converted once per bond / per launch (kernels_bf16e.hip; 0: k_fgemm_bf16, which rounds while staging)
    unsigned short* ebt = nullptr; unsigned short* mbt = nullptr; size_t ebt_cap = 0, mbt_cap = 0;   // bf16 copies of the Label-free environment [planes][NTp][KH] and of the bond vector [planes][2][2][QP][KH]
    const void* ebt_src = nullptr; int ebt_m = 0; unsigned long ebt_epoch = 0, env_epoch = 1;        // what the environment copy was made from; env_epoch advances with every write of an environment
    bool attr_bf16e = false;
    int bf16_grad = 1;                   // option bf16_grad: in the bf16 modes the gradient GEMM runs on the bf16 pipe too (0: the fp32 kernel, as through round 3)
    int small_gemm = 1;                  // option small_gemm: the split's products on k_dgemm_small (0: rocBLAS, as through round 4)
    int bgs_chol = 1;                    // option bgs_chol: block Gram-Schmidt Cholesky QR for 128 < kept columns <= 384 (0: rocSOLVER dpotrf + dtrsm)
    int coll_depth = 0;                  // >0 inside an entry point that every rank calls in step (tnml_fail then aborts an in-process communicator)
    int comm_timeout_s = 120;            // option comm_timeout_s: how long a rank of an in-process communicator waits for its peers
    int opt_fg64_cfg = 0, opt_ldot_cfg = 0;   // kernel-instantiation overrides (0: chosen by the image count)
    void* Zp = nullptr;        // [maxm][NTp]
    float* Mf = nullptr;       // fp32 GEMM operand, M-layout, capacity 10*Kmax*Kmax (env shifts, F32 mode)
    void* slab = nullptr;      // split-K partial slabs
    size_t slab_bytes = 0;
    bool f64() const { return cfg.dtype == TNML_F64 || cfg.dtype == TNML_F64_E32; }   // fp64 MFMA arithmetic
    int bf16() const { return cfg.dtype == TNML_BF16 ? 1 : (cfg.dtype == TNML_BF16X3 ? 2 : 0); }   // forward feature GEMM on the bf16 matrix pipe: 1 plain, 2 hi + lo split
    bool env64() const { return cfg.dtype == TNML_F64; }
    bool single() const { return cfg.mode == TNML_MODE_SINGLE; }
    int nl() const { return single() ? 1 : TNML_NL; }
    int target() const { return single() ? cfg.target_label : -1; }        // fp64 environment / feature storage
    size_t esz() const { return f64() ? 8 : 4; }
    size_t eesz() const { return env64() ? 8 : 4; }
    double* partials = nullptr;  // [nblk][16]
    double* partials2 = nullptr; // second set: the output update that rides in k_cg_step2 writes here while the launch still reads the pAp pass's partials
    bool defer_slab = false;     // the gradient GEMM leaves its split-K slabs unreduced: the CG vector kernel that consumes G sums them (one rank; set inside cgrad_device only)
    int slab_pending = 0;        // > 0: c->slab holds that many unreduced slabs of the last gradient GEMM
    bool rr_from_part = false;   // |r|^2 of the CG's start is still in k_cg_init1's partial sums (no k_cg_init2 launch)
    int partial_cap = 0;
    int part_n = 0;              // rows of `partials` the last forward pass / output update wrote
    bool fold_reduce = true;     // one rank: the CG step kernels sum those rows themselves (no k_reduce_partials launch inside a CG pass); option "fold_reduce"
    double *vB = nullptr, *vR = nullptr, *vP = nullptr;   // CG vectors, M-layout fp64
    double* arbuf = nullptr;   // the all-reduce buffer [tail | G]
    double* tail = nullptr;    // = arbuf
    double* vG = nullptr;      // = arbuf + TNML_TAILN: gradient / A p, M-layout
    double* locals = nullptr;  // [2][16] device: the local scalars (|B|^2, |newB|^2, |B - newB|^2) of the bond updates in flight
    int merged_cg = 1;         // (1: with a communicator, 2: always, 0: never) CG passes with ONE all-reduce each: A p = sum_n (p.v_n) v_n is formed from the pAp pass's outputs before alpha is
                               // known and rides with sum |p.v_n|^2; the residual follows r <- r - alpha (A p + lambda p) (the structure of the
                               // reference's own fast_cgrad, single.h:347-379).  Needs fast_cg.  env TNML_MERGED_CG / option "merged_cg"
    bool defer_tail = true;    // multi-rank: the after-SVD cost partials and the replica fingerprint ride in the next bond update's first all-reduce
    int carry_slot = -1;       // pending report whose carried slots have not been reduced yet
    long allreduce_calls = 0, bcast_calls = 0;  // collectives entered: sum all-reduces (the payload) and broadcasts (rank 0's eigenvalues), for tests and the bench line
    double* scal = nullptr;    // device scalars [SC_N]
    double* cgtrace = nullptr; // device CG trace [TNML_MAX_PASS][4] = pAp, alpha, cost, |r|
    double* vpart = nullptr;   // per-workgroup partial sums of the CG vector kernels [256][2]
    int cg_pass = 0;           // CG pass being issued (selects the parity slot of the convergence flag)
    int rr_slot = 0;           // which of scal[SC_RR], scal[SC_RR+1] holds the current |r|^2
    double* h_scal = nullptr;  // pinned host mirror
    double *tB = nullptr, *tB2 = nullptr;   // bond tensors in ITensor layout (fp64)
    size_t mcap = 0;           // capacity (elements) of M-layout vectors / bond tensors
    // svd workspaces (fp64)
    size_t sM_cap = 0;
    double *sM = nullptr, *sG = nullptr, *sD = nullptr, *sE = nullptr, *sF = nullptr;
    double *sE2 = nullptr, *sTau = nullptr, *sV = nullptr, *sC = nullptr;   // eigh.hip: subdiagonal, tau, reflectors, tridiagonal eigenvectors
    double *sW = nullptr, *sScr = nullptr, *sS = nullptr, *sCm = nullptr, *sQ1 = nullptr, *sDev = nullptr;   // own tridiagonal eigensolver + Newton-Schulz polish
    void* mc_xbuf = nullptr;   // exchange buffer of the multi-workgroup tridiagonalisation (eigh_mc.hip), contexts with maxm > 120 only
    unsigned mc_epoch = 0;
    int mc_spin_max = -1;      // polls before a waiting thread of k_sytrd_mc gives up (-1: default; option "mc_spin_max", 0 in the fallback test)
    bool attr_sr[32] = {false}, attr_fr[4][32] = {{false}};   // LDS attribute set (per device = per context) for k_shift_res<NKS> / k_fwd_res<.., NKA, GEN, NST>: [2 GEN + (NST == 4)][NKA]
    int res_pace = 0;                // pacing of the GEMM waves of k_fwd_res (0: default; option "res_pace")
    int fwd_res = 1;                 // forward pass on k_fwd_res (kernels_res.hip): 1 = from 7 680 images per rank on, 0 never, 2 always; option "fwd_res"
    int shift_res = 1;               // Label-carrying environment shift on k_shift_res (kernels_res.hip): 1 = from 7 680 images per rank on, 0 never, 2 always; option "shift_res"
    int res_grid = 0;                // test knob: workgroups of the resident-operand kernels (0: one per CU)
    int grad_quad = 1;               // gradient GEMM on k_grad_quad (kernels_grad.hip; m = 120, fp64 storage, Label on an environment): 1 = from 4 096 images per rank on, 0 never, 2 always; option "grad_quad", env TNML_GRAD_QUAD
    bool attr_gq = false, attr_gp = false;
    int grad_pair = 1;               // bonds up to 64 x 64: the pair form of k_grad_quad (128 x 128 tile grid, two workgroups): 1 = for grad_pair_min^2 <= mI mO <= grad_pair_max^2 (and whenever grad_quad = 2 forces the kernel), 0 = never (the quad form when forced)
    int grad_pair_min = 33, grad_pair_max = 56;      // sqrt(mI mO) range the pair form takes unforced
    int bgemm_per = 0;               // probe knob (TNML_BGEMM_PER): images per slab of the gradient GEMM, in units of 32 (0: derived from bgemm_wgs)
    int bgemm_wgs = 0;               // workgroups the gradient GEMM aims at when it cuts the image range into slabs (0: per-shape default; option "bgemm_wgs")
    unsigned* counters = nullptr;    // [16] device: arrival counters of the "last workgroup reduces" kernels (zero between launches)
    double* Ppart = nullptr;         // [2][10][NTp]: per-half outputs of k_fwd_res
    bool attr_sytrd = false, attr_invit = false, attr_fused = false;   // per-device function attributes set (a process may drive several devices)
    int cu_count = 0;
    // Speculative split (option spec_split, default on): when the truncation cannot change the outcome (minm >= the columns the split may keep)
    // the new bond dimension is known without the eigenvalues, so the split is enqueued WITHOUT its host synchronisation; eigenvalues and
    // check values are mirrored into pinned host memory by the kernels that produce them and read by tnml_bond_update_end.  The new site
    // tensors go to spare buffers; a failed check rolls the sites back and repeats the bond update with the synchronous split.
    int spec_split = 1; bool force_safe = false; long spec_redos = 0, spec_splits = 0; int debug_fail_split = -1;
    long spec_splits_total = 0; double redo_ms = 0.; std::vector<std::pair<hipEvent_t, hipEvent_t>> redo_events;   // tnml_split_stats
    std::vector<double*> spare_small, spare_big;   // spare site-tensor buffers (capacity 2 maxm^2, x 10 for the Label site)
    double* hrep = nullptr;                        // pinned: [2 slots][hrep_stride] = eigenvalues + check values of a speculative split | CG scalars + trace | norm partials | after-SVD scalars
    double* hmir = nullptr;                        // != nullptr while a bond update is being enqueued: the [scal | trace] mirror of its slot (the CG step kernels write it)
    double* hcost = nullptr;                       // pinned: [2 slots][partial_cap][12]: one rank, the per-block partial sums of the after-SVD quadcost go there (the host adds them)
    int last_dn_pairs = 0;                         // partial pairs the last launch_diffnorm_host wrote
    size_t hrep_stride = 0;
    double svd_last_dev0 = 0., svd_last_dev1 = 0.;   // max|Q^T Q - I| before the 1st / 2nd polish step of the last split
    long svd_fallbacks = 0, svd_cholqr = 0;
    int svd_print = -2, svd_calls = 0, svd_dumped = 0;   // debugging aids of the split, read once in tnml_create (TNML_SVD_PRINT, TNML_SVD_DUMP) / option "svd_print"
    std::string svd_dump;
    double last_bnorm = 0.;         // |B| of the last quadcost
    int* sInfo = nullptr;
    unsigned long long* fprint = nullptr;   // [2] device: fingerprint of replicated tensors (and its complement)
    int check_replicas_mode = 1;            // 1: a mismatch is an error (checked with the deferred tail: no extra collective); 2: checked at once, inside the bond update
                                            //    and before the environment shift; on a mismatch rank 0's two site tensors are re-broadcast and the event is counted
    long replica_repairs = 0;
    int debug_nudge_rank = -1;              // test hook: this rank's copy of W.A(b) is moved by one ulp after every split
    bool check_replicas = true;             // multi-rank: compare the fingerprints of W[b], W[b+1] after every bond update (env TNML_CHECK_REPLICAS=0 disables)
    int svd_n = 0;

    BondPlan plan;
    int currb = -1;
    PendingReport pend[2];     // bond updates begun and not yet ended (tnml_bond_update_begin / _end)
    int pend_tail = 0, pend_count = 0;
    bool tail_zeroed = false;  // the pack kernel of the running bond update has cleared the scalar tail behind G

    // profiling
    bool prof = false;
    unsigned prof_mask = 0xffffffffu;   // kernel classes that are timed while prof is on
    int64_t prof_launches[KC_COUNT] = {0};
    double prof_ms[KC_COUNT] = {0};
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> prof_free;
};

int tnml_fail(tnml_ctx* c, const char* fmt, ...);
void prof_begin(tnml_ctx* c, int kc, hipEvent_t* e0, hipStream_t st = nullptr);
void prof_end(tnml_ctx* c, int kc, hipEvent_t e0, hipStream_t st = nullptr);
void prof_resolve(tnml_ctx* c);

#define HIPCK(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return tnml_fail((c), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define TCK(expr) do { int r_ = (expr); if (r_) return r_; } while (0)
// a host synchronisation with the stream, then the transport's verdict: a collective of the cross-process one-shot transport that timed
// out on the device has left NaNs in its buffer (never unsummed values) and a status word -- whatever the host reads after this
// line has either been summed over every rank or the call fails here
#define SYNCK(c, st) do { HIPCK((c), hipStreamSynchronize(st)); TCK(ipc_comm_check(c)); } while (0)

// RAII-less profiling bracket around a group of launches of one kernel class
struct ProfScope {
    tnml_ctx* c; int kc; hipEvent_t e0 = nullptr; hipStream_t st;
    bool on;
    ProfScope(tnml_ctx* c_, int kc_, hipStream_t st_ = nullptr) : c(c_), kc(kc_), st(st_), on(c_->prof && ((c_->prof_mask >> kc_) & 1u)) { if (on) prof_begin(c, kc, &e0, st); }
    ~ProfScope() { if (on) prof_end(c, kc, e0, st); }
};

// ---- kernels_gemm.hip ---------------------------------------------------------------------
struct FgemmArgs {
    const float* EI; size_t EI_lstride; int mI;      // input env [L?][mI][NTp]
    const float* phiI;                                // [2][NTp]
    const float* M; size_t M_lstride; int Kp, Np;     // [L?][Kp][Np] row-major
    const float* phiO;                                // [2][NTp], null when the output has no site index
    float* out; size_t out_lstride; int mO;           // [L][mO][NTp]
    int NTp; int L;
};
int launch_fgemm(tnml_ctx* c, const FgemmArgs& a);
// ---- kernels_bf16e.hip: the forward feature GEMM of the bf16 modes with operands converted once ----
size_t bf16e_env_elems(int maxm, int NTp, int split);
size_t bf16e_m_elems(int maxm, int split);
int launch_fgemm_bf16e(tnml_ctx* c, const float* EI, int mI, const float* phiI, const double* vec, int Kp, int Np, const float* phiO, float* out, int mO);

struct BgemmArgs {
    const float* EI; int mI; const float* phiI;      // A operand rows: X[n][2a+s]
    const float* Zq; int mO; const float* phiO;      // B operand rows: w[n]*phiO[t][n]*Zq[q][n]
    const float* w; size_t w_lstride;                // per-image weight [L][NTp] or null
    int Kp, Np, NTp, L;
    int bf16 = 0;                                    // 1 / 2: operands rounded to bf16 (plain / hi + lo) on v_mfma_f32_16x16x32_bf16 (TNML_BF16, TNML_BF16X3)
};
// writes split-K partial slabs then reduces them (fixed order, fp64) into G[L][Kp][Np]
int launch_bgemm(tnml_ctx* c, const BgemmArgs& a, double* G);

// fp64-MFMA flavour (operands converted/expanded to fp64 while staging into LDS)
struct Fgemm64Args {
    const void* EI; size_t EI_lstride; int mI;        // environment: fp32 (env64 == 0) or fp64 elements
    const void* phiI;
    const double* M; size_t M_lstride; int Kp, Np;    // the fp64 CG vector itself (M-layout)
    const void* phiO;                                 // null: no output site index (strict-mode env shift)
    double* out; size_t out_lstride; int mO;
    int NTp; int L;
    int env64;
    int out32 = 0;                                    // shift form only: out is a float array (fp32-stored environments, TNML_F64_E32)
    int n_off = 0, n_cnt = 0;                         // image range of this launch (n_cnt = 0: all NTp); multiples of 128
    hipStream_t st = nullptr;                         // nullptr: the context's stream
    int kclass = -1;                                  // profiling class override
};
int launch_fgemm64(tnml_ctx* c, const Fgemm64Args& a);
struct Bgemm64Args {
    const void* EI; int mI; const void* phiI;                            // env-typed (fp32 / fp64 by env64)
    const double* Zq64; const void* Zq32; int mO; const void* phiO;      // exactly one of Zq64 / Zq32 (an env) / EL
    const void* EL; size_t EL_lstride; const double* dPz;                // fused: Z = sum_l EL[l] * dPz[l]
    const double* w; size_t w_lstride;
    int Kp, Np, NTp, L;
    int env64;
};
int launch_bgemm64(tnml_ctx* c, const Bgemm64Args& a, double* G);
void launch_slab_reduce64(tnml_ctx* c, const double* slab, double* G, size_t n, int nsplit);
// ---- kernels_grad.hip: the gradient GEMM with the accumulators resident in a quad of workgroups, three MFMA-issuing waves per SIMD ----
bool grad_quad_applies(const tnml_ctx* c, const Bgemm64Args& a);
int launch_grad_quad(tnml_ctx* c, const Bgemm64Args& a, double* G);

// ---- kernels_stream.hip -------------------------------------------------------------------
enum { LD_MODE_COST = 0, LD_MODE_PAP = 1, LD_MODE_FWD = 2 };
struct LdotArgs {
    const void* A; size_t A_lstride;    // label-carrying operand [10][mq][NTp] (elements)
    const void* Bv;                     // label-free operand [mq][NTp]
    int a_is_env;                       // 1: A is an fp32 environment and Bv the GEMM output; 0: the reverse
    int nl = TNML_NL;                   // label extent actually present (1 in TNML_MODE_SINGLE)
    int target = -1;                    // TNML_MODE_SINGLE: y_n = [label_n == target]; -1: targets delta_{l,label_n}
    int mq, NTp;
    const int* label;
    void* P; void* dP;                  // [10][NTp] in the context's arithmetic type
    int mode;
    int nt = 0;                         // non-temporal loads of A (set by launch_labeldot)
    int blk_off = 0;                    // first image block of this launch (blocks of 64 * images-per-lane)
};
// partial sums -> scal_out[0..11] (device); deterministic
int launch_labeldot(tnml_ctx* c, const LdotArgs& a, double* scal_out, bool reduce = true);
// the two halves of launch_labeldot for a split launch: blocks [blk_off, blk_off + nblk) on `st`, then the reduction of ALL partials
int launch_labeldot_blocks(tnml_ctx* c, const LdotArgs& a, int blk_off, int nblk, hipStream_t st, int kclass, int form = 0);   // form 1: 128-image blocks, 2: 64-image blocks, 0: by image count
int launch_labeldot_reduce(tnml_ctx* c, int nblk_total, double* scal_out, int only_sum = 0);   // only_sum: write slot 11 alone (the pAp pass must leave the cost partials in place)
bool labeldot_streaming(const tnml_ctx* c, int NTp);   // the 128-images-per-workgroup form is in use
int launch_pupdate(tnml_ctx* c, const double* alpha_dev, double* scal_out, bool reduce = true);     // uses c->nl(), c->target(); !reduce: the partial sums stay in c->partials[c->part_n][12]
int launch_zprime(tnml_ctx* c, const void* EL, size_t lstride, const void* dP, void* Z, int mq, int NTp);
int launch_features_u8(tnml_ctx* c, const uint8_t* d_pix, int N, int NT, int NTp, void* phi);

// ---- kernels_fused.hip ----
struct FwdFusedArgs {
    const double* EI; int mI; const double* phiI;     // Label-free input environment [mI][NTp] and its site features [2][NTp]
    const double* M; int Kp, Np;                      // bond matrix, M-layout [Kp][Np]
    const double* phiO;                               // output-site features [2][NTp]
    const double* EL; size_t EL_lstride;              // Label-carrying environment [10][mO][NTp]
    int mO, NTp, ntiles;                              // ntiles = NTp / 64
    const int* label;
    double* P; double* dP;                            // [10][NTp], either may be null
    int mode;                                         // LD_MODE_*
    double* partials;                                 // [ntiles][12]
};
int launch_fwd_fused(tnml_ctx* c, const FwdFusedArgs& a);

// ---- kernels_res.hip: the small operand resident in registers (m = 120, fp64 storage, Label on an environment) ----
struct FwdResArgs {
    const double* EI; const double* phiI;             // Label-free input environment [120][NTp], its site features [2][NTp]
    const double* M;                                  // bond matrix, M-layout [240][240]
    const double* phiO;                               // output-site features [2][NTp]
    const double* EL; size_t EL_lstride;              // Label-carrying environment [10][120][NTp]
    int NTp, ntiles;                                  // ntiles = NTp / 32
    double* Ppart;                                    // out: [2][10][NTp], the label dot over the output links of each half
    int mI = 120, mO = 120, Kp = 240, Np = 240;       // bond dimensions and the M-layout extents Kp = ru16(2 mI), Np = ru16(2 mO)
    long long* dbg = nullptr;                         // probe builds: per-wave cycle counters of workgroup 0
};
bool fwd_res_applies(int mI, int mO);
int launch_fwd_res(tnml_ctx* c, const FwdResArgs& a);
struct PfinishArgs {
    int npart;                                        // 2: P = Ppart[0] + Ppart[1]; 0: P = P + alpha Pp (fast CG update)
    const double* Ppart;                              // [2][10][NTp]
    const double* P; const double* Pp; const double* alpha;
    const double* conv;                               // device flag (may be null): non-zero -> the launch does nothing
    const int* label; int NTp;
    double* Pout; double* dP;                         // [10][NTp], either may be null
    int mode;                                         // LD_MODE_COST / LD_MODE_PAP
    double* partials;                                 // [NTp / 64][12]
    unsigned* counter;                                // zero before and after the launch
    double* out; int only_sum;                        // [12] sums over all images (only_sum: slot 11 alone)
};
int launch_pfinish(tnml_ctx* c, const PfinishArgs& a);
struct ShiftResArgs {
    const double* EI; size_t EI_lstride;              // Label-carrying input environment [L][mI][NTp]
    const double* phiI;                               // features of the absorbed site [2][NTp]
    const double* M;                                  // packed site matrix [Kp][Np] (k = 2 x + s, j = y), zero padded
    double* out; size_t out_lstride; int mO;          // [L][mO][NTp]
    int NTp, L;
    int mI, Kp, Np;                                   // input bond dimension (33..120) and the packed extents: Kp = ru16(2 mI), Np = ru16(mO) <= 128
    int ntiles = 0;                                   // set by the launcher: L * NTp / 64
};
bool shift_res_applies(int mI, int mO);
int launch_shift_res(tnml_ctx* c, const ShiftResArgs& a);

// ---- kernels_small.hip --------------------------------------------------------------------
struct PackDesc {       // M[l][2x+s][TO==2 ? 2y+t : y] <-> T[off + x*sx + s*ss + y*sy + t*st + l*sl]
    int nx, ny, TO, L;
    long sx, ss, sy, st, sl;
    int Kp, Np;
};
// C = op(A) op(B) (alpha 1, beta 0) issued as a strided batch over `strips` column strips of C: at the few-hundred-square sizes of
// the split rocBLAS picks 128 x 64 macro tiles and runs on 2-8 workgroups; the strips give it more, smaller ones
// (tools/probe/probe_gemm_strips.hip: Gram 240^3 16.2 -> 10.0 us with 4 strips, U^T M 16.5 -> 10.0 and A_b A_{b+1} 10.0 -> 6.3 with 2).
static inline rocblas_status dgemm_strips(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int M, int N, int K,
                                          const double* A, int lda, const double* B, int ldb, double* C, int ldc, int strips) {
    const double one = 1.0, zero = 0.0;
    if (strips <= 1 || N % strips || N / strips < 16) return rocblas_dgemm(h, ta, tb, M, N, K, &one, A, lda, B, ldb, &zero, C, ldc);
    const int ns = N / strips;
    const rocblas_stride sb = tb == rocblas_operation_none ? (rocblas_stride)ns * ldb : (rocblas_stride)ns;
    return rocblas_dgemm_strided_batched(h, ta, tb, M, ns, K, &one, A, lda, 0, B, ldb, sb, &zero, C, ldc, (rocblas_stride)ns * ldc, strips);
}
int launch_pack(tnml_ctx* c, const PackDesc& d, const double* T, double* Md, float* Mf, double* zero = nullptr, int nzero = 0);   // either output may be null; zero[0..nzero) is cleared on the way
int launch_unpack(tnml_ctx* c, const PackDesc& d, const double* Md, double* T);
int launch_cvt(tnml_ctx* c, const double* src, float* dst, size_t n);
int launch_bond_form(tnml_ctx* c, const SiteT& A1, const SiteT& A2, double* B);            // B = A1*A2, ITensor layout
// CG vector algebra on device scalars (single-block kernels)
int launch_cg_init(tnml_ctx* c, size_t n, double lambda, double cconv0);   // cconv0 < 0: no entry check          // r = G - lambda B ; p = r ; RR = |r|^2
int launch_cg_step(tnml_ctx* c, size_t n, double lambda, int pass, bool merged = false, const double* pp_part = nullptr, int npp = 0, bool with_update = false);   // pp_part: column 11 of the pAp pass's per-block partial sums, not reduced yet          // pAp, alpha, B += alpha p (merged: also the cost of the previous pass)
int launch_cg_resid(tnml_ctx* c, size_t n, double lambda, double cconv, int pass, bool merged = false, const double* cost_part = nullptr, int ncp = 0);   // merged: G holds A p, residual by recurrence
int launch_cg_fast_resid0(tnml_ctx* c, size_t n, int pass);      // fast_conj: G <- r - a*G before launch_cg_resid   // nr, beta, r, cost, conv, p
int launch_sqnorm(tnml_ctx* c, const double* x, size_t n, double* out);    // out[0] = |x|^2
int launch_diffnorm(tnml_ctx* c, const double* x, const double* y, size_t n, double* out2, int nout = 2);
int launch_diffnorm_host(tnml_ctx* c, const double* x, const double* y, size_t n, double* part_host, int cap_pairs);   // partial pairs to pinned memory, summed by the host  // out2[0]=|x|^2, out2[1]=|x-y|^2 (nout = 3: |x|^2, |x|^2, |x-y|^2)
int launch_fill_f32(tnml_ctx* c, float* p, float v, size_t n);
int launch_fill_f64(tnml_ctx* c, double* p, double v, size_t n);
int launch_nudge(tnml_ctx* c, double* p);
int launch_fingerprint(tnml_ctx* c, const double* x, size_t n, unsigned long long salt, unsigned long long* acc, bool reset);
int launch_fingerprint_pieces(tnml_ctx* c, const unsigned long long* acc, double* out8);   // 16-bit pieces p_i and p_i^2 of the 64-bit fingerprint: sums over ranks stay exact

// ---- kernels_sgemm.hip: the few-hundred-square fp64 products of the split, one wave per output tile ----
struct SmallGemmArgs {
    const double* A; int lda; const double* B; int ldb; double* C; int ldc;     // column-major; C = op(A) op(B), M x N, reduction length K
    int M, N, K; int ta, tb;
    int bmode = 0; double* dev = nullptr;      // bmode 1: op(B) = 1.5 I - 0.5 B (B symmetric, K == N), dev[0] = max |B - I| (atomic max: zero it first)
    // a side job of tile (0, 0) (speculative split): the four check values of the split to their pinned host mirror, and bad[0] = 1 when they fail
    const double* chk_src = nullptr; double* chk_host = nullptr; double* chk_bad = nullptr;
};
int launch_dgemm_small(tnml_ctx* c, const SmallGemmArgs& a);
int launch_split_check_mirror(tnml_ctx* c, const double* src, double* host, double* bad);
// C = op(A) op(B) at the sizes of the split: the in-house kernel up to 4e7 multiply-adds, rocBLAS (as `strips` column strips) beyond
int split_gemm(tnml_ctx* c, bool ta, bool tb, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc, int strips, const SmallGemmArgs* chk = nullptr);   // chk: its chk_* fields ride along (or get a launch of their own)

// ---- eigh.hip -----------------------------------------------------------------------------
int eigh_tridiagonalize(tnml_ctx* c, const double* A, int n, double* D, double* E, double* tau, double* V, double psd_tol = 0.);   // tau: n doubles, tau[n-1] = number of reflectors
int eigh_tridiag_eig(tnml_ctx* c, const double* D, const double* E, int n, double* W, int mk, double* Z, int ldz, double* scratch, double* W_host = nullptr);      // eigh_tri.hip; W_host: pinned mirror of W (may be null)
int eigh_ns_matrix(tnml_ctx* c, const double* S, double* Cm, int m, double* dev);
int eigh_chol_rinv(tnml_ctx* c, const double* S, int m, double* Rinv, double* flag, int zero_prev = 0);   // m <= 128; zero_prev: also clears flag[-1]
#define TNML_CHOL_MAXM 128
#define TEIG_SCRATCH_DOUBLES 5120        // eigh_tridiag_eig scratch (n <= 1 024)
// ---- eigh_mc.hip: tridiagonalisation on a cluster of workgroups, 240 < n <= 1 024
size_t eigh_mc_xbuf_bytes();
int eigh_mc_max_n();
int eigh_mc_tridiagonalize(tnml_ctx* c, hipStream_t st, const double* A, int n, double* D, double* E, double* tau, double* V, double psd_tol,
                           void* xbuf, unsigned* epoch, long long* dbg = nullptr, int xp = 0, int spin_max = -1);
int eigh_mc_workgroups(int n);
const void* eigh_mc_status_ptr(const void* xbuf);
int eigh_backtransform(tnml_ctx* c, const double* V, const double* tau, int n, const double* Z, int ldz, double* U, int ldu, int ncols, hipStream_t st = nullptr);   // Z == nullptr: U = H_0 ... H_{n-2}

// ---- ipc_comm.hip ----
void ipc_comm_release(tnml_ctx* c);
int ipc_comm_exchange(tnml_ctx* c, double* buf, size_t count, int op);   // 0 sum, 1 broadcast from rank 0; in stream order, never blocks the host
int ipc_comm_check(tnml_ctx* c);                                       // non-zero (and an error message) when a collective timed out
int ipc_comm_mem_kind(const tnml_ctx* c);

// ---- local_comm.hip ----
void local_comm_release(tnml_ctx* c);
int local_comm_size(const tnml_ctx* c);
int local_comm_mode(const tnml_ctx* c);     // 0 none, 2 staging buffer on one device, 3 one-shot peer write
void local_comm_abort(tnml_ctx* c);
void local_comm_set_timeout(tnml_ctx* c, int seconds);
int local_comm_exchange(tnml_ctx* c, double* buf, size_t count, int op);   // 0 sum, 1 broadcast from rank 0

// rank 0's values to every rank, in stream order (no-op without a communicator)
int bcast_rank0(tnml_ctx* c, double* buf, size_t count);
int allreduce_sum(tnml_ctx* c, double* buf, size_t count);      // sum over the ranks, in stream order (no-op on one rank)
int ctx_alloc_doubles(tnml_ctx* c, double** p, size_t n);       // device memory owned by the context (counted in tnml_device_bytes)

// ---- svd.hip ------------------------------------------------------------------------------
int svd_split_device(tnml_ctx* c, const double* B_it, int b, int ha, double cutoff, int maxm, int minm,
                     double* truncerr, int* newm, double* sv_host, int* nsv, int spec_slot = -1);   // spec_slot >= 0: may run without its host synchronisation (see tnml_ctx::spec_split)

// ---- wave64 DPP helpers (device) ----
// quad-lane exchange of a double through DPP (lanes 4q..4q+3 hold the 4 column strips of one block)
template <int CTRL>
static __device__ __forceinline__ double dpp_quad(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
static __device__ __forceinline__ double dpp_masked(double x) {       // rows outside ROWMASK receive 0
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// wave64 sum broadcast to every lane, all in DPP (a ds_bpermute shuffle costs an LDS round trip per
// step, ~12 of them per reduction; this sits on the serial path of every Householder step)
static __device__ __forceinline__ double wave_sum(double x) {
    x += dpp_quad<0xB1>(x);                 // quad_perm [1,0,3,2]
    x += dpp_quad<0x4E>(x);                 // quad_perm [2,3,0,1]
    x += dpp_quad<0x141>(x);                // row_half_mirror: 8-lane sums
    x += dpp_quad<0x140>(x);                // row_mirror: 16-lane (row) sums in every lane
    x += dpp_masked<0x142, 0xA>(x);         // row_bcast:15 -> rows 1,3 += row 0,2
    x += dpp_masked<0x143, 0xC>(x);         // row_bcast:31 -> rows 2,3 += rows 0..1
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

// Workgroup partial sums [12]: cost bucket of every label (0..9), number correct (10), plain sum (11, the
// <p|A|p> mode).  Called by whole waves (all 64 lanes active): a fixed DPP tree inside the wave, lane 0
// leaves the wave's 12 values in s_part[wave][12]; after a barrier sum_wave_partials adds the waves in
// order -> deterministic for a given launch shape.  (The serial 128-entry LDS walk this replaces
// was a third of k_pupdate.)
static __device__ __forceinline__ void wave_bucket_partials(double val, int lab, int cor, bool pap, double* s_part, int wave, int lane) {
    if (pap) {
        const double s = wave_sum(val);
        if (lane < 12) s_part[wave * 12 + lane] = lane == 11 ? s : 0.;
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
    if (lane < 12) s_part[wave * 12 + lane] = mine;
}
static __device__ __forceinline__ void sum_wave_partials(const double* s_part, int nwaves, double* out, int tid) {
    if (tid < 12) {
        double s = 0.;
        for (int w = 0; w < nwaves; ++w) s += s_part[w * 12 + tid];
        out[tid] = s;
    }
}



// Fast CG (tnml_ctx::fast_cg): B*t.v is linear in B, so after B <- B + a p the model outputs are P <- P + a (p*t.v) with p*t.v
// already computed by the pAp pass (the idea of the reference's own single.h:290-398 fast_cgrad).  Recomputes dP, the per-label
// cost partials and the argmax count of one unit of LD_IMGS images (two waves; tid = lane within the unit) -> partials[unit][12].
// Used by k_pupdate (kernels_stream.hip) and by the update half of k_cg_step2 (kernels_small.hip).
#define LD_IMGS 128     // images per unit
template <typename T>
static __device__ __forceinline__ void pupdate_unit(T* __restrict__ P, const T* __restrict__ Pp, T* __restrict__ dP, const int* __restrict__ label, int NTp,
                                                    T a, double* __restrict__ partials, int nl, int target, double* s_part, int tid, int unit) {
    const int ni = unit * LD_IMGS + tid;
    const int lab = label[ni];
    T val = 0; T best = 0; int arg = 0; T p0 = 0;
#pragma unroll
    for (int l = 0; l < TNML_NL; ++l) {
        if (l < nl) {
            const T p = fma(a, Pp[(size_t)l * NTp + ni], P[(size_t)l * NTp + ni]);
            P[(size_t)l * NTp + ni] = p;
            const T tgt = target < 0 ? (l == lab ? (T)1 : (T)0) : (lab == target ? (T)1 : (T)0);
            const T d = (lab >= 0) ? (tgt - p) : (T)0;
            dP[(size_t)l * NTp + ni] = d;
            val = fma(d, d, val);
            const T wgt = fabs(p);
            if (l == 0) { best = wgt; p0 = p; } else if (wgt > best) { best = wgt; arg = l; }
        }
    }
    const int cor = target < 0 ? ((lab >= 0 && arg == lab) ? 1 : 0) : ((lab >= 0 && ((p0 > (T)0.5) == (lab == target))) ? 1 : 0);
    wave_bucket_partials((double)val, lab, cor, false, s_part, tid >> 6, tid & 63);
    __syncthreads();
    sum_wave_partials(s_part, LD_IMGS / 64, partials + (size_t)unit * 12, tid);
}
