/*
 * tnml.h -- C-ABI of the MI355X-native fixedL hot path (libtnml.so).
 *
 * The reference (emstoudenmire/TNML) has no plugin/FFI surface: fixedL.cc is one translation unit
 * whose functions take ITensor objects.  This ABI is the seam cut exactly where
 * mldmrg() (fixedL.cc:451-570) calls into TrainStates / cgrad / quadcost / ITensor svd, so a host
 * driver that keeps the reference's CLI and sweep loop binds these entry points instead.  Each
 * function cites the reference interface it replaces.
 *
 * Conventions
 *  - every call returns 0 on success, non-zero on error; tnml_last_error(ctx) gives the message
 *    (the reference's Error()/EXIT become status codes -- nothing throws across the ABI);
 *  - tensors cross the ABI as raw fp64 (the reference's Real) column-major arrays in ITensor index
 *    order, first index fastest:
 *        site tensor  A_j [ml][2][mr]     (+[10] last, only on the label site c0 = N/2)
 *        bond tensor  B   [mL][2][2][mR]  (+[10] last, only when c0 is b or b+1)
 *        environment  E_j [m] or [m][10]  per image
 *    sites/bonds are 1-indexed as in the reference;
 *  - one context per GPU / per rank; calls on a context are serialised by the caller; the
 *    collective calls (tnml_gradient, tnml_quadcost, tnml_cgrad, tnml_bond_update) must be entered
 *    by every rank of the communicator;
 *  - the context owns all device memory (features, labels, environments, W replica, workspaces),
 *    its HIP stream, rocBLAS/rocSOLVER handles and the RCCL communicator.  There is no CPU
 *    fallback: without a usable HIP device tnml_create fails.
 */
#ifndef TNML_H
#define TNML_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNML_NL 10           /* label dimension, fixedL.cc:15 */
#define TNML_MAX_PASS 64

typedef struct tnml_ctx tnml_ctx;

/* arithmetic and storage type of the per-image path:
   TNML_F64      everything in fp64, as the reference: features, environments, v_mfma_f64_16x16x4_f64
                 contractions, CG and SVD algebra.  The default and the parity mode.
   TNML_F64_E32  fp64 MFMA contractions and fp64 CG/SVD algebra over environments and features STORED in fp32:
                 half the HBM footprint and env traffic, ~9 % faster at maxm=120.  A single evaluation agrees with
                 the reference to ~1e-6, but the rounding of the environments (1e-7) is amplified by the CG, so a
                 sweep follows the fp64 trajectory only loosely (DESIGN.md "fp32 environments").
   TNML_F32      v_mfma_f32_16x16x4_f32, exact-fp32 arithmetic -- 2x the MFMA rate, for the tolerance study only:
                 the reference's CG is not reproducible in fp32 (DESIGN.md "why fp64 MFMA").
   TNML_BF16     as TNML_F32 (fp32 storage, fp32 CG / label dot / shifts) with the two image-proportional GEMMs of the bond
                 contraction -- the feature GEMM of B*t.v and the gradient GEMM dP*dag(t.v) -- on v_mfma_f32_16x16x32_bf16:
                 operands rounded to bf16 while staging, fp32 accumulation (BASELINE config 5's "bf16 MFMA bond
                 contraction"; report-don't-gate).  Option bf16_grad = 0 keeps the gradient GEMM on the fp32 kernel.
   TNML_BF16X3   the same with every operand split hi + lo (x = bf16(x) + bf16(x - bf16(x))) and three bf16 MFMAs per product
                 (hi*hi + hi*lo + lo*hi): ~16 mantissa bits */
enum { TNML_F32 = 0, TNML_F64_E32 = 1, TNML_F64 = 2, TNML_BF16 = 3, TNML_BF16X3 = 4 };
/* eigensolver of the Gram matrix inside tnml_svd_split (n = smaller side of the matricised bond tensor):
   TNML_SVD_SYEVD      in-house: one-workgroup tridiagonalisation, bisection + inverse iteration, back
                       transform, Newton-Schulz polish; verified per call, falls back to rocSOLVER dstedc
                       for the tridiagonal stage when the check fails (n <= 240; larger matrices use
                       TNML_SVD_ROCSOLVER automatically)
   TNML_SVD_ROCSOLVER  stock rocSOLVER dsyevd
   TNML_SVD_STEDC      in-house tridiagonalisation + rocSOLVER dstedc */
enum { TNML_SVD_SYEVD = 0, TNML_SVD_ROCSOLVER = 1, TNML_SVD_STEDC = 2 };
/* which sweep of the reference the context serves:
   TNML_MODE_FIXEDL  fixedL.cc: one weight MPS with a 10-valued Label index on site N/2, targets delta_{l,l_n}
   TNML_MODE_SINGLE  single.cc / single.h: a plain weight MPS (no Label index, tnml_set_site has_label = 0 on every
                     site), scalar output f(x_n) regressed on y_n = [l_n == target_label] (single.h:103,193).  The
                     same kernels run with a label extent of 1; tnml_forward / tnml_classify return one value per
                     image, label_cost[] buckets the cost by the true label of the image, ncorrect counts
                     (f > 1/2) == y (the reference prints no accuracy in this variant). */
enum { TNML_MODE_FIXEDL = 0, TNML_MODE_SINGLE = 1 };

typedef struct {
    int device;          /* HIP device ordinal */
    int rank, nranks;    /* data-parallel rank / world size (images are sharded by rank) */
    int N;               /* number of sites (fixedL.cc:615) */
    int NT_local;        /* training images owned by this rank */
    int64_t NT_total;    /* training images over all ranks (costs are reported un-normalised) */
    int maxm;            /* largest bond dimension that will occur (workspace sizing) */
    int dtype;           /* TNML_F64 (default choice), TNML_F64_E32, TNML_F32, TNML_BF16 or TNML_BF16X3 */
    int svd_backend;     /* TNML_SVD_* */
    int mode;            /* TNML_MODE_FIXEDL (0, default) or TNML_MODE_SINGLE */
    int target_label;    /* TNML_MODE_SINGLE: the selected label L (single.cc:19); the target is y_n = [l_n == L] */
} tnml_config;

/* mirrors the prints of fixedL.cc:391,429-439 */
typedef struct {
    int npass_done;
    int converged;                         /* 1: |r| < cconv hit after a pass (fixedL.cc:432, single.h:273); 2: TNML_MODE_SINGLE only,
                                              |r| < cconv at entry, "not optimizing", B untouched (single.h:202-206) */
    double cost[TNML_MAX_PASS];            /* un-normalised C printed at :429 (index pass-1) */
    double rnorm[TNML_MAX_PASS];           /* |r| printed at :434/:439 */
    double pAp[TNML_MAX_PASS];
    double alpha[TNML_MAX_PASS];
} tnml_cg_trace;

/* knobs of one bond update: Sweeps(Nsweep,minm,maxm,cutoff) fixedL.cc:749 + Args :751-759 */
typedef struct {
    int maxm, minm;
    double cutoff;
    int npass;
    double lambda;       /* used by cgrad (fixedL.cc:356) */
    double lambda_cost;  /* used by the "After SVD" quadcost (cargs copy, fixedL.cc:467) */
    double cconv;
    int report_costs;    /* also evaluate the cost of the old bond tensor and of the optimised one before the split
                            (single.h:621-622 "Cost = %.10f --> %.10f"): two more forward passes */
} tnml_sweep_params;

/* mirrors the prints of fixedL.cc:490,523-533,341-342 */
typedef struct {
    int bond, half, c;
    int mL, mR, label_on_B;                /* outer link dimensions of the bond tensor; Label on B? */
    int origm, newm;
    double truncerr;
    double norm_newB, diff_B_newB;
    double cost_after_svd;                 /* un-normalised */
    double label_cost[TNML_NL];
    double reg_cost;
    int64_t ncorrect;
    tnml_cg_trace cg;
    double cost_old, cost_cg, reg_cost_cg; /* report_costs: quadcost(oB), quadcost(B) and lambda|B|^2 before the split */
    double norm_oB;                        /* norm of the old bond tensor (single.h:572) */
} tnml_bond_report;

/* ---- lifetime ------------------------------------------------------------------------- */
int tnml_create(tnml_ctx** out, const tnml_config* cfg);
int tnml_destroy(tnml_ctx* ctx);
const char* tnml_last_error(const tnml_ctx* ctx);      /* ctx may be NULL: error of the failed create */
/* last non-fatal notice of ctx ("" if none): e.g. the split clamped maxm to the context's maxm and so truncates harder than asked */
const char* tnml_last_warning(const tnml_ctx* ctx);

/* RCCL communicator over xGMI.  Rank 0 obtains a 128-byte unique id, the host transports it to
   the other ranks (any channel), every rank then calls tnml_comm_init.  Replaces the host-side
   stdx::accumulate over per-thread partials, fixedL.cc:333,339,385,402,421,427. */
int tnml_comm_unique_id(void* id128);
int tnml_comm_init(tnml_ctx* ctx, const void* id128);
/* In-process communicator for n ranks that share ONE device (RCCL refuses two ranks on a GPU): ctxs[r] must have been
   created as rank r of n on the same device; call once, from one thread, before the ranks start.  Afterwards every
   collective entry point must be driven by one host thread per rank (they meet in a host barrier).  Same results as
   RCCL (rank-ordered sums, bit-identical on every rank); a correctness vehicle for one-GPU boxes, not a fast path. */
int tnml_comm_init_local(tnml_ctx** ctxs, int n);
/* One-shot all-reduce for the ranks of ONE process on several devices (one host thread per rank, as the fixedL driver runs them):
   every rank writes its packed [scalars | gradient] buffer straight into its slot of every peer's receive region (peer stores over the
   direct xGMI links, one hop), the streams meet through events, every rank sums its own n slots in rank order -- bit-identical sums
   everywhere (replaces the ring all-reduce for the latency-bound 461 KB payloads of fixedL.cc:385,402,421,427).
   Ranks may also share a device (how it is tested on one GPU).  Same calling rules as tnml_comm_init_local. */
int tnml_comm_init_oneshot(tnml_ctx** ctxs, int n);
/* which collective path this context uses: 0 none (one rank), 1 RCCL, 2 in-process staging buffer, 3 one-shot peer write */
/* One-shot all-reduce ACROSS PROCESSES (one process per GPU -- how `python bench.py --gpus N --allreduce oneshot` and any
   torch.distributed / MPI launch run; replaces stdx::accumulate, fixedL.cc:385,402,421,427, like tnml_comm_init): every rank exports an
   IPC handle of its receive region (tnml_oneshot_export), the caller gathers the nranks handles over its own control plane in rank
   order and hands all of them to every rank (tnml_oneshot_connect).  From then on every collective of the library is ONE kernel per
   rank -- peer stores into the mapped regions, device-side arrival flags, an ordered local sum (the same bits on every rank) -- with no
   host synchronisation between the ranks (tnml_ctx option comm_timeout_s bounds the device-side wait; tnml_synchronize reports a
   time-out).  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 and peer access between the ranks' devices; ranks may share a device (tests). */
#define TNML_ONESHOT_HANDLE_BYTES 64
int tnml_oneshot_export(tnml_ctx* ctx, void* handle64);
int tnml_oneshot_connect(tnml_ctx* ctx, const void* handles /* nranks * TNML_ONESHOT_HANDLE_BYTES, rank order */);
/* memory kind of this rank's receive region: 1 fine-grained, 2 uncached, 0 no cross-process transport.  (tnml_oneshot_export fails
   rather than fall back to coarse-grained memory, which is coherent at kernel boundaries only.)  A collective whose device-side wait
   timed out fills its buffer with NaNs and every later entry point that synchronises with the stream returns an error. */
int tnml_oneshot_mem_kind(tnml_ctx* ctx);
/* bytes of the receive region tnml_oneshot_export will allocate for this configuration ([2 parities][nranks][10 Kmax^2 + 48] doubles + flags);
   tnml_estimate_bytes / tnml_plan_maxm do not know the transport and do not count it: subtract it from the budget when planning maxm */
int64_t tnml_oneshot_region_bytes(const tnml_config* cfg);
/* transport in use: 0 none, 1 RCCL, 2 in-process staging buffer, 3 in-process one-shot, 4 cross-process one-shot */
int tnml_collective_mode(tnml_ctx* ctx);
/* Collective.  Verifies that the communicator really spans cfg.nranks ranks (ncclCommCount) and that every rank holds a
   bit-identical replica of the weight MPS (a 64-bit fingerprint of all site tensors, max/min-reduced over the ranks);
   non-zero + tnml_last_error on a mismatch.  The replicated CG/SVD algebra relies on identical replicas the way the
   reference relies on one shared W (fixedL.cc:451); tnml_bond_update runs the same check on the two site tensors it
   rewrites after every split (TNML_CHECK_REPLICAS=0 disables that).  nranks_in_comm (nullable) receives the communicator size. */
int tnml_replica_check(tnml_ctx* ctx, int* nranks_in_comm);
int64_t tnml_replica_repairs(tnml_ctx* ctx);           /* re-broadcasts so far in check_replicas mode 2 */
/* collectives this rank has entered so far: sum all-reduces (the payload: gradient / A p + scalars) and broadcasts (rank 0's
   eigenvalues in the split) */
int tnml_collective_stats(tnml_ctx* ctx, int64_t* allreduces, int64_t* broadcasts);

/* ---- training set: TState ctor fixedL.cc:28-47 + feature map :637-642 -------------------- */
/* raw bytes [NT_local][N] with the reference's feature map phi = [1, byte/(255*255*4)] */
int tnml_set_data_u8(tnml_ctx* ctx, const uint8_t* pixels, const int32_t* labels);
/* arbitrary d=2 local features phi[NT_local][N][2] (TState::data layout) */
int tnml_set_data_phi(tnml_ctx* ctx, const double* phi, const int32_t* labels);

/* ---- weight MPS replica (W.A(j) / W.Aref(j)) ------------------------------------------- */
int tnml_set_site(tnml_ctx* ctx, int j, int ml, int mr, int has_label, const double* A);
int tnml_site_dims(tnml_ctx* ctx, int j, int* ml, int* mr, int* has_label);
int tnml_get_site(tnml_ctx* ctx, int j, double* A);

/* ---- TrainStates::init / setBond / shiftE  (fixedL.cc:122-157, :159-190, :192-233) ------- */
int tnml_env_init(tnml_ctx* ctx);
int tnml_set_bond(tnml_ctx* ctx, int b);               /* selects env buffers; t.v is never formed */
int tnml_shift_env(tnml_ctx* ctx, int b, int from_left);
int tnml_env_dims(tnml_ctx* ctx, int j, int* m, int* has_label);
int tnml_get_env(tnml_ctx* ctx, int j, double* E);     /* [NT_local][m(*10)], for parity tests */
/* The host tier of the environments (the reference's Nbatch / proj_images spill, fixedL.cc:115-120,153,177-178,216,231, with host
   memory in the place of its disk files).  tnml_set_option(ctx, "env_budget_mb", MB) caps the environment slabs held on the device
   (0, the default: no cap -- everything stays in HBM); an environment that does not fit is copied to host memory, farthest from
   the current bond first, and copied back when setBond / shiftE needs it.  Results do not depend on the budget.  Also taken when
   hipMalloc of a new slab fails.  tnml_env_stats: copies to the host / back so far, slabs on the device, bytes currently on the host. */
int tnml_env_stats(tnml_ctx* ctx, int64_t* spills, int64_t* fetches, int64_t* slabs_on_device, int64_t* host_bytes);

/* ---- bond tensor oB = W.A(b)*W.A(b+1)  (fixedL.cc:494,527,745) --------------------------- */
int tnml_bond_dims(tnml_ctx* ctx, int b, int* mL, int* mR, int* label_on_B);
int tnml_bond_tensor(tnml_ctx* ctx, int b, double* B);

/* ---- per-image contractions of the current bond ------------------------------------------ */
/* P_n = B*t.v (fixedL.cc:318,377,399,416): P[NT_local][10] */
int tnml_forward(tnml_ctx* ctx, const double* B, double* P);
/* sum_n dP_n*dag(t.v) over ALL ranks (fixedL.cc:375-385): G has the layout of B */
int tnml_gradient(tnml_ctx* ctx, const double* B, double* G);
/* quadcost (fixedL.cc:280-344): *cost = sum_l C_l + lambda|B|^2, un-normalised, over all ranks */
/* <p|A|p> = sum_n |p*t.v_n|^2 + lambda |p|^2 of a direction p (the pAp pass of cgrad, fixedL.cc:394-403); collective */
int tnml_pAp(tnml_ctx* ctx, const double* p, double lambda, double* pAp);
int tnml_quadcost(tnml_ctx* ctx, const double* B, double lambda, double* cost,
                  double label_cost[TNML_NL], double* reg_cost, int64_t* ncorrect);
/* cgrad (fixedL.cc:349-445): B is updated in place */
int tnml_cgrad(tnml_ctx* ctx, double* B, int npass, double lambda, double cconv, tnml_cg_trace* trace);

/* exact (single.h:117-160; TNML_MODE_SINGLE, TNML_F64, one rank): B = y Phi^+ through the SVD of the D x NT design matrix of the
   v_n (D = 4 mL mR <= 4096; its rows come from D forward passes of unit tensors, the SVD is a one-sided Jacobi on the host so
   that small singular values stay accurate), with the reference's filtered inverse s/(s^2 + lambda) for s > pcut.
   B (ITensor layout, D doubles) is output only.  Meant for small problems, like the reference's (single.h:114). */
int tnml_exact(tnml_ctx* ctx, double* B, double lambda, double pcut);
/* pinv of the per-label variant (single.h:404-517; TNML_MODE_SINGLE, one rank, after tnml_set_bond): a subspace iteration on sum_n v_n v_n^T
   from the start V0 (D x r column-major, D = 4 mL mR in ITensor order, r = the reference's Ntarget <= 64), V <- polar factor of E = V^T A
   until V*E moves by less than 1E-4 or npass passes, then B = yUS * F pseudoInv(D) G.  In the reference the start is random and time-seeded and
   the result only has its cost printed (single.h:596-601; the update that follows is cgrad): a diagnostic.  ve[0..npass]: the V*E trace,
   Dsv[r]: the singular values of the last E. */
int tnml_pinv(tnml_ctx* ctx, const double* V0, int r, int npass, double lambda, double pcut, double* B, double* ve, int* npass_done, double* Dsv);

/* ---- svd(B, W.Aref(c), S, W.Aref(c+dc)); W.Aref(c+dc) *= S  (fixedL.cc:519-521) ---------- */
/* ha = 1: c = b (sweeping right), ha = 2: c = b+1 (sweeping left).  Updates the W replica.
   sv (nullable, capacity >= min(rows,cols)) receives all singular values, descending. */
int tnml_svd_split(tnml_ctx* ctx, const double* B, int b, int ha, double cutoff, int maxm, int minm,
                   double* truncerr, int* newm, double* sv, int* nsv);

/* ---- one iteration of the mldmrg loop body, device resident (fixedL.cc:478-540) ----------- */
/* setBond -> oB -> cgrad -> svd -> newB -> quadcost -> shiftE without tensors leaving the GPU */
int tnml_bond_update(tnml_ctx* ctx, int b, int ha, const tnml_sweep_params* p, tnml_bond_report* rep);
/* The same bond update in two halves, for a sweep loop that keeps the GPU queue full across bond boundaries (mldmrg's loop
   body ends with prints only, fixedL.cc:523-561): _begin enqueues the whole update -- it blocks once, inside the split, for
   the eigenvalues that decide the new bond dimension -- and returns; _end waits for the end-of-bond scalars (cost, #correct,
   norms; multi-rank: the replica fingerprint) and fills the report.  bond k+1 may be begun before bond k is ended (at most
   two in flight; reports come back in order), in which case _end does not wait at all.  tnml_bond_update = _begin + _end. */
int tnml_bond_update_begin(tnml_ctx* ctx, int b, int ha, const tnml_sweep_params* p);
int tnml_bond_update_end(tnml_ctx* ctx, tnml_bond_report* rep);

/* ---- host-side rules (no GPU needed) ------------------------------------------------------ */
/* ---- inference (fulltest.cc:7-100; util.h:19-40 toverlap, util.h:123-200 fullTest) ------------------
 * Full contraction of every local image with the weight MPS held by the context:
 *   weights[n][l] = W_l(image n)  ([NT_local][10], may be NULL),
 *   pred[n]       = argmax_l |W_l|, first maximum on ties (util.h:42-57,160-163; may be NULL),
 *   count[l] / nincorrect[l] = images of label l / of those, wrongly predicted (local images; a multi-rank
 *   caller sums them -- no collective is entered).  The images are the ones given to tnml_set_data_*;
 *   a test set gets its own context.  Training environments held by the context are not modified. */
int tnml_classify(tnml_ctx* ctx, double* weights, int32_t* pred, int64_t count[TNML_NL], int64_t nincorrect[TNML_NL]);

/* ITensor truncate(): p = sigma^2 descending; returns kept m */
int tnml_truncate(const double* p, int n, int maxm, int minm, double cutoff, double* truncerr);
/* ITensor sweepnext (fixedL.cc:478) */
void tnml_sweepnext(int* b, int* ha, int N);
/* contiguous image shard of `rank`: ParallelDo's chunking with GPUs as threads (paralleldo.h:32-43) */
void tnml_shard_bounds(int64_t NT_total, int nranks, int rank, int64_t* begin, int64_t* end);

/* ---- measurement -------------------------------------------------------------------------- */
/* per-kernel-class HIP-event timing on the context's stream (bench.py roofline figures) */
int tnml_profile_enable(tnml_ctx* ctx, int on);
/* restrict the timing to one kernel class (e.g. "fgemm_fwd") or a comma-separated list of classes, or NULL / "" for all classes: two event
   records per timed launch cost host time, so a throughput measurement should time only what it reports */
int tnml_profile_select(tnml_ctx* ctx, const char* class_name);
int tnml_profile_count(tnml_ctx* ctx);
int tnml_profile_get(tnml_ctx* ctx, int idx, char* name64, int64_t* launches, double* total_ms);
int tnml_profile_reset(tnml_ctx* ctx);
int tnml_synchronize(tnml_ctx* ctx);
/* run-time switches of the algebraic shortcuts and checks (defaults: all 1; the environment variables TNML_FAST_CG,
   TNML_REUSE_P, TNML_FUSE_Z, TNML_CHECK_REPLICAS set the defaults at tnml_create):
     "fast_cg"        P <- P + a (p*t.v) instead of re-running the forward GEMM inside cgrad (single.h:290-398 idea)
     "reuse_p"        the after-SVD quadcost of one bond update provides the first residuals of the next
     "fuse_z"         the gradient GEMM builds Z = sum_l EL[l] dP[l] itself
     "merged_cg"      one all-reduce per CG pass instead of two: the image sum of a pass is A p = sum_n (p.v_n) v_n, formed from the
                      pAp pass's own outputs, and travels with sum |p.v_n|^2; the residual follows r <- r - a (A p + lambda p)
                      (the structure of the reference's own fast_cgrad, single.h:347-379); needs fast_cg.  1 (default): on ranks that
                      have a communicator -- where it halves the collectives; a single rank keeps the reference's literal residual
                      (the recurrence moves the 4th step size of an ill-conditioned Label-on-B bond by 1e-3, the cost by 1e-10);
                      2: always; 0: never
     "defer_tail"     multi-rank: the cost partials of the "after SVD" quadcost and the replica fingerprint ride in the first
                      all-reduce of the NEXT bond update (tnml_bond_update_end issues one small all-reduce when nothing followed);
                      with both on a bond update enters 5 payload all-reduces instead of 9 + 1 (default 1)
     "check_replicas" multi-rank: fingerprint check of the two site tensors a split rewrote (1, default: folded into the packed
                      sum all-reduce as exact integer pieces, a mismatch is an error when the report is handed out; 2: checked at
                      once inside the bond update, before anything consumes the tensors -- one extra 8-double all-reduce and a
                      host synchronisation -- and on a mismatch rank 0's two tensors are re-broadcast, the sweep continues,
                      tnml_replica_repairs counts; 0: off)
     "fg64_cfg", "ldot_cfg"  force a tile configuration of the feature GEMM / the label dot that is otherwise chosen by the
                      image count (2 / 1 = the large-image-count forms bench.py times; parity tests run them at small sizes)
     "fused_fwd"      the forward pass of a Label-on-environment bond at m = 120 as one persistent kernel (feature GEMM +
                      label dot of the previous tile, kernels_fused.hip): 1 = from 14 336 images per rank on (default),
                      0 = never, 2 = always (parity tests at small sizes), > 2 = always with that many workgroups at most
     "cg_method"      TNML_MODE_SINGLE only: 0 = conj (single.h:162-288, default), 1 = fast_conj (single.h:290-398: one image sum
                      per CG step, residual by recurrence, the reference's regulariser term as written, no cost in the trace),
                      2 = exact (single.h:117-160, see tnml_exact; "pcut" through tnml_set_option_real)
     "sytrd_exit"     the split's tridiagonalisation stops once the trailing block of the Gram matrix is numerically zero
                      (trace <= 1e-15 trace(G); default 1; 0 = all n-2 Householder steps)
   fast_cg = reuse_p = 0 is the reference's literal evaluation order (fixedL.cc:374-421). */
int tnml_set_option(tnml_ctx* ctx, const char* name, int value);
/* real-valued options: "pcut" (PCut of the exact solver inside tnml_bond_update, single.cc:50, default 1E-8);
   "noise" (TNML_MODE_SINGLE, fp64 storage: the noise of the sweeps, single.cc:25,222 -- from 1E-14 on tnml_svd_split / tnml_bond_update split
   through the density matrix of site c plus noise * sum_n dr_n dr_n^dag, single.h:648-672: W_c = UU, W_{c+dc} = UU * B) */
int tnml_set_option_real(tnml_ctx* ctx, const char* name, double value);
/* The speculative split (minm >= the columns the split may keep: no host synchronisation inside tnml_bond_update_begin): how many splits
   ran speculatively, how many were ROLLED BACK by tnml_bond_update_end because their deferred orthogonality check failed (each repeats
   that bond update with the synchronous split, and the one begun after it), and the device time of the repeated work in ms (event-timed;
   call after tnml_synchronize for the full sum).  fixedL.cc:519-521 has no counterpart: ITensor's svd is synchronous. */
int tnml_split_stats(tnml_ctx* ctx, int64_t* spec_splits, int64_t* roll_backs, double* roll_back_ms);
/* health of the in-house eigensolver: number of fallbacks to rocSOLVER so far, number of splits whose kept basis
   held an eigenvalue cluster and was re-orthonormalised by Cholesky QR, and max|Q^T Q - I| of the kept basis
   before the first / second Newton-Schulz polish step of the last split */
int tnml_svd_stats(tnml_ctx* ctx, int64_t* fallbacks, int64_t* cluster_repairs, double* dev_before_polish, double* dev_after_first_polish);
int64_t tnml_device_bytes(tnml_ctx* ctx);              /* device memory currently owned by ctx */

/* ---- workspace planning (host arithmetic + hipMemGetInfo; used by the drivers before tnml_create) ---------------
 * The reference treats `maxm` (fixedL.cc:592, default 5000) as an upper bound only; a context sizes its workspaces and
 * environment slabs by cfg.maxm, so a driver asks for the largest bond dimension that (a) an MPS of N sites can reach
 * and (b) fits the device, and says so, instead of failing in hipMalloc. */
int64_t tnml_estimate_bytes(const tnml_config* cfg);  /* device bytes of a context after a sweep has built every environment; -1 on bad cfg */
int tnml_device_memory(int device, int64_t* free_bytes, int64_t* total_bytes);
/* largest maxm in [floor_m, wanted] reachable by an N-site MPS whose tnml_estimate_bytes fits budget_bytes (<= 0: no memory bound) */
int tnml_plan_maxm(const tnml_config* cfg, int wanted, int floor_m, int64_t budget_bytes);

#ifdef __cplusplus
}
#endif
#endif
