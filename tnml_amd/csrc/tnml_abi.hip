// tnml_abi.hip -- C-ABI entry points (include/tnml.h) and the device-resident orchestration of
// one bond update of the reference's mldmrg loop (fixedL.cc:478-540).
//
// No CPU fallback lives here: every contraction is a HIP kernel launch on the context's stream.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "tnml_internal.h"

static std::string g_create_err;

int tnml_fail(tnml_ctx* c, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) { c->err = buf; if (c->local && c->coll_depth > 0) local_comm_abort(c); } else g_create_err = buf;   // (inside a collective entry point the peers of an in-process communicator must not wait for a rank that has failed; an error of a local query leaves the communicator alone)
    return 1;
}
const char* tnml_last_error(const tnml_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }
const char* tnml_last_warning(const tnml_ctx* c) { return c ? c->warn.c_str() : ""; }
// Entry points every rank calls in step (they contain all-reduces) hold one of these: a failure inside aborts an in-process communicator.
struct CollScope { tnml_ctx* c; explicit CollScope(tnml_ctx* c_) : c(c_) { ++c->coll_depth; } ~CollScope() { --c->coll_depth; } };

// ---- profiling ------------------------------------------------------------------------------
static hipEvent_t prof_event(tnml_ctx* c) {
    if (!c->prof_free.empty()) { hipEvent_t e = c->prof_free.back(); c->prof_free.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void prof_begin(tnml_ctx* c, int, hipEvent_t* e0, hipStream_t st) { *e0 = prof_event(c); (void)hipEventRecord(*e0, st ? st : c->stream); }
void prof_end(tnml_ctx* c, int kc, hipEvent_t e0, hipStream_t st) {
    hipEvent_t e1 = prof_event(c); (void)hipEventRecord(e1, st ? st : c->stream);
    c->prof_pending.push_back({e0, e1, kc});
    if (c->prof_pending.size() > 8192) prof_resolve(c);
}
void prof_resolve(tnml_ctx* c) {
    if (c->prof_pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) { c->prof_ms[p.kc] += ms; c->prof_launches[p.kc] += 1; }
        c->prof_free.push_back(p.e0); c->prof_free.push_back(p.e1);
    }
    c->prof_pending.clear();
}
int tnml_profile_enable(tnml_ctx* c, int on) { prof_resolve(c); c->prof = on != 0; return 0; }
int tnml_profile_select(tnml_ctx* c, const char* class_name) {
    prof_resolve(c);
    if (!class_name || !*class_name) { c->prof_mask = 0xffffffffu; return 0; }
    unsigned mask = 0;                                    // one class name, or several separated by commas
    const char* s = class_name;
    while (*s) {
        const char* e = strchr(s, ',');
        const size_t len = e ? (size_t)(e - s) : strlen(s);
        bool found = false;
        for (int i = 0; i < KC_COUNT; ++i) if (strlen(kclass_names[i]) == len && !strncmp(s, kclass_names[i], len)) { mask |= 1u << i; found = true; }
        if (!found) return tnml_fail(c, "tnml_profile_select: unknown kernel class in %s", class_name);
        s = e ? e + 1 : s + len;
    }
    c->prof_mask = mask;
    return 0;
}
int tnml_profile_count(tnml_ctx*) { return KC_COUNT; }
int tnml_profile_get(tnml_ctx* c, int idx, char* name64, int64_t* launches, double* total_ms) {
    if (idx < 0 || idx >= KC_COUNT) return tnml_fail(c, "profile index out of range");
    prof_resolve(c);
    if (name64) { strncpy(name64, kclass_names[idx], 63); name64[63] = 0; }
    if (launches) *launches = c->prof_launches[idx];
    if (total_ms) *total_ms = c->prof_ms[idx];
    return 0;
}
int tnml_profile_reset(tnml_ctx* c) {
    prof_resolve(c);
    for (int i = 0; i < KC_COUNT; ++i) { c->prof_launches[i] = 0; c->prof_ms[i] = 0.; }
    return 0;
}
int tnml_set_option(tnml_ctx* c, const char* name, int value) {
    if (!c || !name) return tnml_fail(c, "tnml_set_option: null argument");
    if (!strcmp(name, "fast_cg")) c->fast_cg = value != 0;
    else if (!strcmp(name, "reuse_p")) { c->reuse_p = value != 0; c->p_valid = false; }
    else if (!strcmp(name, "fuse_z")) c->fuse_z = value != 0;
    else if (!strcmp(name, "merged_cg")) c->merged_cg = value;
    else if (!strcmp(name, "defer_tail")) { if (c->pend_count) return tnml_fail(c, "defer_tail: a bond update is in flight"); c->defer_tail = value != 0; }
    else if (!strcmp(name, "check_replicas")) { c->check_replicas = value != 0; c->check_replicas_mode = value; }
    else if (!strcmp(name, "fused_fwd")) c->fused_fwd = value;
    else if (!strcmp(name, "fold_reduce")) c->fold_reduce = value != 0;
    else if (!strcmp(name, "fwd_res")) c->fwd_res = value;
    else if (!strcmp(name, "shift_res")) c->shift_res = value;
    else if (!strcmp(name, "res_grid")) c->res_grid = value;
    else if (!strcmp(name, "res_pace")) c->res_pace = value;
    else if (!strcmp(name, "bgemm_wgs")) c->bgemm_wgs = value;
    else if (!strcmp(name, "bgemm_per")) c->bgemm_per = value;
    else if (!strcmp(name, "sytrd_exit")) c->sytrd_exit = value;
    else if (!strcmp(name, "bgs_chol")) c->bgs_chol = value != 0;
    else if (!strcmp(name, "small_gemm")) c->small_gemm = value != 0;
    else if (!strcmp(name, "spec_split")) c->spec_split = value != 0;
    else if (!strcmp(name, "debug_fail_split")) { c->debug_fail_split = value; c->spec_splits = 0; }
    else if (!strcmp(name, "bf16_grad")) c->bf16_grad = value != 0;
    else if (!strcmp(name, "bf16_once")) c->bf16_once = value != 0;
    else if (!strcmp(name, "env_async")) c->env_async = value != 0;
    else if (!strcmp(name, "env_budget_mb")) { if (value < 0) return tnml_fail(c, "env_budget_mb must be >= 0"); c->env_budget_bytes = (long)value << 20; }
    else if (!strcmp(name, "comm_timeout_s")) { if (value < 1) return tnml_fail(c, "comm_timeout_s must be >= 1"); c->comm_timeout_s = value; local_comm_set_timeout(c, value); }
    else if (!strcmp(name, "cg_method")) { if (value < 0 || value > 2 || (value >= 1 && !c->single())) return tnml_fail(c, "cg_method: 0 (conj) or, in TNML_MODE_SINGLE, 1 (fast_conj) / 2 (exact)"); c->cg_method = value; }
    else if (!strcmp(name, "debug_nudge_rank")) c->debug_nudge_rank = value;
    else if (!strcmp(name, "mc_spin_max")) c->mc_spin_max = value;
    else if (!strcmp(name, "svd_print")) { c->svd_print = value; c->svd_calls = 0; }
    else if (!strcmp(name, "grad_quad")) c->grad_quad = value;
    else if (!strcmp(name, "grad_pair")) c->grad_pair = value;
    else if (!strcmp(name, "grad_pair_min")) c->grad_pair_min = value;
    else if (!strcmp(name, "grad_pair_max")) c->grad_pair_max = value;
    else if (!strcmp(name, "fg64_cfg")) c->opt_fg64_cfg = value;
    else if (!strcmp(name, "ldot_cfg")) c->opt_ldot_cfg = value;
    else return tnml_fail(c, "tnml_set_option: unknown option %s", name);
    return 0;
}
int tnml_synchronize(tnml_ctx* c) { HIPCK(c, hipStreamSynchronize(c->stream)); if (c->copy_stream) HIPCK(c, hipStreamSynchronize(c->copy_stream)); return ipc_comm_check(c); }
int64_t tnml_device_bytes(tnml_ctx* c) { return c->bytes; }
int64_t tnml_replica_repairs(tnml_ctx* c) { return c->replica_repairs; }
int tnml_split_stats(tnml_ctx* c, int64_t* spec_splits, int64_t* roll_backs, double* roll_back_ms) {
    // resolves the event pairs of the roll-backs that have finished (call after tnml_synchronize for the full sum)
    for (size_t k = 0; k < c->redo_events.size();) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->redo_events[k].first, c->redo_events[k].second) == hipSuccess) {
            c->redo_ms += ms;
            (void)hipEventDestroy(c->redo_events[k].first); (void)hipEventDestroy(c->redo_events[k].second);
            c->redo_events.erase(c->redo_events.begin() + k);
        } else { (void)hipGetLastError(); ++k; }
    }
    if (spec_splits) *spec_splits = c->spec_splits_total;
    if (roll_backs) *roll_backs = c->spec_redos;
    if (roll_back_ms) *roll_back_ms = c->redo_ms;
    return 0;
}
int tnml_svd_stats(tnml_ctx* c, int64_t* fallbacks, int64_t* cluster_repairs, double* d0, double* d1) {
    if (fallbacks) *fallbacks = c->svd_fallbacks;
    if (cluster_repairs) *cluster_repairs = c->svd_cholqr;
    if (d0) *d0 = c->svd_last_dev0;
    if (d1) *d1 = c->svd_last_dev1;
    return 0;
}

// ---- host-side rules ------------------------------------------------------------------------
// ITensor v2 truncate() as recalled in SURVEY.md 8(a9): always cut to maxm; then with
// scale = sum(p) (DoRelCutoff) discard while (discarded + p_n) < cutoff*scale and kept > minm.
int tnml_truncate(const double* P, int origm, int maxm, int minm, double cutoff, double* truncerr) {
    if (origm <= 1) { if (truncerr) *truncerr = 0.; return origm; }
    int n = origm - 1;
    double te = 0.;
    while (n >= maxm) { te += P[n]; --n; }
    double scale = 0.;
    for (int j = 0; j < origm; ++j) scale += P[j];
    if (scale == 0.) scale = 1.;
    while (n >= 0 && te + P[n] < cutoff * scale && n >= minm) { te += P[n]; --n; }
    if (n < 0) n = 0;
    if (truncerr) *truncerr = te / scale;
    return n + 1;
}
// ITensor sweepnext: b = 1..N-1 (ha=1) then N-1..1 (ha=2); ha==3 ends the sweep
void tnml_sweepnext(int* b, int* ha, int N) {
    const int inc = (*ha == 1) ? +1 : -1;
    *b += inc;
    if (*b == ((*ha == 1) ? N : 0)) { *b -= inc; ++*ha; }
}
// ParallelDo's static chunking (paralleldo.h:32-43) with ranks in place of threads: equal chunks,
// the last rank takes the remainder
void tnml_shard_bounds(int64_t NT_total, int nranks, int rank, int64_t* begin, int64_t* end) {
    const int64_t th = NT_total / nranks;
    *begin = th * rank;
    *end = (rank == nranks - 1) ? NT_total : th * (rank + 1);
}

// ---- allocation helpers ---------------------------------------------------------------------
template <typename T>
static int dmalloc(tnml_ctx* c, T** p, size_t n) {
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) return tnml_fail(c, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    c->bytes += (int64_t)(n * sizeof(T));
    return 0;
}
int ctx_alloc_doubles(tnml_ctx* c, double** p, size_t n) { return dmalloc(c, p, n); }
static inline int ru16(int x) { return (x + 15) / 16 * 16; }

// Device memory a context of this configuration will own once a sweep has touched every environment: the workspaces of
// tnml_create plus the environment slabs (about N/2 Label-carrying + N/2 Label-free environments
// at any time = 0.55 N slabs of 10*maxm*NTp elements, + the three chain buffers of tnml_classify).
int64_t tnml_estimate_bytes(const tnml_config* cfg) {
    if (!cfg || cfg->N < 1 || cfg->NT_local < 1 || cfg->maxm < 1) return -1;
    const double NTp = (double)((cfg->NT_local + TNML_NTPAD - 1) / TNML_NTPAD * TNML_NTPAD);
    const bool bf = cfg->dtype == TNML_BF16 || cfg->dtype == TNML_BF16X3;
    const double m = cfg->maxm, Kmax = bf ? (2 * cfg->maxm + 31) / 32 * 32 : ru16(2 * cfg->maxm), n = 2. * m;   // (as tnml_create pads it)
    const bool is64 = cfg->dtype == TNML_F64 || cfg->dtype == TNML_F64_E32;
    const double esz = is64 ? 8 : 4, eesz = cfg->dtype == TNML_F64 ? 8 : 4;
    const bool single = cfg->mode == TNML_MODE_SINGLE;
    const double mcap = TNML_NL * Kmax * Kmax;
    double b = cfg->N * 2. * NTp * eesz + NTp * (4 + eesz);                                  // features, labels, ones
    b += (TNML_NL * m * NTp + 3. * TNML_NL * NTp + m * NTp) * esz;                          // U, P, dP, Pp, Zp
    b += mcap * (4 + 6 * 8) + 128. * Kmax * Kmax * 4 * (is64 ? 2 : 1);    // Mf, vB vR vP [tail|G] tB tB2, split-K slabs
    b += 8. * (std::max(40. * m * m, TNML_NL * Kmax * (double)ru16(cfg->maxm)) + 3. * n * n + 7. * n * m + 2. * TNML_NL * m * m + 2. * m * m);   // split workspaces
    b += 8. * (cfg->N - 1 + TNML_NL) * 2. * m * m;                                            // W replica
    b += 8. * (4 + 2 * TNML_NL) * 2. * m * m;                                                 // spare site tensors of the speculative split
    if (2 * cfg->maxm > 240) b += (double)eigh_mc_xbuf_bytes();                               // exchange buffer of the multi-workgroup tridiagonalisation
    if (single) b += 8. * (5. * m * NTp + 3. * NTp + 3. * m * m);                             // workspace of the noise split (allocated on first use with noise > 0)
    const double nslab = single ? (cfg->N / 10. + 2.) : (0.55 * cfg->N + 3.);
    b += nslab * TNML_NL * m * NTp * eesz;
    return (int64_t)b;
}
int tnml_device_memory(int device, int64_t* free_bytes, int64_t* total_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return tnml_fail(nullptr, "tnml_device_memory: no HIP device %d", device);
    int cur = 0; (void)hipGetDevice(&cur);
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return tnml_fail(nullptr, "tnml_device_memory: hipMemGetInfo failed");
    (void)hipSetDevice(cur);
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
}
// Largest bond dimension <= wanted (and >= floor_m) whose context fits into budget_bytes; also bounded by what an MPS of
// N sites can reach at all: min over the two sides of a bond of the full dimension, 2^j and 10*2^(N-j).
int tnml_plan_maxm(const tnml_config* cfg, int wanted, int floor_m, int64_t budget_bytes) {
    if (!cfg || wanted < 1) return -1;
    long reach = 1;
    for (int j = 1; j < cfg->N; ++j) {
        const int l = j, r = cfg->N - j;
        const double dl = l >= 40 ? 1e12 : (double)(1L << l) * (cfg->mode == TNML_MODE_SINGLE ? 1 : TNML_NL);   // the Label index may sit on either side
        const double dr = r >= 40 ? 1e12 : (double)(1L << r) * (cfg->mode == TNML_MODE_SINGLE ? 1 : TNML_NL);
        const double d = dl < dr ? dl : dr;
        if (d > reach) reach = d > 1e9 ? 1000000000L : (long)d;
    }
    int hi = wanted < reach ? wanted : (int)reach;
    if (hi < floor_m) hi = floor_m;
    tnml_config t = *cfg;
    t.maxm = hi;
    if (budget_bytes <= 0 || tnml_estimate_bytes(&t) <= budget_bytes) return hi;
    int lo = floor_m < 1 ? 1 : floor_m;
    t.maxm = lo;
    if (tnml_estimate_bytes(&t) > budget_bytes) return lo;      // even the floor does not fit: let tnml_create report it
    while (hi - lo > 1) { const int mid = lo + (hi - lo) / 2; t.maxm = mid; if (tnml_estimate_bytes(&t) <= budget_bytes) lo = mid; else hi = mid; }
    return lo;
}

int tnml_create(tnml_ctx** out, const tnml_config* cfg) {
    if (!out || !cfg) return tnml_fail(nullptr, "tnml_create: null argument");
    *out = nullptr;
    if (cfg->N < 4) return tnml_fail(nullptr, "tnml_create: need N >= 4 sites");
    if (cfg->NT_local < 1 || cfg->maxm < 1) return tnml_fail(nullptr, "tnml_create: NT_local and maxm must be positive");
    if (cfg->dtype < TNML_F32 || cfg->dtype > TNML_BF16X3) return tnml_fail(nullptr, "tnml_create: dtype must be TNML_F64, TNML_F64_E32, TNML_F32, TNML_BF16 or TNML_BF16X3");
    if (cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks) return tnml_fail(nullptr, "tnml_create: bad rank/nranks");
    if (cfg->mode != TNML_MODE_FIXEDL && cfg->mode != TNML_MODE_SINGLE) return tnml_fail(nullptr, "tnml_create: mode must be TNML_MODE_FIXEDL or TNML_MODE_SINGLE");
    if (cfg->mode == TNML_MODE_SINGLE && (cfg->target_label < 0 || cfg->target_label >= TNML_NL)) return tnml_fail(nullptr, "tnml_create: target_label must be in 0..9");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return tnml_fail(nullptr, "tnml_create: no HIP device available (the HIP path is the only path; there is no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return tnml_fail(nullptr, "tnml_create: device %d out of range (%d visible)", cfg->device, ndev);
    if (hipSetDevice(cfg->device) != hipSuccess) return tnml_fail(nullptr, "tnml_create: hipSetDevice failed");
    tnml_ctx* c = new tnml_ctx();
    c->cfg = *cfg;
    c->N = cfg->N; c->NT = cfg->NT_local; c->maxm = cfg->maxm;
    c->c0 = cfg->mode == TNML_MODE_SINGLE ? -1 : cfg->N / 2;      // fixedL.cc:616; no Label site in the per-label variant
    c->NTp = (cfg->NT_local + TNML_NTPAD - 1) / TNML_NTPAD * TNML_NTPAD;
    int rc = 0;
    auto bail = [&](int r) { g_create_err = c->err; tnml_destroy(c); return r; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(tnml_fail(c, "hipStreamCreate failed"));
    if (const char* e = getenv("TNML_FUSED_FWD")) c->fused_fwd = atoi(e);
    if (const char* e = getenv("TNML_FWD_RES")) c->fwd_res = atoi(e);
    if (const char* e = getenv("TNML_SHIFT_RES")) c->shift_res = atoi(e);
    if (const char* e = getenv("TNML_RES_PACE")) c->res_pace = atoi(e);
    if (const char* e = getenv("TNML_BGEMM_WGS")) c->bgemm_wgs = atoi(e);
    if (const char* e = getenv("TNML_GRAD_QUAD")) c->grad_quad = atoi(e);
    if (const char* e = getenv("TNML_GRAD_PAIR")) c->grad_pair = atoi(e);
    if (const char* e = getenv("TNML_GRAD_PAIR_MIN")) c->grad_pair_min = atoi(e);
    if (const char* e = getenv("TNML_GRAD_PAIR_MAX")) c->grad_pair_max = atoi(e);
    if (const char* e = getenv("TNML_BGEMM_PER")) c->bgemm_per = atoi(e);
    if (const char* e = getenv("TNML_BGS_CHOL")) c->bgs_chol = atoi(e) != 0;
    if (const char* e = getenv("TNML_SMALL_GEMM")) c->small_gemm = atoi(e) != 0;
    if (rocblas_create_handle(&c->blas) != rocblas_status_success) return bail(tnml_fail(c, "rocblas_create_handle failed"));
    rocblas_set_stream(c->blas, c->stream);
    // replicas of W must stay bit-identical over the ranks: no atomics-based split-K inside rocBLAS
    rocblas_set_atomics_mode(c->blas, rocblas_atomics_not_allowed);
    const size_t NTp = c->NTp;
    const int Kmax = c->bf16() ? (2 * c->maxm + 31) / 32 * 32 : ru16(2 * c->maxm);
    c->mcap = (size_t)TNML_NL * Kmax * Kmax;
    c->small_elems = (size_t)c->maxm * NTp;
    c->big_elems = (size_t)TNML_NL * c->maxm * NTp;
    c->svd_n = 2 * c->maxm;
    c->slab_bytes = (size_t)128 * Kmax * Kmax * 4 * ((cfg->dtype == TNML_F64 || cfg->dtype == TNML_F64_E32) ? 2 : 1);
    c->partial_cap = (int)(NTp / 64);
    c->W.resize(c->N + 2);
    c->env.resize(c->N + 2);
    if ((rc = dmalloc(c, (char**)&c->phi, (size_t)c->N * 2 * NTp * c->eesz()))) return bail(rc);
    if ((rc = dmalloc(c, &c->label, NTp))) return bail(rc);
    if ((rc = dmalloc(c, (char**)&c->ones, NTp * c->eesz()))) return bail(rc);
    const size_t esz = c->esz();
    if ((rc = dmalloc(c, (char**)&c->U, c->big_elems * esz))) return bail(rc);
    if ((rc = dmalloc(c, (char**)&c->P, (size_t)TNML_NL * NTp * esz))) return bail(rc);
    if ((rc = dmalloc(c, (char**)&c->dP, (size_t)TNML_NL * NTp * esz))) return bail(rc);
    if ((rc = dmalloc(c, (char**)&c->Pp, (size_t)TNML_NL * NTp * esz))) return bail(rc);
    if (const char* e = getenv("TNML_FAST_CG")) c->fast_cg = atoi(e) != 0;
    if (const char* e = getenv("TNML_FUSE_Z")) c->fuse_z = atoi(e) != 0;
    if (const char* e = getenv("TNML_REUSE_P")) c->reuse_p = atoi(e) != 0;
    if (const char* e = getenv("TNML_MERGED_CG")) c->merged_cg = atoi(e);
    if (const char* e = getenv("TNML_DEFER_TAIL")) c->defer_tail = atoi(e) != 0;
    if (const char* e = getenv("TNML_FG64_CFG")) c->opt_fg64_cfg = atoi(e);
    if (const char* e = getenv("TNML_LDOT_CFG")) c->opt_ldot_cfg = atoi(e);
    if ((rc = dmalloc(c, (char**)&c->Zp, c->small_elems * esz))) return bail(rc);
    if ((rc = dmalloc(c, &c->Mf, c->mcap))) return bail(rc);
    if ((rc = dmalloc(c, (char**)&c->slab, c->slab_bytes))) return bail(rc);
    if ((rc = dmalloc(c, &c->partials, (size_t)c->partial_cap * 12))) return bail(rc);
    if ((rc = dmalloc(c, &c->partials2, (size_t)c->partial_cap * 12))) return bail(rc);
    if ((rc = dmalloc(c, &c->counters, 16))) return bail(rc);
    if (hipMemsetAsync(c->counters, 0, 16 * sizeof(unsigned), c->stream) != hipSuccess) return bail(tnml_fail(c, "memset failed"));
    if (cfg->dtype == TNML_F64 && cfg->mode == TNML_MODE_FIXEDL && c->maxm >= 33 && (rc = dmalloc(c, &c->Ppart, (size_t)2 * TNML_NL * NTp))) return bail(rc);   // k_fwd_res (input dimensions 33..120)
    if (c->bf16()) {                                    // bf16 copies of the forward pass's operands (kernels_bf16e.hip)
        c->ebt_cap = bf16e_env_elems(c->maxm, NTp, c->bf16() == 2); c->mbt_cap = bf16e_m_elems(c->maxm, c->bf16() == 2);
        if ((rc = dmalloc(c, &c->ebt, c->ebt_cap)) || (rc = dmalloc(c, &c->mbt, c->mbt_cap))) return bail(rc);
    }
    if ((rc = dmalloc(c, &c->vB, c->mcap))) return bail(rc);
    if ((rc = dmalloc(c, &c->vR, c->mcap))) return bail(rc);
    if ((rc = dmalloc(c, &c->vP, c->mcap))) return bail(rc);
    if ((rc = dmalloc(c, &c->arbuf, c->mcap + TNML_TAILN))) return bail(rc);
    c->tail = c->arbuf; c->vG = c->arbuf + TNML_TAILN;
    if ((rc = dmalloc(c, &c->locals, 32))) return bail(rc);
    if ((rc = dmalloc(c, &c->scal, SC_N + (size_t)4 * TNML_MAX_PASS))) return bail(rc);   // CG scalars, then the per-pass trace: one copy to the host
    c->cgtrace = c->scal + SC_N;
    if ((rc = dmalloc(c, &c->vpart, 1024 + 16))) return bail(rc);   // [256][2] phase-1 partials, then [256][2] for |p|^2 of the next pass, then the summed cost of an output update
    if ((rc = dmalloc(c, &c->tB, c->mcap))) return bail(rc);
    if ((rc = dmalloc(c, &c->tB2, c->mcap))) return bail(rc);
    // sM holds (a) the Label-permuted bond matrix of the split, 40 maxm^2, and (b) the 16-padded site matrix of an
    // environment shift, L * ru16(2 m) * ru16(m) -- at small maxm the padding of (b) dominates
    c->sM_cap = std::max((size_t)40 * c->maxm * c->maxm, (size_t)TNML_NL * Kmax * ru16(c->maxm));
    if ((rc = dmalloc(c, &c->sM, c->sM_cap))) return bail(rc);
    if ((rc = dmalloc(c, &c->sG, (size_t)c->svd_n * c->svd_n))) return bail(rc);
    if ((rc = dmalloc(c, &c->sD, (size_t)c->svd_n))) return bail(rc);
    if ((rc = dmalloc(c, &c->sE, (size_t)2 * c->svd_n))) return bail(rc);
    if ((rc = dmalloc(c, &c->sF, (size_t)c->svd_n * c->maxm + (size_t)2 * TNML_NL * c->maxm * c->maxm))) return bail(rc);
    if ((rc = dmalloc(c, &c->sInfo, 4))) return bail(rc);
    if ((rc = dmalloc(c, &c->fprint, 2))) return bail(rc);
    if (const char* e = getenv("TNML_CHECK_REPLICAS")) { c->check_replicas = atoi(e) != 0; c->check_replicas_mode = atoi(e); }
    if ((rc = dmalloc(c, &c->sE2, (size_t)c->svd_n))) return bail(rc);
    if ((rc = dmalloc(c, &c->sTau, (size_t)c->svd_n))) return bail(rc);
    if ((rc = dmalloc(c, &c->sV, (size_t)c->svd_n * c->svd_n))) return bail(rc);
    if ((rc = dmalloc(c, &c->sC, (size_t)c->svd_n * c->svd_n))) return bail(rc);
    if ((rc = dmalloc(c, &c->sW, (size_t)c->svd_n + 8))) return bail(rc);           // + room for the orthogonality check values behind the eigenvalues
    if ((rc = dmalloc(c, &c->sScr, std::max<size_t>((size_t)5 * c->svd_n * c->maxm, TEIG_SCRATCH_DOUBLES)))) return bail(rc);
    if ((rc = dmalloc(c, &c->sS, (size_t)c->maxm * c->maxm))) return bail(rc);
    if ((rc = dmalloc(c, &c->sCm, (size_t)c->maxm * c->maxm))) return bail(rc);
    if ((rc = dmalloc(c, &c->sQ1, (size_t)c->svd_n * c->maxm))) return bail(rc);
    if ((rc = dmalloc(c, &c->sDev, 4))) return bail(rc);
    if (c->svd_n > 240) {                               // multi-workgroup tridiagonalisation (eigh_mc.hip)
        if ((rc = dmalloc(c, (char**)&c->mc_xbuf, eigh_mc_xbuf_bytes()))) return bail(rc);
        if (hipMemsetAsync(c->mc_xbuf, 0, eigh_mc_xbuf_bytes(), c->stream) != hipSuccess) return bail(tnml_fail(c, "memset failed"));
    }
    if (const char* e = getenv("TNML_SVD_BACKEND")) c->cfg.svd_backend = atoi(e);
    if (const char* e = getenv("TNML_SVD_PRINT")) c->svd_print = atoi(e);
    if (const char* e = getenv("TNML_SVD_DUMP")) c->svd_dump = e;
    for (int k = 0; k < 2; ++k)
        if (hipEventCreateWithFlags(&c->pend[k].ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->pend[k].ev2, hipEventDisableTiming) != hipSuccess)
            return bail(tnml_fail(c, "hipEventCreate failed"));
    if (hipHostMalloc((void**)&c->h_scal, sizeof(double) * (2 * c->svd_n + 64 + SC_N + 4 * TNML_MAX_PASS + 2 * 64)) != hipSuccess) return bail(tnml_fail(c, "hipHostMalloc failed"));
    for (int j = 1; j <= c->N; ++j) {
        const size_t cap = (size_t)2 * c->maxm * c->maxm * (j == c->c0 ? TNML_NL : 1);
        if ((rc = dmalloc(c, &c->W[j].a, cap))) return bail(rc);
    }
    // speculative split: pinned mirrors [eigenvalues + 4 check values | CG scalars + trace] per bond update in flight, and spare site
    // tensors (two bond updates in flight replace two sites each; the Label site has its own size class)
    c->hrep_stride = (size_t)c->svd_n + 8 + SC_N + (size_t)4 * TNML_MAX_PASS + 512 + 64;      // eigenvalues + checks | scal + trace | norm partial pairs | after-SVD scalars
    if (hipHostMalloc((void**)&c->hrep, sizeof(double) * 2 * c->hrep_stride) != hipSuccess) return bail(tnml_fail(c, "hipHostMalloc failed"));
    if (hipHostMalloc((void**)&c->hcost, sizeof(double) * 2 * (size_t)c->partial_cap * 12) != hipSuccess) return bail(tnml_fail(c, "hipHostMalloc failed"));
    memset(c->hrep, 0, sizeof(double) * 2 * c->hrep_stride);
    for (int k = 0; k < 4 + (c->c0 > 0 ? 2 : 0); ++k) {
        double* sp = nullptr;
        if ((rc = dmalloc(c, &sp, (size_t)2 * c->maxm * c->maxm * (k >= 4 ? TNML_NL : 1)))) return bail(rc);
        (k >= 4 ? c->spare_big : c->spare_small).push_back(sp);
    }
    if (const char* e = getenv("TNML_SPEC_SPLIT")) c->spec_split = atoi(e);
    if (hipMemsetAsync(c->arbuf, 0, sizeof(double) * (c->mcap + TNML_TAILN), c->stream) != hipSuccess) return bail(tnml_fail(c, "memset failed"));
    if (hipMemsetAsync(c->locals, 0, sizeof(double) * 32, c->stream) != hipSuccess) return bail(tnml_fail(c, "memset failed"));
    if (hipMemsetAsync(c->scal, 0, sizeof(double) * SC_N, c->stream) != hipSuccess) return bail(tnml_fail(c, "memset failed"));
    if ((rc = c->env64() ? launch_fill_f64(c, (double*)c->ones, 1.0, NTp) : launch_fill_f32(c, (float*)c->ones, 1.0f, NTp))) return bail(rc);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(tnml_fail(c, "sync failed"));
    *out = c;
    return 0;
}

int tnml_destroy(tnml_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) ncclCommDestroy(c->comm);
    local_comm_release(c);
    ipc_comm_release(c);
    for (int k = 0; k < 2; ++k) { if (c->pend[k].ev) (void)hipEventDestroy(c->pend[k].ev); if (c->pend[k].ev2) (void)hipEventDestroy(c->pend[k].ev2); }
    for (auto& p : c->prof_pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto e : c->prof_free) (void)hipEventDestroy(e);
    for (auto& p : c->redo_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    void* ptrs[] = {c->phi, c->label, c->ones, c->U, c->P, c->dP, c->Pp, c->Zp, c->Mf, c->slab, c->partials, c->partials2, c->vB, c->vR, c->vP,
                    c->arbuf, c->locals, c->scal, c->vpart, c->counters, c->Ppart, c->tB, c->tB2, c->sM, c->sG, c->sD, c->sE, c->sF, c->sInfo, c->sE2, c->sTau, c->sV, c->sC, c->sW, c->sScr, c->sS, c->sCm, c->sQ1, c->sDev, c->mc_xbuf, c->fprint, c->noise_ws, c->ebt, c->mbt};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    // (site tensors and spares have changed places during speculative splits: every buffer is in exactly one of the two sets)
    for (auto& s : c->W) if (s.a) (void)hipFree(s.a);
    for (size_t k = 0; k < c->spare_small.size(); ++k) (void)hipFree(c->spare_small[k]);
    for (size_t k = 0; k < c->spare_big.size(); ++k) (void)hipFree(c->spare_big[k]);
    for (int k = 0; k < 2; ++k) for (int u = 0; u < c->pend[k].nundo; ++u) if (c->pend[k].undo[u].old) (void)hipFree(c->pend[k].undo[u].old);
    if (c->hrep) (void)hipHostFree(c->hrep);
    if (c->hcost) (void)hipHostFree(c->hcost);
    for (auto& sl : c->slabs) if (sl.base) (void)hipFree(sl.base);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->ev_compute) (void)hipEventDestroy(c->ev_compute);
    for (auto& e : c->env) { if (e.host) { if (e.host_pinned) (void)hipHostFree(e.host); else free(e.host); } if (e.ev) (void)hipEventDestroy(e.ev); }
    for (auto& sl : c->slabs) if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->blas) rocblas_destroy_handle(c->blas);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

// ---- RCCL -----------------------------------------------------------------------------------
int tnml_comm_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return tnml_fail(nullptr, "ncclGetUniqueId failed");
    memcpy(id128, &id, sizeof id);
    return 0;
}
int tnml_comm_init(tnml_ctx* c, const void* id128) {
    // a single rank needs no communicator; TNML_FORCE_COMM=1 builds a 1-rank one anyway so that the RCCL path
    // (communicator setup, stream-ordered all-reduce) can be exercised on a one-GPU box
    if (c->cfg.nranks == 1 && !(getenv("TNML_FORCE_COMM") && atoi(getenv("TNML_FORCE_COMM")))) return 0;
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    HIPCK(c, hipSetDevice(c->cfg.device));
    ncclResult_t r = ncclCommInitRank(&c->comm, c->cfg.nranks, id, c->cfg.rank);
    if (r != ncclSuccess) return tnml_fail(c, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    return 0;
}
// sum over ranks of a fp64 device buffer, in stream order (replaces stdx::accumulate, fixedL.cc:385,402,421,427)
static int allreduce(tnml_ctx* c, double* buf, size_t count) {
    if (c->ipc) { ProfScope ps(c, KC_ALLREDUCE); c->allreduce_calls += 1; return ipc_comm_exchange(c, buf, count, 0); }
    if (c->local) { ProfScope ps(c, KC_ALLREDUCE); c->allreduce_calls += 1; return local_comm_exchange(c, buf, count, 0); }
    if (!c->comm) {
        if (c->cfg.nranks == 1) return 0;
        return tnml_fail(c, "nranks > 1 but tnml_comm_init was not called");
    }
    ProfScope ps(c, KC_ALLREDUCE);
    c->allreduce_calls += 1;
    ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) return tnml_fail(c, "ncclAllReduce failed: %s", ncclGetErrorString(r));
    return 0;
}
int allreduce_sum(tnml_ctx* c, double* buf, size_t count) { return allreduce(c, buf, count); }
static double* pend_host(tnml_ctx* c, int slot) { return c->hrep + (size_t)slot * c->hrep_stride + c->svd_n + 8 + SC_N + 4 * TNML_MAX_PASS + 512; }
static double* dn_host(tnml_ctx* c, int slot) { return c->hrep + (size_t)slot * c->hrep_stride + c->svd_n + 8 + SC_N + 4 * TNML_MAX_PASS; }
// the carried slots of a finished bond update (after-SVD cost partials, fingerprint pieces) have just been summed over the ranks by an
// all-reduce that covered them: hand them to the host report they belong to
static int carry_deliver(tnml_ctx* c) {
    if (c->carry_slot < 0) return 0;
    const int slot = c->carry_slot;
    c->carry_slot = -1;
    if (!c->pend[slot].carry_direct)      // (one rank: k_reduce_partials has mirrored the cost partials into the report block itself)
        HIPCK(c, hipMemcpyAsync(pend_host(c, slot) + TNML_CARRY, c->tail + TNML_CARRY, sizeof(double) * TNML_CARRYN, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipEventRecord(c->pend[slot].ev2, c->stream));
    if (c->multi())                                        // delivered: the next packed all-reduce must not sum (and so scale by nranks) what is left here
        HIPCK(c, hipMemsetAsync(c->tail + TNML_CARRY, 0, sizeof(double) * TNML_CARRYN, c->stream));
    return 0;
}
// the packed buffer [tail | G] of the current bond (n = elements of G)
static int allreduce_packed(tnml_ctx* c, size_t n) {
    TCK(allreduce(c, c->arbuf, TNML_TAILN + n));
    return carry_deliver(c);
}
int bcast_rank0(tnml_ctx* c, double* buf, size_t count) {
    if (c->ipc) { c->bcast_calls += 1; return ipc_comm_exchange(c, buf, count, 1); }
    if (c->local) { c->bcast_calls += 1; return local_comm_exchange(c, buf, count, 1); }
    if (!c->comm) return 0;
    c->bcast_calls += 1;
    ncclResult_t r = ncclBroadcast(buf, buf, count, ncclDouble, 0, c->comm, c->stream);
    if (r != ncclSuccess) return tnml_fail(c, "ncclBroadcast failed: %s", ncclGetErrorString(r));
    return 0;
}
int tnml_collective_mode(tnml_ctx* c) { return c->ipc ? 4 : (c->local ? local_comm_mode(c) : (c->comm ? 1 : 0)); }
int tnml_collective_stats(tnml_ctx* c, int64_t* allreduces, int64_t* broadcasts) {
    if (allreduces) *allreduces = c->allreduce_calls;
    if (broadcasts) *broadcasts = c->bcast_calls;
    return 0;
}
static int check_W(tnml_ctx* c);
// fingerprint of the replicated site tensors j0..j1 as exact integer pieces -> out8 (device; see k_fingerprint_pieces)
static int replica_fingerprint(tnml_ctx* c, int j0, int j1, double* out8) {
    for (int j = j0; j <= j1; ++j) {
        const SiteT& s = c->W[j];
        TCK(launch_fingerprint(c, s.a, (size_t)s.ml * 2 * s.mr * s.L, 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * j + 1), c->fprint, j == j0));
    }
    return launch_fingerprint_pieces(c, c->fprint, out8);
}
// sums S_i, Q_i of the fingerprint pieces over R ranks: every rank held the same fingerprint iff R Q_i == S_i^2 for all four pieces
static bool fingerprint_agrees(const double* sums8, int nranks) {
    for (int i = 0; i < 4; ++i) if ((double)nranks * sums8[4 + i] != sums8[i] * sums8[i]) return false;
    return true;
}
int tnml_replica_check(tnml_ctx* c, int* nranks_in_comm) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (nranks_in_comm) *nranks_in_comm = 1;
    if (!c->multi()) return c->cfg.nranks == 1 ? 0 : tnml_fail(c, "tnml_replica_check: nranks > 1 but tnml_comm_init was not called");
    int cnt = 0;
    if (c->local) cnt = local_comm_size(c);
    else if (c->ipc) cnt = c->cfg.nranks;
    else if (ncclCommCount(c->comm, &cnt) != ncclSuccess) return tnml_fail(c, "ncclCommCount failed");
    if (nranks_in_comm) *nranks_in_comm = cnt;
    if (cnt != c->cfg.nranks) return tnml_fail(c, "communicator has %d ranks, context was created for %d", cnt, c->cfg.nranks);
    TCK(check_W(c));
    if (c->pend_count) return tnml_fail(c, "tnml_replica_check: a bond update is in flight");
    TCK(replica_fingerprint(c, 1, c->N, c->tail + TNML_FPSLOT));
    TCK(allreduce(c, c->tail + TNML_FPSLOT, 8));
    double h[8];
    HIPCK(c, hipMemcpyAsync(h, c->tail + TNML_FPSLOT, sizeof h, hipMemcpyDeviceToHost, c->stream));
    SYNCK(c, c->stream);
    if (!fingerprint_agrees(h, c->cfg.nranks)) return tnml_fail(c, "replicas of the weight MPS differ between ranks");
    return 0;
}

// ---- training set -----------------------------------------------------------------------------
static int set_labels(tnml_ctx* c, const int32_t* labels) {
    std::vector<int> lab(c->NTp, -1);
    for (int i = 0; i < c->NT; ++i) {
        if (labels[i] < 0 || labels[i] >= TNML_NL) return tnml_fail(c, "label %d of image %d out of range", labels[i], i);
        lab[i] = labels[i];
    }
    HIPCK(c, hipMemcpy(c->label, lab.data(), sizeof(int) * c->NTp, hipMemcpyHostToDevice));
    return 0;
}
int tnml_set_data_u8(tnml_ctx* c, const uint8_t* pixels, const int32_t* labels) {
    HIPCK(c, hipSetDevice(c->cfg.device));
    TCK(set_labels(c, labels));
    uint8_t* d_pix = nullptr;
    const size_t nb = (size_t)c->NT * c->N;
    HIPCK(c, hipMalloc((void**)&d_pix, nb));
    HIPCK(c, hipMemcpy(d_pix, pixels, nb, hipMemcpyHostToDevice));
    int rc = launch_features_u8(c, d_pix, c->N, c->NT, c->NTp, c->phi);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_pix);
    if (rc) return rc;
    c->data_set = true; c->currb = -1; c->p_valid = false;
    return 0;
}
int tnml_set_data_phi(tnml_ctx* c, const double* phi, const int32_t* labels) {
    HIPCK(c, hipSetDevice(c->cfg.device));
    TCK(set_labels(c, labels));
    // TState::data[(j-1)*d + (n-1)] (fixedL.cc:39-46) -> [N][2][NTp], rounded once to fp32
    const size_t ne = (size_t)c->N * 2 * c->NTp;
    if (c->env64()) {
        std::vector<double> h(ne, 0.);
        for (int i = 0; i < c->NT; ++i) for (int j = 0; j < c->N; ++j) for (int s = 0; s < 2; ++s)
            h[((size_t)j * 2 + s) * c->NTp + i] = phi[((size_t)i * c->N + j) * 2 + s];
        HIPCK(c, hipMemcpy(c->phi, h.data(), sizeof(double) * ne, hipMemcpyHostToDevice));
    } else {
        std::vector<float> h(ne, 0.f);
        for (int i = 0; i < c->NT; ++i) for (int j = 0; j < c->N; ++j) for (int s = 0; s < 2; ++s)
            h[((size_t)j * 2 + s) * c->NTp + i] = (float)phi[((size_t)i * c->N + j) * 2 + s];
        HIPCK(c, hipMemcpy(c->phi, h.data(), sizeof(float) * ne, hipMemcpyHostToDevice));
    }
    c->data_set = true; c->currb = -1; c->p_valid = false;
    return 0;
}

// ---- weight MPS replica -------------------------------------------------------------------------
int tnml_set_site(tnml_ctx* c, int j, int ml, int mr, int has_label, const double* A) {
    if (j < 1 || j > c->N) return tnml_fail(c, "tnml_set_site: site %d out of range", j);
    if (c->single() && has_label) return tnml_fail(c, "tnml_set_site: the per-label variant has no Label index");
    if ((j == c->c0) != (has_label != 0)) return tnml_fail(c, "Label Index not on site %d", c->c0);     // fixedL.cc:734
    if (ml < 1 || mr < 1 || ml > c->maxm || mr > c->maxm) return tnml_fail(c, "tnml_set_site: bond dimension outside 1..maxm");
    if ((j == 1 && ml != 1) || (j == c->N && mr != 1)) return tnml_fail(c, "tnml_set_site: edge sites must have outer dimension 1");
    SiteT& s = c->W[j];
    s.ml = ml; s.mr = mr; s.L = has_label ? TNML_NL : 1; s.set = true;
    HIPCK(c, hipMemcpy(s.a, A, sizeof(double) * (size_t)ml * 2 * mr * s.L, hipMemcpyHostToDevice));
    c->currb = -1; c->p_valid = false;
    return 0;
}
int tnml_site_dims(tnml_ctx* c, int j, int* ml, int* mr, int* has_label) {
    if (j < 1 || j > c->N || !c->W[j].set) return tnml_fail(c, "tnml_site_dims: site %d not set", j);
    *ml = c->W[j].ml; *mr = c->W[j].mr; *has_label = c->W[j].L == TNML_NL;
    return 0;
}
int tnml_get_site(tnml_ctx* c, int j, double* A) {
    if (j < 1 || j > c->N || !c->W[j].set) return tnml_fail(c, "tnml_get_site: site %d not set", j);
    const SiteT& s = c->W[j];
    SYNCK(c, c->stream);
    HIPCK(c, hipMemcpy(A, s.a, sizeof(double) * (size_t)s.ml * 2 * s.mr * s.L, hipMemcpyDeviceToHost));
    return 0;
}
static int check_W(tnml_ctx* c) {
    for (int j = 1; j <= c->N; ++j) {
        if (!c->W[j].set) return tnml_fail(c, "W: site %d not set", j);
        if (j > 1 && c->W[j].ml != c->W[j - 1].mr) return tnml_fail(c, "W: bond dimension mismatch between sites %d and %d", j - 1, j);
    }
    return 0;
}

// ---- environments -------------------------------------------------------------------------------
static void slot_release(tnml_ctx* c, EnvSlot& e) {
    e.on_host = false;                                   // (a spilled copy of an environment that is being rebuilt is stale)
    if (!e.ptr) return;
    if (e.ev_pending) {                                  // a prefetch of the value that is being replaced is still in flight: the unit's next user must not overtake it
        EnvSlab& sl = c->slabs[e.slab];
        if (!sl.ev) (void)hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming);
        (void)hipEventRecord(sl.ev, c->copy_stream);
        sl.ev_pending = true; e.ev_pending = false;
    }
    c->slabs[e.slab].mask &= (e.unit < 0) ? 0u : ~(1u << e.unit);
    e.ptr = nullptr; e.slab = e.unit = -1;
}
// The host tier.  With option env_budget_mb the environment slabs on the device are capped; when a new slab would exceed the cap (or
// hipMalloc fails) a whole slab is evicted: the one whose environments lie farthest from the current bond -- in a sweep those are
// needed last -- and none of which is an operand of the operation in flight.  Copies run on the compute stream (in order with the
// kernels that wrote / will read the data); pinned host buffers when the host grants them, pageable ones otherwise.
static bool env_is_protected(const tnml_ctx* c, int j) { for (int k = 0; k < 4; ++k) if (c->env_protect[k] == j) return true; return false; }
static int env_copy_stream(tnml_ctx* c) {
    if (c->copy_stream) return 0;
    HIPCK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCK(c, hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming));
    return 0;
}
static int env_host_buffer(tnml_ctx* c, EnvSlot& e, size_t bytes, int j) {
    if (e.host_cap >= bytes) return 0;
    if (e.host) { if (e.host_pinned) (void)hipHostFree(e.host); else free(e.host); e.host = nullptr; e.host_cap = 0; }
    void* hp = nullptr;
    if (hipHostMalloc(&hp, bytes, hipHostMallocDefault) == hipSuccess) { e.host = (char*)hp; e.host_pinned = true; }
    else { (void)hipGetLastError(); e.host = (char*)malloc(bytes); e.host_pinned = false; }
    if (!e.host) return tnml_fail(c, "environment spill: no host memory for %zu bytes (site %d)", bytes, j);
    e.host_cap = bytes;
    return 0;
}
// environment j -> host.  Asynchronous form (pinned buffer, option env_async): the copy runs on the copy stream once everything the
// compute stream holds so far has finished, and the slab remembers the event that marks its end; whoever takes a unit of that slab
// next waits for it.  Otherwise: on the compute stream, in order.
static int env_spill(tnml_ctx* c, int j) {
    EnvSlot& e = c->env[j];
    const size_t bytes = (size_t)e.L * e.m * c->NTp * c->eesz();
    TCK(env_host_buffer(c, e, bytes, j));
    EnvSlab& sl = c->slabs[e.slab];
    if (c->env_async && e.host_pinned) {
        TCK(env_copy_stream(c));
        HIPCK(c, hipEventRecord(c->ev_compute, c->stream));
        HIPCK(c, hipStreamWaitEvent(c->copy_stream, c->ev_compute, 0));
        HIPCK(c, hipMemcpyAsync(e.host, e.ptr, bytes, hipMemcpyDeviceToHost, c->copy_stream));
        if (!sl.ev) HIPCK(c, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        HIPCK(c, hipEventRecord(sl.ev, c->copy_stream));
        sl.ev_pending = true;
    } else {
        if (e.ev_pending) { HIPCK(c, hipStreamWaitEvent(c->stream, e.ev, 0)); }
        HIPCK(c, hipMemcpyAsync(e.host, e.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
        if (!e.host_pinned) SYNCK(c, c->stream);
    }
    e.ev_pending = false;                               // (a copy back that was still in flight is ordered before this one: same stream, or waited for above)
    sl.mask &= (e.unit < 0) ? 0u : ~(1u << e.unit);
    e.ptr = nullptr; e.slab = e.unit = -1; e.on_host = true;
    c->env_spills += 1;
    return 0;
}
static int env_evict_slab(tnml_ctx* c) {                // frees one whole slab; 1 = nothing could be evicted
    const int pos = c->currb > 0 ? c->currb : 1;
    int best = -1, best_d = -1;
    for (size_t k = 0; k < c->slabs.size(); ++k) {
        if (!c->slabs[k].mask) continue;
        int dmin = 1 << 30; bool ok = true, any = false;
        for (int j = 1; j <= c->N; ++j) {
            const EnvSlot& e = c->env[j];
            if (!e.ptr || e.slab != (int)k) continue;
            any = true;
            if (env_is_protected(c, j)) { ok = false; break; }
            const int d = j > pos ? j - pos : pos - j;
            if (d < dmin) dmin = d;
        }
        if (!ok || !any) continue;                      // (slabs that hold classify's chain buffers have no environment: never evicted)
        if (dmin > best_d) { best_d = dmin; best = (int)k; }
    }
    if (best < 0) return 1;
    for (int j = 1; j <= c->N; ++j) if (c->env[j].ptr && c->env[j].slab == best) TCK(env_spill(c, j));
    return 0;
}
// consumer: the stream whose work will touch the new unit first (it waits for a copy to the host that may still be reading the slab)
static int slot_acquire(tnml_ctx* c, EnvSlot& e, int m, int L, hipStream_t consumer = nullptr) {
    slot_release(c, e);
    if (!consumer) consumer = c->stream;
    const unsigned FULL = (1u << TNML_NL) - 1;
    const size_t slab_bytes = c->big_elems * c->eesz();
    for (;;) {
        int pick = -1;
        if (L != TNML_NL) for (size_t k = 0; k < c->slabs.size(); ++k) if (c->slabs[k].mask && c->slabs[k].mask != FULL) { pick = (int)k; break; }   // fill split slabs first
        if (pick < 0) for (size_t k = 0; k < c->slabs.size(); ++k) if (!c->slabs[k].mask) { pick = (int)k; break; }
        if (pick < 0) {
            const bool capped = c->env_budget_bytes > 0 && (c->slabs.size() + 1) * slab_bytes > (size_t)c->env_budget_bytes;
            EnvSlab sl;
            if (!capped && hipMalloc((void**)&sl.base, slab_bytes) == hipSuccess) {
                c->bytes += (int64_t)slab_bytes;
                c->slabs.push_back(sl); pick = (int)c->slabs.size() - 1;
            } else {
                if (!capped) (void)hipGetLastError();
                if (env_evict_slab(c) != 0)
                    return tnml_fail(c, capped ? "environment memory: the budget of %ld MB holds no slab that could be evicted (%zu slabs of %zu MB; every one holds an operand of the operation in flight)"
                                               : "environment memory: hipMalloc failed and no slab could be evicted (env_budget %ld MB, %zu slabs of %zu MB)",
                                     c->env_budget_bytes >> 20, c->slabs.size(), slab_bytes >> 20);
                continue;
            }
        }
        EnvSlab& sl = c->slabs[pick];
        if (sl.ev_pending) HIPCK(c, hipStreamWaitEvent(consumer, sl.ev, 0));
        e.slab = pick; e.m = m; e.L = L;
        if (L == TNML_NL) { e.unit = -1; sl.mask = FULL; e.ptr = sl.base; }
        else {
            int u = 0; while (sl.mask & (1u << u)) ++u;
            e.unit = u; sl.mask |= 1u << u; e.ptr = sl.base + (size_t)u * c->small_elems * c->eesz();
        }
        return 0;
    }
}
// host -> device for environment j, started now; the compute stream is made to wait for it by env_ensure
static int env_fetch(tnml_ctx* c, int j) {
    EnvSlot& e = c->env[j];
    const int m = e.m, L = e.L;
    const size_t bytes = (size_t)L * m * c->NTp * c->eesz();
    const bool async = c->env_async && e.host_pinned;
    c->env_epoch += 1;
    if (async) TCK(env_copy_stream(c));
    TCK(slot_acquire(c, e, m, L, async ? c->copy_stream : c->stream));     // (clears on_host; the host copy stays valid until the copy below has read it)
    if (async) {
        // the unit may have been vacated by slot_release a moment ago (no event of its own): order the copy behind everything the compute
        // stream has been given so far -- work that is normally long finished, so the overlap with the current bond update stays
        HIPCK(c, hipEventRecord(c->ev_compute, c->stream));
        HIPCK(c, hipStreamWaitEvent(c->copy_stream, c->ev_compute, 0));
        HIPCK(c, hipMemcpyAsync(e.ptr, e.host, bytes, hipMemcpyHostToDevice, c->copy_stream));
        if (!e.ev) HIPCK(c, hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
        HIPCK(c, hipEventRecord(e.ev, c->copy_stream));
        e.ev_pending = true;
    } else {
        HIPCK(c, hipMemcpyAsync(e.ptr, e.host, bytes, hipMemcpyHostToDevice, c->stream));
        if (!e.host_pinned) SYNCK(c, c->stream);
    }
    c->env_fetches += 1;
    return 0;
}
// the environment of site j on the device and visible to the compute stream (no-op when it is there)
static int env_ensure(tnml_ctx* c, int j) {
    EnvSlot& e = c->env[j];
    if (e.on_host) TCK(env_fetch(c, j));
    if (e.ev_pending) { HIPCK(c, hipStreamWaitEvent(c->stream, e.ev, 0)); e.ev_pending = false; }
    return 0;
}
struct EnvProtect {                                     // the operands of one operation: resident and not evictable while it is set up
    tnml_ctx* c; int keep[4];
    EnvProtect(tnml_ctx* c_, int a, int b = 0, int d = 0, int e = 0) : c(c_) { for (int k = 0; k < 4; ++k) keep[k] = c->env_protect[k]; c->env_protect[0] = a; c->env_protect[1] = b; c->env_protect[2] = d; c->env_protect[3] = e; }
    ~EnvProtect() { for (int k = 0; k < 4; ++k) c->env_protect[k] = keep[k]; }
};
static int env_alloc(tnml_ctx* c, int j, int m, int L) { return slot_acquire(c, c->env[j], m, L); }
// Host tier, beside the bond update that is about to be enqueued: the environment the NEXT bond of the sweep will need is started on
// its way back (half 1 moves right: bond b + 1 reads the right environment of site b + 3; half 2 moves left: site b - 2), and one slab
// is kept free for the environment shiftE will build at the end of this bond update -- its eviction, if one is needed, then runs beside
// this bond update's kernels instead of in front of the shift.
static int env_lookahead(tnml_ctx* c, int b, int ha) {
    const int next = ha == 1 ? b + 3 : b - 2;
    EnvProtect keep(c, b - 1 > 0 ? b - 1 : 0, b + 2 <= c->N ? b + 2 : 0, (next >= 1 && next <= c->N) ? next : 0);
    if (next >= 1 && next <= c->N && c->env[next].on_host) { TCK(env_fetch(c, next)); c->env_prefetches += 1; }
    bool free_slab = false;
    for (const auto& sl : c->slabs) if (!sl.mask) { free_slab = true; break; }
    const size_t slab_bytes = c->big_elems * c->eesz();
    if (!free_slab && (c->slabs.size() + 1) * slab_bytes > (size_t)c->env_budget_bytes) (void)env_evict_slab(c);     // (nothing evictable: the shift will say so if it matters)
    return 0;
}
static const void* phi_site(const tnml_ctx* c, int j) { return (const char*)c->phi + (size_t)(j - 1) * 2 * c->NTp * c->eesz(); }

// dst = src*(t.A(cs)*W.A(cs)) (fixedL.cc:142-149,221-228); src == nullptr: chain end.  dst is an environment
// ([Lout][m_out][NTp] in the env type) or, with acc_out, a buffer of the arithmetic type (the last step of toverlap)
static int shift_core(tnml_ctx* c, int cs, bool from_left, const void* src, int Le, void* dst, bool acc_out, int* Lout_p) {
    const SiteT& A = c->W[cs];
    const int m_in = from_left ? A.ml : A.mr, m_out = from_left ? A.mr : A.ml;
    if (!src && m_in != 1) return tnml_fail(c, "shift: chain-end site %d has outer dimension %d", cs, m_in);
    if (Le == TNML_NL && A.L == TNML_NL) return tnml_fail(c, "shift: Label index on both env and site");
    const int Lout = A.L > Le ? A.L : Le;
    if (Lout_p) *Lout_p = Lout;
    c->env_epoch += 1;                                  // an environment is about to be written: bf16 copies made from environments are stale
    PackDesc d;
    d.TO = 1; d.L = A.L; d.st = 0; d.ss = A.ml; d.sl = (long)2 * A.ml * A.mr;
    if (from_left) { d.nx = A.ml; d.sx = 1; d.ny = A.mr; d.sy = 2 * A.ml; }
    else           { d.nx = A.mr; d.sx = 2 * A.ml; d.ny = A.ml; d.sy = 1; }
    d.Kp = ru16(2 * d.nx); d.Np = ru16(d.ny);
    if ((size_t)d.L * d.Kp * d.Np > c->sM_cap || (size_t)d.L * d.Kp * d.Np > c->mcap)
        return tnml_fail(c, "shift: packed site matrix of site %d (%d x %d x %d) exceeds the workspace", cs, d.L, d.Kp, d.Np);
    if (c->f64()) {                                     // fp64 MFMA shift (M in the free SVD workspace); fp32-stored environments (TNML_F64_E32) are rounded once, on the store
        TCK(launch_pack(c, d, A.a, c->sM, nullptr));
        Fgemm64Args f;
        f.EI = src ? src : c->ones;
        f.EI_lstride = (Le == TNML_NL) ? (size_t)m_in * c->NTp : 0;
        f.mI = m_in; f.phiI = phi_site(c, cs);
        f.M = c->sM; f.M_lstride = (A.L == TNML_NL) ? (size_t)d.Kp * d.Np : 0; f.Kp = d.Kp; f.Np = d.Np;
        f.phiO = nullptr;
        f.out = (double*)dst; f.out_lstride = (size_t)m_out * c->NTp; f.mO = m_out;
        f.NTp = c->NTp; f.L = Lout; f.env64 = c->env64(); f.out32 = !c->env64() && !acc_out;
        // the Label-carrying shift with the site matrix resident in registers (kernels_res.hip): input dimensions 33..120, output up to 128
        if (c->shift_res && c->env64() && !acc_out && src && Le == TNML_NL && A.L == 1 && shift_res_applies(m_in, m_out) && d.Np <= 128 &&
            (c->shift_res >= 2 || c->NTp >= 7680) &&
            (size_t)TNML_NL * m_in * c->NTp * sizeof(double) < ((size_t)1 << 32)) {      // (32-bit lane offsets: beyond ~447 000 images per rank the generic kernel takes over)
            ShiftResArgs sa{(const double*)src, (size_t)m_in * c->NTp, (const double*)phi_site(c, cs), c->sM, (double*)dst, (size_t)m_out * c->NTp, m_out, c->NTp, Lout, m_in, d.Kp, d.Np};
            return launch_shift_res(c, sa);
        }
        return launch_fgemm64(c, f);
    }
    TCK(launch_pack(c, d, A.a, nullptr, c->Mf));
    FgemmArgs f;
    f.EI = src ? (const float*)src : (const float*)c->ones;
    f.EI_lstride = (Le == TNML_NL) ? (size_t)m_in * c->NTp : 0;
    f.mI = m_in; f.phiI = (const float*)phi_site(c, cs);
    f.M = c->Mf; f.M_lstride = (A.L == TNML_NL) ? (size_t)d.Kp * d.Np : 0; f.Kp = d.Kp; f.Np = d.Np;
    f.phiO = nullptr;
    f.out = (float*)dst; f.out_lstride = (size_t)m_out * c->NTp; f.mO = m_out;
    f.NTp = c->NTp; f.L = Lout;
    return launch_fgemm(c, f);
}
// new env at site cs from the env at ps (0: chain end)
static int shift_site(tnml_ctx* c, int cs, int ps, bool from_left) {
    const SiteT& A = c->W[cs];
    const bool has_prev = ps >= 1 && ps <= c->N;
    if (has_prev && !c->env[ps].built()) return tnml_fail(c, "shift: environment of site %d missing", ps);
    // the source, the destination and the two environments of the bond in flight (its plan holds their addresses) stay on the device
    EnvProtect keep(c, has_prev ? ps : 0, cs, c->currb > 0 ? c->currb - 1 : 0, c->currb > 0 ? c->currb + 2 : 0);
    if (has_prev) TCK(env_ensure(c, ps));
    const int m_in = from_left ? A.ml : A.mr, m_out = from_left ? A.mr : A.ml;
    const int Le = has_prev ? c->env[ps].L : 1;
    if (has_prev && c->env[ps].m != m_in) return tnml_fail(c, "shift: env dim %d != site dim %d at site %d", c->env[ps].m, m_in, cs);
    if (Le == TNML_NL && A.L == TNML_NL) return tnml_fail(c, "shift: Label index on both env and site");
    TCK(env_alloc(c, cs, m_out, A.L > Le ? A.L : Le));
    return shift_core(c, cs, from_left, has_prev ? c->env[ps].ptr : nullptr, Le, c->env[cs].ptr, false, nullptr);
}

int tnml_env_init(tnml_ctx* c) {       // TrainStates::init, fixedL.cc:122-157
    CollScope coll_(c);                // every rank calls it in step: a rank that fails here (host tier out of memory) tells its peers at once
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (!c->data_set) return tnml_fail(c, "tnml_env_init: training data not set");
    TCK(check_W(c));
    for (int n = c->N; n >= 3; --n) TCK(shift_site(c, n, n == c->N ? 0 : n + 1, false));   // :136-153
    c->currb = -1;
    return tnml_set_bond(c, 1);                                                            // :156
}
int tnml_shift_env(tnml_ctx* c, int b, int from_left) {   // TrainStates::shiftE, fixedL.cc:192-233
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (b < 1 || b > c->N - 1) return tnml_fail(c, "tnml_shift_env: bond %d out of range", b);
    const int cs = from_left ? b : b + 1;              // :196
    const int prevc = from_left ? b - 1 : b + 2;       // :199
    TCK(shift_site(c, cs, (prevc >= 1 && prevc <= c->N) ? prevc : 0, from_left != 0));
    return 0;
}
int tnml_env_stats(tnml_ctx* c, int64_t* spills, int64_t* fetches, int64_t* slabs_on_device, int64_t* host_bytes) {
    if (spills) *spills = c->env_spills;
    if (fetches) *fetches = c->env_fetches;
    if (slabs_on_device) *slabs_on_device = (int64_t)c->slabs.size();
    if (host_bytes) { int64_t hb = 0; for (const auto& e : c->env) if (e.on_host) hb += (int64_t)e.L * e.m * c->NTp * (int64_t)c->eesz(); *host_bytes = hb; }
    return 0;
}
int tnml_env_dims(tnml_ctx* c, int j, int* m, int* has_label) {
    if (j < 1 || j > c->N || !c->env[j].built()) return tnml_fail(c, "tnml_env_dims: environment of site %d not built", j);
    *m = c->env[j].m; *has_label = c->env[j].L == TNML_NL;
    return 0;
}
int tnml_get_env(tnml_ctx* c, int j, double* E) {
    if (j < 1 || j > c->N || !c->env[j].built()) return tnml_fail(c, "tnml_get_env: environment of site %d not built", j);
    {
        EnvProtect keep(c, j, c->currb > 0 ? c->currb - 1 : 0, c->currb > 0 ? c->currb + 2 : 0);
        TCK(env_ensure(c, j));
    }
    const EnvSlot& e = c->env[j];
    const size_t ne = (size_t)e.L * e.m * c->NTp;
    std::vector<char> h(ne * c->eesz());
    SYNCK(c, c->stream);
    HIPCK(c, hipMemcpy(h.data(), e.ptr, h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < c->NT; ++i)
        for (int l = 0; l < e.L; ++l)
            for (int q = 0; q < e.m; ++q) {
                const size_t k = ((size_t)l * e.m + q) * c->NTp + i;
                E[(size_t)i * e.m * e.L + q + (size_t)e.m * l] = c->env64() ? ((const double*)h.data())[k] : (double)((const float*)h.data())[k];
            }
    return 0;
}

// ---- inference: toverlap / fullTest (util.h:19-40,123-200) ------------------------------------------
// W_n[l] = (prod_{j<c} phi_j*A_j) * (phi_c*A_c) * (prod_{j>c} phi_j*A_j) for every local image, with rolling
// chain buffers borrowed from the environment pools (the training environments are left untouched).
int tnml_classify(tnml_ctx* c, double* weights, int32_t* pred, int64_t count[TNML_NL], int64_t nincorrect[TNML_NL]) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (!c->data_set) return tnml_fail(c, "tnml_classify: image data not set");
    EnvProtect keep(c, c->currb > 0 ? c->currb - 1 : 0, c->currb > 0 ? c->currb + 2 : 0);      // the chain buffers below may evict, but not the operands of the bond that is set
    c->p_valid = false;
    TCK(check_W(c));
    EnvSlot buf[3];
    int rc = 0;
    for (int k = 0; k < 3 && !rc; ++k) rc = slot_acquire(c, buf[k], c->maxm, 1);
    auto give_back = [&]() { for (auto& b : buf) slot_release(c, b); };
    if (rc) { give_back(); return rc; }
    const int cs = c->single() ? 1 : c->c0;                // per-label variant: site 1 plays the centre, no left chain
    // right chain N -> c+1 (util.h:25-29), ping-pong between buf[0] and buf[1]
    const void* R = nullptr; int cur = 0;
    for (int j = c->N; j > cs && !rc; --j) { rc = shift_core(c, j, false, R, 1, buf[cur].ptr, false, nullptr); R = buf[cur].ptr; cur ^= 1; }
    // left chain 1 -> c-1 (util.h:32-37), ping-pong between buf[2] and the free one of the pair above
    const void* Lc = nullptr; void* lbuf[2] = {buf[2].ptr, buf[cur].ptr}; int lcur = 0;
    for (int j = 1; j < cs && !rc; ++j) { rc = shift_core(c, j, true, Lc, 1, lbuf[lcur], false, nullptr); Lc = lbuf[lcur]; lcur ^= 1; }
    // centre site: T[l][r][n] = sum_{a,s} L[a][n] phi_c[s][n] A_c[a,s,r,l], then W_n[l] = sum_r T[l][r][n] R[r][n]
    if (!rc) rc = shift_core(c, cs, true, Lc, 1, c->U, true, nullptr);
    double* tail = c->tail;
    if (!rc) {
        LdotArgs a;
        a.A = c->U; a.A_lstride = (size_t)c->W[cs].mr * c->NTp; a.Bv = R; a.a_is_env = 0;
        a.mq = c->W[cs].mr; a.NTp = c->NTp; a.label = c->label; a.nl = c->nl(); a.target = c->target();
        a.P = c->P; a.dP = nullptr; a.mode = LD_MODE_COST;
        rc = launch_labeldot(c, a, tail);
    }
    give_back();
    if (rc) return rc;
    std::vector<char> h((size_t)TNML_NL * c->NTp * c->esz());
    std::vector<int> lab(c->NTp);
    SYNCK(c, c->stream);
    HIPCK(c, hipMemcpy(h.data(), c->P, h.size(), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(lab.data(), c->label, sizeof(int) * c->NTp, hipMemcpyDeviceToHost));
    HIPCK(c, hipMemsetAsync(tail, 0, sizeof(double) * TNML_NSCAL_AR, c->stream));
    if (count) for (int l = 0; l < TNML_NL; ++l) count[l] = 0;
    if (nincorrect) for (int l = 0; l < TNML_NL; ++l) nincorrect[l] = 0;
    const int nl = c->nl();
    for (int i = 0; i < c->NT; ++i) {
        double w[TNML_NL];
        for (int l = 0; l < nl; ++l) {
            const size_t k = (size_t)l * c->NTp + i;
            w[l] = c->f64() ? ((const double*)h.data())[k] : (double)((const float*)h.data())[k];
            if (weights) weights[(size_t)i * nl + l] = w[l];
        }
        bool wrong;
        if (c->single()) {                                         // decision function f(x): pred = [f > 1/2]
            const int pl = w[0] > 0.5 ? 1 : 0;
            if (pred) pred[i] = pl;
            wrong = pl != (lab[i] == c->target() ? 1 : 0);
        } else {
            int pl = 0; double best = std::fabs(w[0]);             // argmax of |W_l|, first maximum (util.h:42-57,160-163)
            for (int l = 1; l < TNML_NL; ++l) if (std::fabs(w[l]) > best) { best = std::fabs(w[l]); pl = l; }
            if (pred) pred[i] = pl;
            wrong = pl != lab[i];
        }
        if (count) count[lab[i]] += 1;
        if (nincorrect && wrong) nincorrect[lab[i]] += 1;
    }
    return 0;
}

// ---- bond plan (TrainStates::setBond, fixedL.cc:159-190: pointer selection only) ---------------
static PackDesc bond_pack_desc(const BondPlan& p) {
    PackDesc d;
    d.TO = 2; d.L = p.LB;
    const long mL = p.mL, mR = p.mR;
    if (p.kind == 1) { d.nx = p.mR; d.sx = 4 * mL; d.ss = 2 * mL; d.ny = p.mL; d.sy = 1; d.st = mL; }
    else             { d.nx = p.mL; d.sx = 1; d.ss = mL; d.ny = p.mR; d.sy = 4 * mL; d.st = 2 * mL; }
    d.sl = 4 * mL * mR;
    d.Kp = p.Kp; d.Np = p.Np;
    return d;
}
static int set_bond_impl(tnml_ctx* c, int b);
int tnml_set_bond(tnml_ctx* c, int b) {
    CollScope coll_(c);
    const int rc = set_bond_impl(c, b);
    if (rc) { c->currb = -1; c->plan = BondPlan(); }          // no dangling environment pointers after a failed setBond: the next use has to set a bond again
    return rc;
}
static int set_bond_impl(tnml_ctx* c, int b) {
    if (b < 1 || b > c->N - 1) return tnml_fail(c, "tnml_set_bond: bond %d out of range", b);
    TCK(check_W(c));
    const int lc = b - 1, rc = b + 2;                         // :164-165
    const bool useL = lc > 0, useR = rc < c->N + 1;           // :166-167
    if (useL && !c->env[lc].built()) return tnml_fail(c, "setBond: left environment (site %d) missing", lc);
    if (useR && !c->env[rc].built()) return tnml_fail(c, "setBond: right environment (site %d) missing", rc);
    {
        EnvProtect keep(c, useL ? lc : 0, useR ? rc : 0);
        c->currb = b;                                         // (eviction keeps what is nearest to the bond that is being set)
        if (useL) TCK(env_ensure(c, lc));
        if (useR) TCK(env_ensure(c, rc));
    }
    BondPlan p;
    p.b = b; p.mL = c->W[b].ml; p.mR = c->W[b + 1].mr;
    if ((useL ? c->env[lc].m : 1) != p.mL || (useR ? c->env[rc].m : 1) != p.mR) return tnml_fail(c, "setBond: env dims do not match W at bond %d", b);
    const int LL = useL ? c->env[lc].L : 1, LR = useR ? c->env[rc].L : 1;
    const bool onB = (c->c0 == b || c->c0 == b + 1);
    const void* LE = useL ? c->env[lc].ptr : c->ones;
    const void* RE = useR ? c->env[rc].ptr : c->ones;
    if (c->single()) {          // single.h:581-596: no Label anywhere; runs the "Label on B" kernels with a label extent of 1
        p.kind = 2; p.LB = 1; p.mI = p.mL; p.mO = p.mR; p.EI = LE; p.phiI = phi_site(c, b); p.EX = RE; p.phiO = phi_site(c, b + 1);
    } else if (onB) {
        if (LL != 1 || LR != 1) return tnml_fail(c, "setBond: Label index on an environment and on B at bond %d", b);
        p.kind = 2; p.LB = TNML_NL; p.mI = p.mL; p.mO = p.mR; p.EI = LE; p.phiI = phi_site(c, b); p.EX = RE; p.phiO = phi_site(c, b + 1);
    } else if (LR == TNML_NL && LL == 1) {
        p.kind = 0; p.LB = 1; p.mI = p.mL; p.mO = p.mR; p.EI = LE; p.phiI = phi_site(c, b); p.EX = RE; p.phiO = phi_site(c, b + 1);
    } else if (LL == TNML_NL && LR == 1) {
        p.kind = 1; p.LB = 1; p.mI = p.mR; p.mO = p.mL; p.EI = RE; p.phiI = phi_site(c, b + 1); p.EX = LE; p.phiO = phi_site(c, b);
    } else {
        return tnml_fail(c, "Couldn't find Label index at bond %d", b);       // fixedL.cc:291-296,362
    }
    p.Kp = ru16(2 * p.mI); p.Np = ru16(2 * p.mO);
    if (c->bf16()) p.Kp = (2 * p.mI + 31) / 32 * 32;            // the bf16 MFMA reduces 32 indices at a time
    c->plan = p; c->currb = b;
    return 0;
}
int tnml_bond_dims(tnml_ctx* c, int b, int* mL, int* mR, int* label_on_B) {
    if (b < 1 || b > c->N - 1 || !c->W[b].set || !c->W[b + 1].set) return tnml_fail(c, "tnml_bond_dims: bad bond %d", b);
    *mL = c->W[b].ml; *mR = c->W[b + 1].mr; *label_on_B = (c->c0 == b || c->c0 == b + 1);
    return 0;
}
static size_t bond_elems(const tnml_ctx* c, int b) {
    return (size_t)c->W[b].ml * 4 * c->W[b + 1].mr * ((c->c0 == b || c->c0 == b + 1) ? TNML_NL : 1);
}
int tnml_bond_tensor(tnml_ctx* c, int b, double* B) {
    HIPCK(c, hipSetDevice(c->cfg.device));
    int mL, mR, lab; TCK(tnml_bond_dims(c, b, &mL, &mR, &lab));
    if (c->W[b].mr != c->W[b + 1].ml) return tnml_fail(c, "bond %d: link dimensions differ", b);
    TCK(launch_bond_form(c, c->W[b], c->W[b + 1], c->tB));
    SYNCK(c, c->stream);
    HIPCK(c, hipMemcpy(B, c->tB, sizeof(double) * bond_elems(c, b), hipMemcpyDeviceToHost));
    return 0;
}

// ---- per-image contractions -----------------------------------------------------------------------
// forward pass with the M-layout fp64 vector `vec` as bond tensor: P = vec*t.v, then mode-specific
// reductions into tail[0..11] (device)
static int forward_pass(tnml_ctx* c, const double* vec, int mode, double* tail, bool want_P, bool reduce = true) {      // !reduce: the partial sums stay in c->partials[c->part_n][12]
    const BondPlan& p = c->plan;
    const size_t ustride = (size_t)p.mO * c->NTp;
    LdotArgs a;
    if (p.kind == 2) { a.A = c->U; a.A_lstride = ustride; a.Bv = p.EX; a.a_is_env = 0; }
    else             { a.A = p.EX; a.A_lstride = ustride; a.Bv = c->U; a.a_is_env = 1; }
    a.mq = p.mO; a.NTp = c->NTp; a.label = c->label; a.nl = c->nl(); a.target = c->target();
    a.P = want_P ? (mode == LD_MODE_PAP ? c->Pp : c->P) : nullptr; a.dP = (mode == LD_MODE_PAP) ? nullptr : c->dP; a.mode = mode;
    if (c->f64()) {
        Fgemm64Args f;
        f.EI = p.EI; f.EI_lstride = 0; f.mI = p.mI; f.phiI = p.phiI;
        f.M = vec; f.M_lstride = p.kind == 2 ? (size_t)p.Kp * p.Np : 0; f.Kp = p.Kp; f.Np = p.Np;
        f.phiO = p.phiO;
        f.out = (double*)c->U; f.out_lstride = ustride; f.mO = p.mO;
        f.NTp = c->NTp; f.L = p.LB; f.env64 = c->env64();
        // the bond matrix resident in the registers of a pair of workgroups (kernels_res.hip): from 7 680 images per rank on (the 7 500-image
        // shard of an 8-GPU run: 0.162 ms per bond update against 0.250 for the feature GEMM + label dot pair, profiles/r04_shard7500_res_kernels.txt)
        if (c->fwd_res && c->Ppart && c->env64() && !c->single() && p.kind != 2 && fwd_res_applies(p.mI, p.mO) &&
            (c->fwd_res >= 2 || c->NTp >= 7680) &&
            (size_t)TNML_NL * ustride * sizeof(double) < ((size_t)1 << 32)) {           // (32-bit lane offsets of k_fwd_res: larger shards fall through to the kernels below)
            FwdResArgs fr{(const double*)p.EI, (const double*)p.phiI, vec, (const double*)p.phiO, (const double*)p.EX, ustride, c->NTp, c->NTp / 32, c->Ppart, p.mI, p.mO, p.Kp, p.Np};
            TCK(launch_fwd_res(c, fr));
            PfinishArgs pf{2, c->Ppart, nullptr, nullptr, nullptr, nullptr, c->label, c->NTp, (double*)a.P, (double*)a.dP, mode, c->partials, c->counters, tail, mode == LD_MODE_PAP ? 1 : 0};
            TCK(launch_pfinish(c, pf));
            c->part_n = c->NTp / 64;
            return reduce ? launch_labeldot_reduce(c, c->NTp / 64, tail, mode == LD_MODE_PAP ? 1 : 0) : 0;
        }
        // one persistent kernel for both halves of B*t.v where it pays (kernels_fused.hip)
        if (c->fused_fwd && c->env64() && !c->single() && p.kind != 2 && p.Kp == 240 && p.Np == 240 && p.mI == 120 && p.mO == 120 &&
            (c->fused_fwd >= 2 || c->NTp / 64 >= 224)) {
            FwdFusedArgs ff;
            ff.EI = (const double*)p.EI; ff.mI = p.mI; ff.phiI = (const double*)p.phiI; ff.M = vec; ff.Kp = p.Kp; ff.Np = p.Np;
            ff.phiO = (const double*)p.phiO; ff.EL = (const double*)p.EX; ff.EL_lstride = ustride; ff.mO = p.mO; ff.NTp = c->NTp; ff.ntiles = c->NTp / 64;
            ff.label = c->label; ff.P = (double*)a.P; ff.dP = (double*)a.dP; ff.mode = mode; ff.partials = c->partials;
            TCK(launch_fwd_fused(c, ff));
            c->part_n = ff.ntiles;
            return reduce ? launch_labeldot_reduce(c, ff.ntiles, tail, mode == LD_MODE_PAP ? 1 : 0) : 0;
        }
        TCK(launch_fgemm64(c, f));
    } else {
        if (c->bf16() && c->bf16_once && c->ebt && p.kind != 2) {        // operands converted once per bond / per launch (kernels_bf16e.hip)
            TCK(launch_fgemm_bf16e(c, (const float*)p.EI, p.mI, (const float*)p.phiI, vec, p.Kp, p.Np, (const float*)p.phiO, (float*)c->U, p.mO));
        } else {
            TCK(launch_cvt(c, vec, c->Mf, p.msize()));
            FgemmArgs f;
            f.EI = (const float*)p.EI; f.EI_lstride = 0; f.mI = p.mI; f.phiI = (const float*)p.phiI;
            f.M = c->Mf; f.M_lstride = p.kind == 2 ? (size_t)p.Kp * p.Np : 0; f.Kp = p.Kp; f.Np = p.Np;
            f.phiO = (const float*)p.phiO;
            f.out = (float*)c->U; f.out_lstride = ustride; f.mO = p.mO;
            f.NTp = c->NTp; f.L = p.LB;
            TCK(launch_fgemm(c, f));
        }
    }
    return launch_labeldot(c, a, tail, reduce);
}
// G = sum_n dP_n*dag(t.v) over all ranks for the bond tensor in vB; cost partials ride in the tail.
// weights_pp: the image sum A p = sum_n (p.v_n) v_n instead, weights p.v_n as left in Pp by the pAp pass (fast_conj of the per-label
// variant, single.h:347-379, and the merged CG of every variant); the tail is left as it is
static int grad_eval(tnml_ctx* c, bool from_P_update = false, bool outputs_current = false, bool weights_pp = false, bool reduce = true, bool fold = false, bool p_updated = false) {
    const BondPlan& p = c->plan;
    const size_t n = p.msize();
    if (weights_pp) {}
    else if (outputs_current)    { if (!c->tail_zeroed) HIPCK(c, hipMemsetAsync(c->tail, 0, sizeof(double) * TNML_NSCAL_AR, c->stream)); }   // P/dP already hold B*t.v and the residuals (the pack kernel of tnml_bond_update has cleared the tail)
    else if (from_P_update) { if (!p_updated) TCK(launch_pupdate(c, c->scal + SC_ALPHA, c->tail, !fold)); }   // P += a (p*t.v): no GEMM (p_updated: the CG step kernel has done it)
    else                    TCK(forward_pass(c, c->vB, LD_MODE_COST, c->tail, c->fast_cg)); // keeps P when fast CG is on
    const void* wsrc = weights_pp ? c->Pp : c->dP;           // the per-image weights of the sum
    const bool fuse = c->f64() && c->fuse_z && p.kind != 2;
    if (p.kind != 2 && !fuse) TCK(launch_zprime(c, p.EX, (size_t)p.mO * c->NTp, wsrc, c->Zp, p.mO, c->NTp));
    if (c->f64()) {
        Bgemm64Args g;
        g.EL = nullptr; g.EL_lstride = 0; g.dPz = nullptr; g.env64 = c->env64();
        g.EI = p.EI; g.mI = p.mI; g.phiI = p.phiI; g.phiO = p.phiO; g.mO = p.mO;
        g.Kp = p.Kp; g.Np = p.Np; g.NTp = c->NTp; g.L = p.LB;
        if (p.kind == 2) { g.Zq64 = nullptr; g.Zq32 = p.EX; g.w = (const double*)wsrc; g.w_lstride = c->NTp; }
        else if (fuse)   { g.Zq64 = nullptr; g.Zq32 = nullptr; g.w = nullptr; g.w_lstride = 0; g.EL = p.EX; g.EL_lstride = (size_t)p.mO * c->NTp; g.dPz = (const double*)wsrc; }
        else             { g.Zq64 = (const double*)c->Zp; g.Zq32 = nullptr; g.w = nullptr; g.w_lstride = 0; }
        TCK(launch_bgemm64(c, g, c->vG));
    } else {
        BgemmArgs g;
        g.EI = (const float*)p.EI; g.mI = p.mI; g.phiI = (const float*)p.phiI; g.phiO = (const float*)p.phiO; g.mO = p.mO;
        g.Kp = p.Kp; g.Np = p.Np; g.NTp = c->NTp; g.L = p.LB; g.bf16 = c->bf16() && c->bf16_grad ? c->bf16() : 0;
        if (p.kind == 2) { g.Zq = (const float*)p.EX; g.w = (const float*)wsrc; g.w_lstride = c->NTp; }
        else             { g.Zq = (const float*)c->Zp; g.w = nullptr; g.w_lstride = 0; }
        TCK(launch_bgemm(c, g, c->vG));
    }
    return reduce ? allreduce_packed(c, n) : 0;
}
static int read_scal(tnml_ctx* c, const double* dev, int count, double* host_out) {
    double* h = c->h_scal + 2 * c->svd_n + 64;
    if (count > SC_N + 4 * TNML_MAX_PASS) return tnml_fail(c, "read_scal: count too large");
    HIPCK(c, hipMemcpyAsync(h, dev, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    SYNCK(c, c->stream);
    memcpy(host_out, h, sizeof(double) * count);
    return 0;
}

// cgrad, fixedL.cc:349-445, on the bond tensor in vB (M-layout)
// issues the whole CG without a host round trip: the |r| < cconv exit (fixedL.cc:432-436) is a device
// flag that turns the state-changing kernels of later passes into no-ops; the per-pass numbers the
// reference prints are collected in a device trace and fetched once by cgrad_fetch_trace().
static int cgrad_device(tnml_ctx* c, int npass, double lambda, double cconv, bool outputs_current = false) {
    if (npass < 1 || npass > TNML_MAX_PASS) return tnml_fail(c, "cgrad: Npass must be in 1..%d", TNML_MAX_PASS);
    const size_t n = c->plan.msize();
    const bool fastc = c->single() && c->cg_method == 1;   // method = fast_conj of the per-label variant (single.h:290-398)
    // Merged passes (one all-reduce per pass instead of two): with P <- P + a (p*t.v) already in use, the image sum of a pass can be
    // A p = sum_n (p.v_n) v_n -- formed from the pAp pass's own outputs BEFORE alpha is known -- so that it travels with
    // sum_n |p.v_n|^2; the residual then follows nr = r - a (A p + lambda p) instead of being re-summed from the new dP (the same
    // algebra; rounding differs at 1e-16 |r| per pass).  The cost partials of a pass's update ride in the NEXT pass's all-reduce.
    // It is used where it buys something -- when the sum over images is also a sum over ranks (merged_cg = 1) -- because the
    // recurrence is not the reference's literal order: on the reference's own, badly conditioned feature map the fourth step size of
    // a Label-on-B bond moves by 1e-3 (the cost by 1e-10); merged_cg = 2 forces it on a single rank (parity tests), 0 disables it.
    const bool merged = c->fast_cg && !fastc && (c->merged_cg >= 2 || (c->merged_cg == 1 && c->multi()));
    // one rank, literal pass order: the per-block partial sums of a pAp pass / an output update are summed by the CG step kernel that
    // consumes them (k_cg_step2: sum |p.v|^2, k_cg_resid2: the cost of the trace) -- seven k_reduce_partials launches less per bond update
    const bool fold = c->fold_reduce && !c->multi() && !merged && !fastc && c->fast_cg;
    // one rank, fp64: the slab reduction of every gradient GEMM is folded into the CG vector kernel that consumes G, the output update
    // P <- P + a (p*t.v) rides in the CG step kernel, and k_cg_init2's work is split between its neighbours (round 5: eight launches less)
    c->defer_slab = fold && c->f64();
    const bool step_updates = fold && c->f64();
    int rc = grad_eval(c, false, outputs_current);       // :374-385
    if (!rc) rc = launch_cg_init(c, n, lambda, c->single() ? cconv : -1.);   // :386-388 (single.h:200-208 with the entry check)
    for (int pass = 1; !rc && pass <= npass; ++pass) {   // :389
        c->cg_pass = pass;
        rc = forward_pass(c, c->vP, LD_MODE_PAP, c->tail, c->fast_cg || fastc, !fold);   // :394-401 (keeps p*t.v for the fast update)
        if (rc) break;
        if (merged && pass < npass) rc = grad_eval(c, false, false, true);        // A p, all-reduced with the tail
        else if (!fold) rc = allreduce(c, c->tail, TNML_NSCAL_AR);                 // :402
        if (rc) break;
        const bool upd = step_updates && pass < npass;
        const int npp = c->part_n;                       // rows of the pAp pass's partial sums (the update below re-sets part_n)
        rc = launch_cg_step(c, n, lambda, pass, merged, fold ? c->partials : nullptr, npp, upd); // :403-407
        if (rc || pass == npass) break;                  // :409
        if (merged) {
            rc = launch_pupdate(c, c->scal + SC_ALPHA, c->tail);                  // P, dP and the cost partials of the new B (:414-420, without the GEMM)
        } else if (fastc) {                              // single.h:347-379: A p from the p.v of this pass, residual by recurrence
            rc = grad_eval(c, false, false, true);
            if (!rc) rc = launch_cg_fast_resid0(c, n, pass);
        } else rc = grad_eval(c, c->fast_cg, false, false, true, fold, upd);      // :412-421
        if (rc) break;
        rc = launch_cg_resid(c, n, lambda, cconv, pass, merged, fold ? (upd ? c->partials2 : c->partials) : nullptr, c->part_n); // :422-428, :432-436, :442
    }
    c->defer_slab = false; c->slab_pending = 0;
    return rc;
}
// the CG's device scalars and per-pass trace: enqueue the copies, parse after any later synchronisation of the stream
// slot >= 0: into the report block of that bond update in flight (parsed by tnml_bond_update_end: two may be in flight)
static double* trace_host(tnml_ctx* c, int slot) { return slot >= 0 ? c->hrep + (size_t)slot * c->hrep_stride + c->svd_n + 8 : c->h_scal + 2 * c->svd_n + 64; }
static int cgrad_trace_enqueue(tnml_ctx* c, int slot = -1) {
    HIPCK(c, hipMemcpyAsync(trace_host(c, slot), c->scal, sizeof(double) * (SC_N + 4 * TNML_MAX_PASS), hipMemcpyDeviceToHost, c->stream));
    return 0;
}
static void cgrad_trace_parse(tnml_ctx* c, int npass, tnml_cg_trace* tr, int slot = -1) {
    memset(tr, 0, sizeof *tr);
    const double* hp = trace_host(c, slot);
    const int done = (int)llround(hp[SC_NPASS]);
    tr->npass_done = done;
    tr->converged = (int)llround(hp[SC_CONV]);
    for (int p = 0; p < done && p < npass; ++p) {
        const double* t = hp + SC_N + 4 * p;
        tr->pAp[p] = t[0]; tr->alpha[p] = t[1]; tr->cost[p] = t[2]; tr->rnorm[p] = t[3];
    }
}
static int cgrad_fetch_trace(tnml_ctx* c, int npass, tnml_cg_trace* tr) {
    if (!tr) return 0;
    TCK(cgrad_trace_enqueue(c));
    SYNCK(c, c->stream);
    cgrad_trace_parse(c, npass, tr);
    return 0;
}
// quadcost, fixedL.cc:280-344, on the bond tensor in vB
// launches only: cost partials, #correct and |B|^2 end up in the 13 doubles behind G (t[0..9] per-label costs, t[10] ncorrect, t[12] |B|^2)
static int quadcost_launch(tnml_ctx* c, bool want_P) {
    const size_t n = c->plan.msize();
    TCK(forward_pass(c, c->vB, LD_MODE_COST, c->tail, want_P));
    TCK(allreduce(c, c->tail, TNML_NSCAL_AR));
    TCK(launch_sqnorm(c, c->vB, n, c->tail + 12));              // |B|^2 rides behind the cost partials (local: written after the reduction)
    return 0;
}
static void quadcost_parse(tnml_ctx* c, const double* t, double lambda, double* cost, double* label_cost, double* reg_cost, int64_t* ncorrect) {
    const double bn2 = t[12];
    c->last_bnorm = std::sqrt(bn2);
    const double CR = lambda * bn2;                       // :329
    double C = 0.;
    for (int l = 0; l < TNML_NL; ++l) { if (label_cost) label_cost[l] = t[l]; C += t[l]; }   // :331-336
    C += CR;                                              // :338
    if (cost) *cost = C;
    if (reg_cost) *reg_cost = CR;
    if (ncorrect) *ncorrect = (int64_t)llround(t[SC_NCORR]);
}
static int quadcost_device(tnml_ctx* c, double lambda, double* cost, double* label_cost, double* reg_cost, int64_t* ncorrect, bool want_P) {
    TCK(quadcost_launch(c, want_P));
    double t[13];
    TCK(read_scal(c, c->tail, 13, t));
    quadcost_parse(c, t, lambda, cost, label_cost, reg_cost, ncorrect);
    return 0;
}

// One-sided Jacobi (Hestenes) SVD of a tall column-major matrix A (R x C, R >= C), in place on the host: on return column j of A
// is u_j s_j, V (C x C) holds the right singular vectors, s the singular values (unsorted).  Small singular values keep their
// relative accuracy, which the pcut test of the exact solver needs (a Gram matrix loses everything below sqrt(eps) s_max).
static bool hestenes_svd(int R, int C, double* A, double* sv, double* V) {      // false: 60 sweeps did not converge
    for (int j = 0; j < C; ++j) for (int i = 0; i < C; ++i) V[i + (size_t)C * j] = i == j ? 1. : 0.;
    bool converged = false;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < C - 1; ++p)
            for (int q = p + 1; q < C; ++q) {
                double* ap = A + (size_t)R * p; double* aq = A + (size_t)R * q;
                double alpha = 0., beta = 0., gamma = 0.;
                for (int i = 0; i < R; ++i) { alpha += ap[i] * ap[i]; beta += aq[i] * aq[i]; gamma += ap[i] * aq[i]; }
                if (!(std::fabs(gamma) > 1e-15 * std::sqrt(alpha * beta))) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2. * gamma);
                const double t = (zeta >= 0. ? 1. : -1.) / (std::fabs(zeta) + std::sqrt(1. + zeta * zeta));
                const double cs = 1. / std::sqrt(1. + t * t), sn = cs * t;
                for (int i = 0; i < R; ++i) { const double x = ap[i], y = aq[i]; ap[i] = cs * x - sn * y; aq[i] = sn * x + cs * y; }
                double* vp = V + (size_t)C * p; double* vq = V + (size_t)C * q;
                for (int i = 0; i < C; ++i) { const double x = vp[i], y = vq[i]; vp[i] = cs * x - sn * y; vq[i] = sn * x + cs * y; }
            }
        if (!rotated) { converged = true; break; }
    }
    for (int j = 0; j < C; ++j) { double t = 0.; const double* a = A + (size_t)R * j; for (int i = 0; i < R; ++i) t += a[i] * a[i]; sv[j] = std::sqrt(t); }
    return converged;
}
// exact (single.h:117-160, per-label variant): B = y Phi^+ with the filtered inverse s/(s^2 + lambda) above pcut, Phi = [v_1 ... v_NT]
// (D x NT, D = 4 mL mR).  "Only works for rather small number of training samples" (single.h:114).  The dense per-image tensors
// are not formed on the device here either: row j of Phi is the output vector of ONE forward pass of the unit tensor e_j
// (p.v_n = v_n[j], the pAp pass of the CG).  The D x NT matrix then goes to the host, whose one-sided Jacobi SVD keeps the small
// singular values accurate enough for the reference's `s > pcut` test (pcut = 1E-8 by default).  Result in vB (M-layout) and tB
// (ITensor layout).  One rank only: the images of other ranks would have to be gathered.
static int exact_device(tnml_ctx* c, double lambda, double pcut) {
    if (!c->single()) return tnml_fail(c, "exact: only the per-label variant (TNML_MODE_SINGLE) has this solver");
    if (c->cfg.dtype != TNML_F64) return tnml_fail(c, "exact: TNML_F64 contexts only");
    if (c->cfg.nranks > 1) return tnml_fail(c, "exact: one rank only (the design matrix of all images is needed in one place)");
    const BondPlan p = c->plan;
    const PackDesc pd = bond_pack_desc(p);
    const int D = p.mL * 4 * p.mR, NT = c->NT;
    // the one-sided Jacobi below costs ~6 min(D, NT)^2 max(D, NT) flops per sweep on ONE host thread and needs up to a few dozen sweeps
    if (D > 4096 || (double)D * NT > 4e8 || (double)std::min(D, NT) * std::min(D, NT) * std::max(D, NT) > 2e10)
        return tnml_fail(c, "exact: %d unknowns x %d images -- the dense solver is meant for small problems (\"Only works for rather small number of training samples\", single.h:114)", D, NT);
    std::vector<double> Pt((size_t)NT * D);                             // Phi^T, column j = row j of Phi
    std::vector<int> lab((size_t)NT);
    HIPCK(c, hipMemcpyAsync(lab.data(), c->label, sizeof(int) * (size_t)NT, hipMemcpyDeviceToHost, c->stream));
    for (int j = 0; j < D; ++j) {
        HIPCK(c, hipMemsetAsync(c->tB2, 0, sizeof(double) * D, c->stream));
        TCK(launch_fill_f64(c, c->tB2 + j, 1.0, 1));
        TCK(launch_pack(c, pd, c->tB2, c->vP, nullptr));
        TCK(forward_pass(c, c->vP, LD_MODE_PAP, c->tail, true));      // Pp[n] = v_n . e_j
        HIPCK(c, hipMemcpyAsync(Pt.data() + (size_t)NT * j, c->Pp, sizeof(double) * (size_t)NT, hipMemcpyDeviceToHost, c->stream));
    }
    SYNCK(c, c->stream);
    std::vector<double> hB((size_t)D, 0.);
    const int tgt = c->target();
    if (NT >= D) {                                                      // Phi^T = U S V^T: columns u_j s_j (images), V in tensor space
        std::vector<double> V((size_t)D * D), sv((size_t)D);
        if (!hestenes_svd(NT, D, Pt.data(), sv.data(), V.data())) return tnml_fail(c, "exact: the Jacobi SVD of the %d x %d design matrix did not converge in 60 sweeps", NT, D);
        for (int j = 0; j < D; ++j) {
            const double s1 = sv[j];
            if (!(s1 > pcut)) continue;                                 // pseudoInv, single.h:145-153
            double yu = 0.;
            const double* u = Pt.data() + (size_t)NT * j;
            for (int i = 0; i < NT; ++i) if (lab[i] == tgt) yu += u[i];  // y . (u_j s_j)
            const double f = yu / (s1 * s1 + lambda);                   // (y.u_j) s/(s^2+lambda) = (y.u_j s_j)/(s^2+lambda)
            const double* v = V.data() + (size_t)D * j;
            for (int k = 0; k < D; ++k) hB[k] += f * v[k];
        }
    } else {                                                            // more unknowns than images: Phi = V' S U'^T on the D x NT matrix
        std::vector<double> Ph((size_t)D * NT), U((size_t)NT * NT), sv((size_t)NT);
        for (int j = 0; j < D; ++j) for (int i = 0; i < NT; ++i) Ph[j + (size_t)D * i] = Pt[i + (size_t)NT * j];
        if (!hestenes_svd(D, NT, Ph.data(), sv.data(), U.data())) return tnml_fail(c, "exact: the Jacobi SVD of the %d x %d design matrix did not converge in 60 sweeps", D, NT);
        for (int j = 0; j < NT; ++j) {
            const double s1 = sv[j];
            if (!(s1 > pcut)) continue;
            double yu = 0.;
            const double* u = U.data() + (size_t)NT * j;
            for (int i = 0; i < NT; ++i) if (lab[i] == tgt) yu += u[i];
            const double f = yu / (s1 * s1 + lambda);                   // (y.u'_j) s/(s^2+lambda) v'_j with v'_j = column / s
            const double* vs = Ph.data() + (size_t)D * j;
            for (int k = 0; k < D; ++k) hB[k] += f * vs[k];
        }
    }
    HIPCK(c, hipMemcpyAsync(c->tB, hB.data(), sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
    TCK(launch_pack(c, pd, c->tB, c->vB, nullptr));
    SYNCK(c, c->stream);                          // hB is a local
    return 0;
}

static int upload_bond(tnml_ctx* c, const double* B) {     // host ITensor layout -> tB, vB
    if (c->currb < 1) return tnml_fail(c, "setBond has not been called");
    const BondPlan& p = c->plan;
    const size_t ne = (size_t)p.mL * 4 * p.mR * p.LB;
    HIPCK(c, hipMemcpyAsync(c->tB, B, sizeof(double) * ne, hipMemcpyHostToDevice, c->stream));
    return launch_pack(c, bond_pack_desc(p), c->tB, c->vB, nullptr);
}
static int download_bond(tnml_ctx* c, const double* Mvec, double* B) {   // M-layout -> host ITensor layout
    const BondPlan& p = c->plan;
    const size_t ne = (size_t)p.mL * 4 * p.mR * p.LB;
    TCK(launch_unpack(c, bond_pack_desc(p), Mvec, c->tB2));
    SYNCK(c, c->stream);
    HIPCK(c, hipMemcpy(B, c->tB2, sizeof(double) * ne, hipMemcpyDeviceToHost));
    return 0;
}

int tnml_forward(tnml_ctx* c, const double* B, double* P) {
    HIPCK(c, hipSetDevice(c->cfg.device));
    c->p_valid = false;
    TCK(upload_bond(c, B));
    TCK(forward_pass(c, c->vB, LD_MODE_COST, c->tail, true));
    const size_t ne = (size_t)TNML_NL * c->NTp;
    std::vector<char> h(ne * c->esz());
    SYNCK(c, c->stream);
    HIPCK(c, hipMemcpy(h.data(), c->P, h.size(), hipMemcpyDeviceToHost));
    const int nl = c->nl();                                // [NT][10], or [NT] in the per-label variant
    for (int i = 0; i < c->NT; ++i)
        for (int l = 0; l < nl; ++l)
            P[(size_t)i * nl + l] = c->f64() ? ((const double*)h.data())[(size_t)l * c->NTp + i] : (double)((const float*)h.data())[(size_t)l * c->NTp + i];
    return 0;
}
int tnml_gradient(tnml_ctx* c, const double* B, double* G) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    c->p_valid = false;
    TCK(upload_bond(c, B));
    TCK(grad_eval(c));
    return download_bond(c, c->vG, G);
}
// sum_n |p.v_n|^2 + lambda |p|^2 for a direction p (fixedL.cc:394-403), collective
int tnml_pAp(tnml_ctx* c, const double* p, double lambda, double* pAp) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    c->p_valid = false;
    TCK(upload_bond(c, p));
    const size_t n = c->plan.msize();
    TCK(forward_pass(c, c->vB, LD_MODE_PAP, c->tail, false));
    TCK(allreduce(c, c->tail, TNML_NSCAL_AR));
    TCK(launch_sqnorm(c, c->vB, n, c->tail + 12));
    double t[13];
    TCK(read_scal(c, c->tail, 13, t));
    if (pAp) *pAp = t[SC_PP] + lambda * t[12];
    return 0;
}
int tnml_quadcost(tnml_ctx* c, const double* B, double lambda, double* cost, double label_cost[TNML_NL], double* reg_cost, int64_t* ncorrect) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    c->p_valid = false;
    TCK(upload_bond(c, B));
    return quadcost_device(c, lambda, cost, label_cost, reg_cost, ncorrect, false);
}
// pinv (single.h:404-517, per-label variant): subspace iteration on A = sum_n v_n v_n^T from the start V0 (D x r, columns in ITensor
// order).  E_k = A V_k is the CG's own pair of passes -- a forward pass of V_k (its outputs V_k.v_n stay in Pp) and the gradient GEMM
// weighted with them -- so the dense v_n are not formed here either; the r x r algebra (polar factor, SVD of E through a one-sided Jacobi
// on its r columns, the filtered inverse) runs on the host.  The reference starts from a time-seeded random V and only prints the cost of
// the result (single.h:596-601): a diagnostic, with the start an argument here.  One rank only.
int tnml_pinv(tnml_ctx* c, const double* V0, int r, int npass, double lambda, double pcut, double* B, double* ve, int* npass_done, double* Dsv) {
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (!c->single()) return tnml_fail(c, "tnml_pinv: only the per-label variant (TNML_MODE_SINGLE) has this solver");
    if (c->currb < 1) return tnml_fail(c, "tnml_pinv: setBond has not been called");
    if (c->cfg.nranks > 1) return tnml_fail(c, "tnml_pinv: one rank only");
    const BondPlan p = c->plan;
    const int D = p.mL * 4 * p.mR, NT = c->NT;
    if (r < 1 || r > D || r > 64) return tnml_fail(c, "tnml_pinv: Ntarget = %d outside 1..min(%d, 64)", r, D);
    if (npass < 0) return tnml_fail(c, "tnml_pinv: Npass must be >= 0");
    c->p_valid = false;
    std::vector<double> V((size_t)D * r), E((size_t)D * r), A((size_t)D * r), sv((size_t)r), W((size_t)r * r), hp((size_t)c->NTp);
    std::vector<int> lab((size_t)NT);
    HIPCK(c, hipMemcpy(lab.data(), c->label, sizeof(int) * NT, hipMemcpyDeviceToHost));
    // V = polarU(V0) (:458): V0 W = U S  ->  U W^T
    A.assign(V0, V0 + (size_t)D * r);
    if (!hestenes_svd(D, r, A.data(), sv.data(), W.data())) return tnml_fail(c, "tnml_pinv: the Jacobi SVD of the start did not converge");
    auto polar_from = [&](std::vector<double>& out) {                  // columns of A are u_g s_g, W the right vectors: out = U W^T
        for (int g = 0; g < r; ++g) if (!(sv[g] > 0.)) return false;
        for (int k = 0; k < r; ++k) for (int d = 0; d < D; ++d) { double t = 0.; for (int g = 0; g < r; ++g) t += A[d + (size_t)D * g] / sv[g] * W[k + (size_t)r * g]; out[d + (size_t)D * k] = t; }
        return true;
    };
    if (!polar_from(V)) return tnml_fail(c, "tnml_pinv: the start V0 has linearly dependent columns");
    std::vector<double> yus((size_t)r);
    auto make_E = [&](bool want_yus) -> int {                          // E_k = A V_k for all k (:469-473, :482-486); optionally yUS_k (:513-518)
        for (int k = 0; k < r; ++k) {
            TCK(upload_bond(c, V.data() + (size_t)D * k));
            HIPCK(c, hipMemcpyAsync(c->vP, c->vB, sizeof(double) * p.msize(), hipMemcpyDeviceToDevice, c->stream));
            TCK(forward_pass(c, c->vP, LD_MODE_PAP, c->tail, true));   // Pp[n] = V_k . v_n
            if (want_yus) {
                HIPCK(c, hipMemcpyAsync(hp.data(), c->Pp, sizeof(double) * c->NTp, hipMemcpyDeviceToHost, c->stream));
                SYNCK(c, c->stream);
                double t = 0.; for (int n = 0; n < NT; ++n) if (lab[n] == c->target()) t += hp[n];
                yus[k] = t;
            } else {
                TCK(grad_eval(c, false, false, true, false));          // vG = sum_n (V_k . v_n) v_n
                TCK(download_bond(c, c->vG, E.data() + (size_t)D * k));
            }
        }
        return 0;
    };
    auto dotVE = [&]() { double t = 0.; for (size_t i = 0; i < (size_t)D * r; ++i) t += V[i] * E[i]; return t; };
    TCK(make_E(false));
    double last = dotVE();                                             // :475
    if (ve) ve[0] = last;
    int done = 0;
    for (int pass = 1; pass <= npass; ++pass) {
        TCK(make_E(false));
        A = E;                                                         // E^T (D x r): columns E_k; E W = U S -> F[a][g] = W[a + r g], G[g][:] = U[:, g]
        if (!hestenes_svd(D, r, A.data(), sv.data(), W.data())) return tnml_fail(c, "tnml_pinv: the Jacobi SVD of E did not converge");
        // sort by singular value (descending) so that D reads like the reference's PrintData(D)
        std::vector<int> ord((size_t)r); for (int g = 0; g < r; ++g) ord[g] = g;
        std::sort(ord.begin(), ord.end(), [&](int x, int y) { return sv[x] > sv[y]; });
        std::vector<double> A2((size_t)D * r), W2((size_t)r * r), s2((size_t)r);
        for (int g = 0; g < r; ++g) { s2[g] = sv[ord[g]]; std::copy(A.begin() + (size_t)D * ord[g], A.begin() + (size_t)D * (ord[g] + 1), A2.begin() + (size_t)D * g); std::copy(W.begin() + (size_t)r * ord[g], W.begin() + (size_t)r * (ord[g] + 1), W2.begin() + (size_t)r * g); }
        A.swap(A2); W.swap(W2); sv.swap(s2);
        if (!polar_from(V)) {                                          // rank-deficient E: the polar factor over the non-zero part only
            for (int k = 0; k < r; ++k) for (int d = 0; d < D; ++d) { double t = 0.; for (int g = 0; g < r; ++g) if (sv[g] > 0.) t += A[d + (size_t)D * g] / sv[g] * W[k + (size_t)r * g]; V[d + (size_t)D * k] = t; }
        }
        const double VE = dotVE();                                     // :497 (= sum of the singular values)
        done = pass;
        if (ve) ve[pass] = VE;
        if (std::fabs(VE - last) < 1E-4) break;                        // :500
        last = VE;
    }
    if (npass_done) *npass_done = done;
    std::fill(B, B + D, 0.);
    if (done > 0) {
        if (Dsv) std::copy(sv.begin(), sv.end(), Dsv);
        TCK(make_E(true));                                             // yUS with the V of the last pass
        for (int a = 0; a < r; ++a)
            for (int g = 0; g < r; ++g) {
                const double s1 = sv[g];
                if (!(s1 > pcut)) continue;                            // pseudoInv :417-421
                const double cf = yus[a] * W[a + (size_t)r * g] / (s1 * s1 + lambda);   // F[a][g] s/(s^2+lambda) G[g][:], G = column / s
                for (int d = 0; d < D; ++d) B[d] += cf * A[d + (size_t)D * g];
            }
    }
    return 0;
}
int tnml_exact(tnml_ctx* c, double* B, double lambda, double pcut) {   // single.h:117-160 on the bond chosen by tnml_set_bond
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (c->currb < 1) return tnml_fail(c, "tnml_exact: setBond has not been called");
    c->p_valid = false;
    TCK(exact_device(c, lambda, pcut));
    return download_bond(c, c->vB, B);
}
int tnml_set_option_real(tnml_ctx* c, const char* name, double value) {
    if (!strcmp(name, "pcut")) { if (!(value >= 0.)) return tnml_fail(c, "pcut must be >= 0"); c->pcut = value; return 0; }
    if (!strcmp(name, "noise")) {                                     // single.cc:25,222: the noise of every sweep
        if (!(value >= 0.)) return tnml_fail(c, "noise must be >= 0");
        if (value >= 1e-14 && !c->single()) return tnml_fail(c, "noise: the density-matrix split exists in the per-label variant only (single.h:648-672)");
        if (value >= 1e-14 && !c->env64()) return tnml_fail(c, "noise: needs fp64 environments (dtype f64)");
        c->noise = value; return 0;
    }
    return tnml_fail(c, "tnml_set_option_real: unknown option %s", name);
}
int tnml_cgrad(tnml_ctx* c, double* B, int npass, double lambda, double cconv, tnml_cg_trace* trace) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    c->p_valid = false;
    TCK(upload_bond(c, B));
    TCK(cgrad_device(c, npass, lambda, cconv));
    TCK(cgrad_fetch_trace(c, npass, trace));
    return download_bond(c, c->vB, B);
}
int tnml_svd_split(tnml_ctx* c, const double* B, int b, int ha, double cutoff, int maxm, int minm,
                   double* truncerr, int* newm, double* sv, int* nsv) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (b < 1 || b > c->N - 1 || (ha != 1 && ha != 2)) return tnml_fail(c, "tnml_svd_split: bad bond/half");
    c->p_valid = false;
    HIPCK(c, hipMemcpyAsync(c->tB, B, sizeof(double) * bond_elems(c, b), hipMemcpyHostToDevice, c->stream));
    TCK(svd_split_device(c, c->tB, b, ha, cutoff, maxm, minm, truncerr, newm, sv, nsv));
    c->currb = -1;
    return 0;
}

// ---- one iteration of the mldmrg loop body (fixedL.cc:478-540) ------------------------------------
// One iteration of the mldmrg loop body in two halves, so that a sweep can keep the GPU queue full across bond boundaries:
// tnml_bond_update_begin enqueues the whole bond update (it blocks once, inside the split, for the eigenvalues that fix the
// new bond dimension) and returns; tnml_bond_update_end hands out the report once the end-of-bond scalars have landed.  A
// caller may begin bond k+1 before ending bond k (at most two bond updates in flight): the wait of `end` then costs nothing
// because `begin` of the next bond has already passed its own synchronisation point.
int tnml_bond_update_begin(tnml_ctx* c, int b, int ha, const tnml_sweep_params* sp) {
    CollScope coll_(c);
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (ha != 1 && ha != 2) return tnml_fail(c, "tnml_bond_update: half must be 1 or 2");
    if (c->pend_count >= 2) return tnml_fail(c, "tnml_bond_update_begin: two bond updates are in flight, call tnml_bond_update_end first");
    const int slot = (c->pend_tail + c->pend_count) & 1;
    PendingReport& pr = c->pend[slot];
    tnml_bond_report* rep = &pr.rep;
    memset(rep, 0, sizeof *rep);
    pr.b = b; pr.ha = ha; pr.sp = *sp; pr.spec = false; pr.nundo = 0;
    // what the report needs reaches its pinned block through the kernels that compute it (round 5: four copy kernels per bond update less):
    // the CG scalars and trace (k_cg_step2 / k_cg_resid2 of the fp64 literal or merged CG), the norms of the new bond tensor (partial pairs,
    // summed by tnml_bond_update_end), and -- on one rank -- the after-SVD cost partials (k_reduce_partials)
    const bool exact_ = c->single() && c->cg_method == 2;
    const bool fastc_ = c->single() && c->cg_method == 1;
    pr.trace_mirrored = !c->single() && !exact_ && !fastc_ && !sp->report_costs;    // (the per-label variant's entry check writes its flag in k_cg_init2: it keeps the copy)
    pr.carry_direct = !c->multi();
    if (pr.trace_mirrored) { memset(trace_host(c, slot), 0, sizeof(double) * (SC_N + 4 * TNML_MAX_PASS)); c->hmir = trace_host(c, slot); }
    memset(pend_host(c, slot), 0, sizeof(double) * 64);
    struct MirrorOff { tnml_ctx* c; ~MirrorOff() { c->hmir = nullptr; } } mirror_off_{c};
    TCK(tnml_set_bond(c, b));                                         // :488
    if (c->env_budget_bytes > 0 && c->env_async) TCK(env_lookahead(c, b, ha));
    const BondPlan p = c->plan;
    const size_t ne = (size_t)p.mL * 4 * p.mR * p.LB;
    rep->bond = b; rep->half = ha; rep->c = (ha == 1) ? b : b + 1;    // :482
    rep->origm = c->W[b].mr;                                          // :493
    rep->mL = p.mL; rep->mR = p.mR; rep->label_on_B = (p.kind == 2);
    const PackDesc pd = bond_pack_desc(p);
    TCK(launch_bond_form(c, c->W[b], c->W[b + 1], c->tB));            // :494
    bool outputs_current = c->reuse_p && c->p_valid;                  // left by the previous bond update's quadcost
    c->p_valid = false;
    // with carried outputs no label dot rewrites the [cost | ncorrect | pAp] head of the tail before the first all-reduce: the
    // pack kernel clears it on the way
    TCK(launch_pack(c, pd, c->tB, c->vB, nullptr, outputs_current && !sp->report_costs ? c->tail : nullptr, TNML_NSCAL_AR));
    c->tail_zeroed = outputs_current && !sp->report_costs;
    if (sp->report_costs) {                                           // single.h:572,621: norm(oB), quadcost(oB)
        TCK(quadcost_device(c, sp->lambda_cost, &rep->cost_old, nullptr, nullptr, nullptr, true));
        rep->norm_oB = c->last_bnorm;
        outputs_current = c->reuse_p;                                 // that was the forward pass of the first gradient
    }
    const bool exact = c->single() && c->cg_method == 2;              // method = exact (single.h:600)
    if (exact) TCK(exact_device(c, sp->lambda, c->pcut));
    else TCK(cgrad_device(c, sp->npass, sp->lambda, sp->cconv, outputs_current));   // :504
    c->tail_zeroed = false;
    if (sp->report_costs) TCK(quadcost_device(c, sp->lambda_cost, &rep->cost_cg, nullptr, &rep->reg_cost_cg, nullptr, false));   // single.h:622,626
    if (c->carry_slot >= 0) { TCK(allreduce(c, c->tail + TNML_CARRY, TNML_CARRYN)); TCK(carry_deliver(c)); }   // (only when no packed all-reduce ran above: the exact solver)
    TCK(launch_unpack(c, pd, c->vB, c->tB));
    c->hmir = nullptr;
    if (!pr.trace_mirrored) TCK(cgrad_trace_enqueue(c, slot));        // parsed by tnml_bond_update_end
    TCK(svd_split_device(c, c->tB, b, ha, sp->cutoff, sp->maxm, sp->minm, &rep->truncerr, &rep->newm, nullptr, nullptr, slot));   // :519-522 (may run without its host synchronisation: tnml_ctx::spec_split)
    if (c->debug_nudge_rank == c->cfg.rank) TCK(launch_nudge(c, c->W[b].a));
    // replicas: the two site tensors the split just wrote must be bit-identical on every rank.  Their fingerprint goes into the
    // carried slots of the tail as exact integer pieces (mode 1: summed with the next packed all-reduce, checked when the report
    // is handed out -- no collective of its own).  Mode 2 checks at once, BEFORE anything consumes the tensors (bond tensor, P/dP,
    // the shifted environment): on a mismatch rank 0's two tensors replace everybody's, counted.
    pr.fp = c->multi() && c->check_replicas;
    if (pr.fp) {
        TCK(replica_fingerprint(c, b, b + 1, c->tail + TNML_FPSLOT));
        if (c->check_replicas_mode == 2) {
            TCK(allreduce(c, c->tail + TNML_FPSLOT, 8));
            double* hf = pend_host(c, slot) + 48;
            HIPCK(c, hipMemcpyAsync(hf, c->tail + TNML_FPSLOT, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            SYNCK(c, c->stream);
            if (!fingerprint_agrees(hf, c->cfg.nranks)) {             // every rank sees the same sums: every rank takes this branch together
                for (int j = b; j <= b + 1; ++j) { SiteT& sT = c->W[j]; TCK(bcast_rank0(c, sT.a, (size_t)sT.ml * 2 * sT.mr * sT.L)); }
                c->replica_repairs += 1;
            }
            pr.fp = false;                                            // settled
        }
    }
    TCK(launch_bond_form(c, c->W[b], c->W[b + 1], c->tB2));           // :527
    TCK(launch_pack(c, pd, c->tB2, c->vB, nullptr));
    // :532 quadcost(newB); P and dP stay for the next bond update.  Its cost partials land in the CARRIED slots of the tail.
    if (pr.carry_direct) {
        // one rank: nothing on the device consumes these cost partials -- the per-block sums go straight to the pinned report block and the
        // host adds them (no k_reduce_partials launch, no copy)
        double* dev_partials = c->partials;
        c->partials = c->hcost + (size_t)slot * c->partial_cap * 12;
        const int rc_ = forward_pass(c, c->vB, LD_MODE_COST, c->tail + TNML_CARRY, true, false);
        c->partials = dev_partials;
        TCK(rc_);
        pr.cost_rows = c->part_n;
    } else TCK(forward_pass(c, c->vB, LD_MODE_COST, c->tail + TNML_CARRY, true));
    TCK(launch_diffnorm_host(c, c->tB2, c->tB, ne, dn_host(c, slot), 256));   // |newB|^2 (slot 12 of quadcost, :528) and |newB - B|^2 (:530) as partial pairs
    pr.dn_pairs = c->last_dn_pairs;
    const bool multi = c->multi();
    if (multi && c->defer_tail) c->carry_slot = slot;                 // summed by the next packed all-reduce (tnml_bond_update_end flushes otherwise)
    else {
        c->carry_slot = slot;
        if (multi) TCK(allreduce(c, c->tail + TNML_CARRY, TNML_CARRYN));
        TCK(carry_deliver(c));
    }
    HIPCK(c, hipEventRecord(pr.ev, c->stream));
    TCK(tnml_shift_env(c, b, ha == 1));                               // :540
    pr.lambda_cost = sp->lambda_cost;
    c->pend_count += 1;
    c->p_valid = true;                                                // in stream order: P/dP of the after-SVD quadcost
    return 0;
}
// a speculative split whose deferred check failed: the site tensors it replaced come back, the buffers it wrote return to the pool
static void spec_rollback(tnml_ctx* c, PendingReport& pr) {
    for (int u = pr.nundo - 1; u >= 0; --u) {
        SiteT& S = c->W[pr.undo[u].j];
        ((pr.undo[u].j == c->c0) ? c->spare_big : c->spare_small).push_back(S.a);
        S.a = pr.undo[u].old; S.ml = pr.undo[u].ml; S.mr = pr.undo[u].mr;
    }
    pr.nundo = 0; pr.spec = false;
}
// ... verified: the replaced buffers are free again
static void spec_commit(tnml_ctx* c, PendingReport& pr) {
    for (int u = 0; u < pr.nundo; ++u) ((pr.undo[u].j == c->c0) ? c->spare_big : c->spare_small).push_back(pr.undo[u].old);
    pr.nundo = 0; pr.spec = false;
}
int tnml_bond_update_end(tnml_ctx* c, tnml_bond_report* rep) {
    CollScope coll_(c);
    if (c->pend_count < 1) return tnml_fail(c, "tnml_bond_update_end: no bond update in flight");
    const int slot = c->pend_tail;
    PendingReport& pr = c->pend[slot];
    if (c->carry_slot == slot) {                                      // nothing followed that would have carried them: one small all-reduce
        TCK(allreduce(c, c->tail + TNML_CARRY, TNML_CARRYN));
        TCK(carry_deliver(c));
    }
    HIPCK(c, hipEventSynchronize(pr.ev));
    HIPCK(c, hipEventSynchronize(pr.ev2));
    // a collective of this bond update that gave up waiting for a peer left its buffer unsummed: say so before anything below reads
    // the sums (it would show up as a failed replica check or a failed split check otherwise)
    TCK(ipc_comm_check(c));
    double* hq = pend_host(c, slot);
    if (pr.spec) {
        // the deferred check of the speculative split: its verdict came with the carried slots (summed over the ranks: every rank sees the same number)
        const double* hm = c->hrep + (size_t)slot * c->hrep_stride;
        const int n = pr.split_n;
        if ((c->multi() ? hq[TNML_SPECSLOT] : hm[n + 4]) != 0.) {    // (one rank: straight from the mirror of the check values)
            // dependent vectors even after re-orthonormalisation (or the test hook): everything this bond update and the one begun after
            // it wrote is dropped -- site tensors back from their spare buffers -- and both run again, this one with the synchronous split
            // and its rocSOLVER fallback.  Rare (a few per sweep), so the repeat may cost what it costs.
            const bool had_next = c->pend_count == 2;
            PendingReport& nx = c->pend[slot ^ 1];
            SYNCK(c, c->stream);
            if (c->copy_stream) HIPCK(c, hipStreamSynchronize(c->copy_stream));
            const int b1 = nx.b, ha1 = nx.ha; const tnml_sweep_params sp1 = nx.sp;
            const int b0 = pr.b, ha0 = pr.ha; const tnml_sweep_params sp0 = pr.sp;
            if (had_next) { if (nx.nundo == 2) spec_rollback(c, nx); else return tnml_fail(c, "bond %d: cannot repeat after a failed split check (the next bond update kept no undo record)", b0); }
            spec_rollback(c, pr);
            c->pend_count = 0; c->carry_slot = -1; c->p_valid = false; c->currb = -1;
            HIPCK(c, hipMemsetAsync(c->tail + TNML_CARRY, 0, sizeof(double) * TNML_CARRYN, c->stream));
            c->spec_redos += 1; c->svd_fallbacks += 1;
            // what a roll-back costs = the device time of the work enqueued again (tnml_split_stats reports count and sum)
            hipEvent_t re0 = nullptr, re1 = nullptr;
            if (hipEventCreate(&re0) == hipSuccess && hipEventCreate(&re1) == hipSuccess) (void)hipEventRecord(re0, c->stream);
            c->force_safe = true;
            int rc = tnml_bond_update_begin(c, b0, ha0, &sp0);        // lands in `slot` again (pend_tail has not moved)
            c->force_safe = false;
            if (rc) return rc;
            if (had_next) TCK(tnml_bond_update_begin(c, b1, ha1, &sp1));
            if (re0 && re1) { (void)hipEventRecord(re1, c->stream); c->redo_events.push_back({re0, re1}); }
            return tnml_bond_update_end(c, rep);
        }
        spec_commit(c, pr);
        // what the synchronous form does right after its host round trip: truncation error from the eigenvalues, statistics
        std::vector<double> p(n);
        for (int g = 0; g < n; ++g) { double lam = hm[n - 1 - g]; if (!(lam > 0.)) lam = 0.; p[g] = lam; }
        double te = 0.;
        const int mx = pr.sp.maxm < c->maxm ? pr.sp.maxm : c->maxm;
        const int m = tnml_truncate(p.data(), n, mx, pr.sp.minm < mx ? pr.sp.minm : mx, pr.sp.cutoff, &te);
        if (m != pr.rep.newm) return tnml_fail(c, "bond %d: speculative split kept %d columns, the truncation rule says %d", pr.rep.bond, pr.rep.newm, m);
        pr.rep.truncerr = te;
        c->svd_last_dev0 = hm[n]; c->svd_last_dev1 = 0.75 * hm[n] * hm[n];
        if (hm[n + 2] != 0.) c->svd_cholqr += 1;
    } else spec_commit(c, pr);                                        // synchronous split: verified when it ran
    c->pend_tail ^= 1; c->pend_count -= 1;
    const bool exact = c->single() && c->cg_method == 2;
    cgrad_trace_parse(c, pr.sp.npass, &pr.rep.cg, slot);
    if (exact) memset(&pr.rep.cg, 0, sizeof pr.rep.cg);              // no CG ran
    if (pr.fp && !fingerprint_agrees(hq + TNML_FPSLOT, c->cfg.nranks))   // every rank sees the same sums
        return tnml_fail(c, "bond %d: replicas of W.A(%d), W.A(%d) differ between ranks after the split", pr.rep.bond, pr.rep.bond, pr.rep.bond + 1);
    double t[13];
    if (pr.carry_direct) {
        const double* hp = c->hcost + (size_t)slot * c->partial_cap * 12;
        for (int l = 0; l < 12; ++l) { double a = 0.; for (int r = 0; r < pr.cost_rows; ++r) a += hp[(size_t)r * 12 + l]; t[l] = a; }
    } else for (int l = 0; l < 12; ++l) t[l] = hq[TNML_CARRY + l];
    double nb2 = 0., df2 = 0.;                                        // the partial pairs of k_diffnorm1, in workgroup order
    { const double* dp = dn_host(c, slot); for (int k = 0; k < pr.dn_pairs; ++k) { nb2 += dp[2 * k]; df2 += dp[2 * k + 1]; } }
    t[12] = nb2;
    quadcost_parse(c, t, pr.lambda_cost, &pr.rep.cost_after_svd, pr.rep.label_cost, &pr.rep.reg_cost, &pr.rep.ncorrect);
    pr.rep.norm_newB = std::sqrt(nb2); pr.rep.diff_B_newB = std::sqrt(df2);
    if (rep) *rep = pr.rep;
    return 0;
}
int tnml_bond_update(tnml_ctx* c, int b, int ha, const tnml_sweep_params* sp, tnml_bond_report* rep) {
    if (c->pend_count != 0) return tnml_fail(c, "tnml_bond_update: a pipelined bond update is still in flight");
    TCK(tnml_bond_update_begin(c, b, ha, sp));
    return tnml_bond_update_end(c, rep);
}
