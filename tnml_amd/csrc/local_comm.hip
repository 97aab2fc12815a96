// local_comm.hip -- an in-process communicator for ranks that share ONE device.
//
// RCCL refuses two ranks on the same GPU, so on a one-GPU box the multi-rank logic of the library (image shards,
// packed [G | cost | ncorrect | pAp] all-reduce, collective truncation decision, replica fingerprints) could only ever
// run with a 1-rank communicator.  This communicator gives every rank of one process its own context and stream on the
// same device and implements the two collectives the library uses -- sum all-reduce of fp64, broadcast from rank 0 --
// through a staging buffer in device memory, ordered by HIP events between the ranks'
// streams and a host barrier between their threads (one host thread per rank, as in the C++ fixedL driver).  The sum
// runs over the ranks in rank order on every rank: bit-identical results everywhere, like a ring all-reduce.
//
// It is a correctness vehicle (tests, `ngpu` > visible devices in the drivers), not a performance path: ranks on
// different devices use RCCL over xGMI (tnml_comm_init).
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "tnml_internal.h"

struct LocalComm {
    int n = 0;
    int device = 0;
    size_t cap = 0;                       // doubles per rank slot
    double* staging[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> written[2], read_done[2];
    std::vector<char> read_rec[2];        // read_done[p][r] has been recorded at least once
    std::vector<long> gen;                // per rank: collectives entered so far
    int refs = 0;
    // host barrier (generation counting)
    std::mutex mu; std::condition_variable cv; int waiting = 0; long bgen = 0;
    bool aborted = false;                 // a rank failed (or never arrived): every later collective fails at once instead of hanging
    bool barrier() {                      // false: the communicator is aborted
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const long g = bgen;
        if (++waiting == n) { waiting = 0; ++bgen; cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(120), [&] { return bgen != g || aborted; });
        if (!ok || aborted) { aborted = true; cv.notify_all(); return false; }     // a peer left the protocol (error return on its side, or it never entered)
        return true;
    }
    void abort() { std::lock_guard<std::mutex> lk(mu); aborted = true; cv.notify_all(); }
};
#define LCK(c, lc, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (lc)->abort(); return tnml_fail((c), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

__global__ void k_lc_sum(const double* __restrict__ st, int n, size_t cap, size_t count, double* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double s = st[i];
        for (int r = 1; r < n; ++r) s += st[(size_t)r * cap + i];
        out[i] = s;
    }
}
__global__ void k_lc_copy(const double* __restrict__ st, size_t count, double* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) out[i] = st[i];
}

int tnml_comm_init_local(tnml_ctx** ctxs, int n) {
    if (!ctxs || n < 1) return tnml_fail(nullptr, "tnml_comm_init_local: bad arguments");
    size_t cap = 0;
    for (int r = 0; r < n; ++r) {
        tnml_ctx* c = ctxs[r];
        if (!c) return tnml_fail(nullptr, "tnml_comm_init_local: null context");
        if (c->cfg.nranks != n || c->cfg.rank != r) return tnml_fail(c, "tnml_comm_init_local: context %d was created as rank %d of %d", r, c->cfg.rank, c->cfg.nranks);
        if (c->cfg.device != ctxs[0]->cfg.device) return tnml_fail(c, "tnml_comm_init_local: ranks on different devices use RCCL (tnml_comm_init)");
        if (c->comm || c->local) return tnml_fail(c, "tnml_comm_init_local: context already has a communicator");
        if (c->mcap + TNML_TAILN > cap) cap = c->mcap + TNML_TAILN;
        if (c->mcap != ctxs[0]->mcap) return tnml_fail(c, "tnml_comm_init_local: contexts must share maxm");
    }
    LocalComm* lc = new LocalComm();
    lc->n = n; lc->device = ctxs[0]->cfg.device; lc->cap = cap; lc->refs = n;
    if (hipSetDevice(lc->device) != hipSuccess) { delete lc; return tnml_fail(ctxs[0], "hipSetDevice failed"); }
    for (int p = 0; p < 2; ++p) {
        if (hipMalloc((void**)&lc->staging[p], sizeof(double) * cap * n) != hipSuccess) { delete lc; return tnml_fail(ctxs[0], "tnml_comm_init_local: hipMalloc failed"); }
        lc->written[p].resize(n); lc->read_done[p].resize(n); lc->read_rec[p].assign(n, 0);
        for (int r = 0; r < n; ++r) {
            (void)hipEventCreateWithFlags(&lc->written[p][r], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&lc->read_done[p][r], hipEventDisableTiming);
        }
    }
    lc->gen.assign(n, 0);
    for (int r = 0; r < n; ++r) ctxs[r]->local = lc;
    return 0;
}
void local_comm_release(tnml_ctx* c) {
    LocalComm* lc = c->local;
    if (!lc) return;
    c->local = nullptr;
    bool last;
    { std::lock_guard<std::mutex> lk(lc->mu); last = --lc->refs == 0; }
    if (!last) return;
    for (int p = 0; p < 2; ++p) {
        if (lc->staging[p]) (void)hipFree(lc->staging[p]);
        for (auto e : lc->written[p]) (void)hipEventDestroy(e);
        for (auto e : lc->read_done[p]) (void)hipEventDestroy(e);
    }
    delete lc;
}
int local_comm_size(const tnml_ctx* c) { return c->local ? c->local->n : 0; }

// op: 0 = sum of doubles, 1 = copy of rank 0's values
int local_comm_exchange(tnml_ctx* c, double* buf, size_t count, int op) {
    LocalComm* lc = c->local;
    if (count > lc->cap) { lc->abort(); return tnml_fail(c, "local communicator: %zu elements exceed the staging capacity %zu", count, lc->cap); }
    const int r = c->cfg.rank, n = lc->n;
    const int p = (int)(lc->gen[r]++ & 1);
    hipStream_t st = c->stream;
    // the slot of this parity was read two collectives ago: wait for those readers
    for (int j = 0; j < n; ++j) if (lc->read_rec[p][j]) LCK(c, lc, hipStreamWaitEvent(st, lc->read_done[p][j], 0));
    if (op != 1 || r == 0) LCK(c, lc, hipMemcpyAsync(lc->staging[p] + (size_t)r * lc->cap, buf, sizeof(double) * count, hipMemcpyDeviceToDevice, st));
    LCK(c, lc, hipEventRecord(lc->written[p][r], st));
    if (!lc->barrier()) return tnml_fail(c, "local communicator: a rank left the collective (aborted)");   // every rank has recorded its `written` event
    for (int j = 0; j < n; ++j) if (j != r) LCK(c, lc, hipStreamWaitEvent(st, lc->written[p][j], 0));
    const int nb = (int)((count + 255) / 256 > 1024 ? 1024 : (count + 255) / 256);
    if (op == 0)      hipLaunchKernelGGL(k_lc_sum, dim3(nb), dim3(256), 0, st, (const double*)lc->staging[p], n, lc->cap, count, buf);
    else              hipLaunchKernelGGL(k_lc_copy, dim3(nb), dim3(256), 0, st, (const double*)lc->staging[p], count, buf);
    LCK(c, lc, hipGetLastError());
    LCK(c, lc, hipEventRecord(lc->read_done[p][r], st));
    if (!lc->barrier()) return tnml_fail(c, "local communicator: a rank left the collective (aborted)");   // ... and its `read_done` event, before anyone re-uses the parity
    lc->read_rec[p][r] = 1;
    return 0;
}
// a rank that fails outside a collective (tnml_fail in a bond update) tells its peers, so that they do not wait for it
void local_comm_abort(tnml_ctx* c) { if (c->local) c->local->abort(); }
