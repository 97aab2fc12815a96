// local_comm.hip -- in-process communicators: (a) for ranks that share ONE device, (b) the one-shot peer-write all-reduce for ranks
// of one process on SEVERAL devices (tnml_comm_init_oneshot).
//
// (b) The five 461 KB all-reduces of a bond update are latency bound on a ring (2 (n-1) hops).  One shot instead: every rank WRITES its
// [tail | G] buffer straight into its own slot of every peer's receive region (one kernel, peer stores over the direct xGMI links, all
// seven links at once, one hop), the ranks' streams meet through HIP events, and every rank sums the n slots of its OWN receive region
// in rank order -- local reads, bit-identical sums on every rank (what the replicated CG scalars and the split rely on).  Receive
// regions have two parities; a slot is rewritten only after its reader's event of two collectives ago.  With all ranks on one device
// the same code runs with plain device pointers: that is how it is tested on a one-GPU box (tests/test_multirank_one_gpu.py).
//
// (a)
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
    int oneshot = 0;                      // (b): per-rank receive regions, peer writes
    size_t cap = 0;                       // doubles per rank slot
    double* staging[2] = {nullptr, nullptr};
    std::vector<double*> recv[2];         // (b) recv[p][j]: receive region of rank j ([n][cap] doubles, on rank j's device)
    std::vector<int> devs;
    std::vector<hipEvent_t> written[2], read_done[2];
    std::vector<char> read_rec[2];        // read_done[p][r] has been recorded at least once
    std::vector<long> gen;                // per rank: collectives entered so far
    int refs = 0;
    // host barrier (generation counting)
    std::mutex mu; std::condition_variable cv; int waiting = 0; long bgen = 0;
    int timeout_s = 120;                  // option comm_timeout_s (the longest wait of a rank for its peers)
    bool aborted = false;                 // a rank failed (or never arrived): every later collective fails at once instead of hanging
    bool barrier() {                      // false: the communicator is aborted
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const long g = bgen;
        if (++waiting == n) { waiting = 0; ++bgen; cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return bgen != g || aborted; });
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
// (b) this rank's buffer -> its slot in the receive region of every rank (blockIdx.y = destination rank; peer stores)
struct LcPeers { double* dst[16]; };
__global__ void k_lc_push(const double* __restrict__ buf, size_t count, LcPeers P, size_t slot_off) {
    double* d = P.dst[blockIdx.y] + slot_off;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) d[i] = buf[i];
}
__global__ void k_lc_copy(const double* __restrict__ st, size_t count, double* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) out[i] = st[i];
}

static int comm_init_inproc(tnml_ctx** ctxs, int n, int oneshot) {
    const char* who = oneshot ? "tnml_comm_init_oneshot" : "tnml_comm_init_local";
    if (!ctxs || n < 1 || n > 16) return tnml_fail(nullptr, "%s: bad arguments (1..16 ranks)", who);
    size_t cap = 0;
    for (int r = 0; r < n; ++r) {
        tnml_ctx* c = ctxs[r];
        if (!c) return tnml_fail(nullptr, "%s: null context", who);
        if (c->cfg.nranks != n || c->cfg.rank != r) return tnml_fail(c, "%s: context %d was created as rank %d of %d", who, r, c->cfg.rank, c->cfg.nranks);
        if (!oneshot && c->cfg.device != ctxs[0]->cfg.device) return tnml_fail(c, "tnml_comm_init_local: ranks on different devices use tnml_comm_init_oneshot or RCCL (tnml_comm_init)");
        if (c->multi()) return tnml_fail(c, "%s: context already has a communicator", who);
        if (c->mcap + TNML_TAILN > cap) cap = c->mcap + TNML_TAILN;
        if (c->mcap != ctxs[0]->mcap) return tnml_fail(c, "%s: contexts must share maxm", who);
    }
    LocalComm* lc = new LocalComm();
    lc->n = n; lc->device = ctxs[0]->cfg.device; lc->cap = cap; lc->refs = n; lc->oneshot = oneshot;
    for (int r = 0; r < n; ++r) lc->timeout_s = std::min(lc->timeout_s, ctxs[r]->comm_timeout_s);
    lc->devs.resize(n);
    for (int r = 0; r < n; ++r) lc->devs[r] = ctxs[r]->cfg.device;
    auto fail = [&](tnml_ctx* c, const char* what) { for (int p = 0; p < 2; ++p) { if (lc->staging[p]) (void)hipFree(lc->staging[p]); for (double* q : lc->recv[p]) if (q) (void)hipFree(q); } delete lc; return tnml_fail(c, "%s: %s", who, what); };
    if (oneshot) {
        // every pair of distinct devices must be able to store into each other's memory (xGMI peer access)
        for (int a = 0; a < n; ++a)
            for (int b2 = 0; b2 < n; ++b2) {
                if (lc->devs[a] == lc->devs[b2]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, lc->devs[a], lc->devs[b2]) != hipSuccess || !can) return fail(ctxs[a], "no peer access between the devices of two ranks");
                if (hipSetDevice(lc->devs[a]) != hipSuccess) return fail(ctxs[a], "hipSetDevice failed");
                const hipError_t e = hipDeviceEnablePeerAccess(lc->devs[b2], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(ctxs[a], "hipDeviceEnablePeerAccess failed");
                (void)hipGetLastError();
            }
        for (int p = 0; p < 2; ++p) {
            lc->recv[p].assign(n, nullptr);
            for (int r = 0; r < n; ++r) {
                if (hipSetDevice(lc->devs[r]) != hipSuccess || hipMalloc((void**)&lc->recv[p][r], sizeof(double) * cap * n) != hipSuccess) return fail(ctxs[r], "hipMalloc of a receive region failed");
            }
        }
    } else {
        if (hipSetDevice(lc->device) != hipSuccess) return fail(ctxs[0], "hipSetDevice failed");
        for (int p = 0; p < 2; ++p)
            if (hipMalloc((void**)&lc->staging[p], sizeof(double) * cap * n) != hipSuccess) return fail(ctxs[0], "hipMalloc failed");
    }
    for (int p = 0; p < 2; ++p) {
        lc->written[p].resize(n); lc->read_done[p].resize(n); lc->read_rec[p].assign(n, 0);
        for (int r = 0; r < n; ++r) {
            (void)hipSetDevice(lc->devs[r]);
            (void)hipEventCreateWithFlags(&lc->written[p][r], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&lc->read_done[p][r], hipEventDisableTiming);
        }
    }
    lc->gen.assign(n, 0);
    for (int r = 0; r < n; ++r) ctxs[r]->local = lc;
    (void)hipSetDevice(ctxs[0]->cfg.device);             // (the calling thread -- rank 0's -- gets its own device back; every entry point sets it anyway)
    return 0;
}
int tnml_comm_init_local(tnml_ctx** ctxs, int n) { return comm_init_inproc(ctxs, n, 0); }
int tnml_comm_init_oneshot(tnml_ctx** ctxs, int n) { return comm_init_inproc(ctxs, n, 1); }
int local_comm_mode(const tnml_ctx* c) { return c->local ? (c->local->oneshot ? 3 : 2) : 0; }
void local_comm_release(tnml_ctx* c) {
    LocalComm* lc = c->local;
    if (!lc) return;
    c->local = nullptr;
    bool last;
    { std::lock_guard<std::mutex> lk(lc->mu); last = --lc->refs == 0; }
    if (!last) return;
    for (int p = 0; p < 2; ++p) {
        if (lc->staging[p]) (void)hipFree(lc->staging[p]);
        for (double* q : lc->recv[p]) if (q) (void)hipFree(q);
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
    const int nb = (int)((count + 255) / 256 > 1024 ? 1024 : (count + 255) / 256);
    if (lc->oneshot) {
        // one shot: this rank's values into its slot of every rank's receive region (a broadcast: rank 0 alone writes)
        if (op != 1 || r == 0) {
            LcPeers P;
            for (int j = 0; j < n; ++j) P.dst[j] = lc->recv[p][j];
            hipLaunchKernelGGL(k_lc_push, dim3(nb > 256 ? 256 : nb, n), dim3(256), 0, st, (const double*)buf, count, P, (size_t)r * lc->cap);
            LCK(c, lc, hipGetLastError());
        }
    } else if (op != 1 || r == 0) LCK(c, lc, hipMemcpyAsync(lc->staging[p] + (size_t)r * lc->cap, buf, sizeof(double) * count, hipMemcpyDeviceToDevice, st));
    LCK(c, lc, hipEventRecord(lc->written[p][r], st));
    if (!lc->barrier()) return tnml_fail(c, "local communicator: a rank left the collective (aborted)");   // every rank has recorded its `written` event
    for (int j = 0; j < n; ++j) if (j != r) LCK(c, lc, hipStreamWaitEvent(st, lc->written[p][j], 0));
    const double* src = lc->oneshot ? lc->recv[p][r] : lc->staging[p];                              // (one shot: the n slots of the OWN receive region)
    if (op == 0)      hipLaunchKernelGGL(k_lc_sum, dim3(nb), dim3(256), 0, st, src, n, lc->cap, count, buf);
    else              hipLaunchKernelGGL(k_lc_copy, dim3(nb), dim3(256), 0, st, src, count, buf);
    LCK(c, lc, hipGetLastError());
    LCK(c, lc, hipEventRecord(lc->read_done[p][r], st));
    if (!lc->barrier()) return tnml_fail(c, "local communicator: a rank left the collective (aborted)");   // ... and its `read_done` event, before anyone re-uses the parity
    lc->read_rec[p][r] = 1;
    return 0;
}
// a rank that fails outside a collective (tnml_fail in a bond update) tells its peers, so that they do not wait for it
void local_comm_abort(tnml_ctx* c) { if (c->local) c->local->abort(); }
void local_comm_set_timeout(tnml_ctx* c, int seconds) { if (c->local) { std::lock_guard<std::mutex> lk(c->local->mu); c->local->timeout_s = seconds; } }
