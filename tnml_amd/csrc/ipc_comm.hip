// ipc_comm.hip -- the one-shot all-reduce ACROSS PROCESSES (one process per GPU, as `python bench.py --gpus N` and every
// torch.distributed launch run): the sums of fixedL.cc:385,402,421,427 over the image shards of the ranks.
//
// A bond update enters five sum all-reduces of the packed [48 scalars | G] buffer (461 KB at m = 120): latency bound on a ring
// (2 (R - 1) hops over point-to-point xGMI).  One shot instead, and nothing on the host between the ranks:
//   * every rank owns a RECEIVE REGION in its HBM -- [2 parities][R slots][cap] doubles plus arrival flags -- allocated fine-grained
//     and exported through hipIpcGetMemHandle; the handles travel once over the caller's control plane (gloo, MPI, a file: any
//     channel), every rank maps its peers' regions (tnml_oneshot_export / tnml_oneshot_connect);
//   * ONE kernel per collective and rank (k_os_exchange), workgroup = one 2048-element chunk of the buffer: it stores its chunk into
//     its slot of EVERY rank's region (system-scope stores: all links at once, one hop), fences, raises the chunk's arrival flag
//     on every rank to the collective's sequence number, then polls its OWN flags of that chunk until every rank's store has landed
//     and sums the R slots in rank order -- local reads, the same bits on every rank (what the replicated CG scalars and the split
//     rely on), no host barrier, no event: tnml_bond_update_begin never blocks.  A broadcast is the same kernel with rank 0 the only
//     writer of data; every rank still raises and awaits all flags, so that a broadcast is a rendezvous like a sum (the argument below needs it).
//   * two parities: rank j can store collective s + 2 only after it finished s + 1, which needs this rank's stores of s + 1, which
//     this rank's stream orders after its own kernel of s -- so a slot is never overwritten while its owner still reads it, without
//     acknowledgements.  A poll that sees nothing for `comm_timeout_s` sets a status word and leaves (tnml_synchronize and the next
//     collective report it): a peer that died costs a time-out, not a hung GPU.
// On one GPU the same code runs between two PROCESSES sharing the device (IPC handles open on the exporting device too): that is how
// it is tested on a one-GPU box (tests/test_multirank_one_gpu.py).  Ranks that are threads of one process keep local_comm.hip.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "tnml_internal.h"

#define OS_CHUNK 2048
#define OS_MAXR 16

struct IpcComm {
    int n = 0, r = 0;
    size_t cap = 0;                       // doubles per slot
    int nbmax = 0;                        // chunks per slot
    char* region = nullptr;               // this rank's region: [flags n * nbmax u64 | status 8 u64 | recv 2 * n * cap doubles]
    size_t region_bytes = 0, recv_off = 0;
    char* peer[OS_MAXR] = {nullptr};      // every rank's region as mapped here (peer[r] = region)
    bool opened[OS_MAXR] = {false};
    unsigned long long seq = 0;           // collectives entered so far
    int mem_kind = 0;                     // 1 fine-grained, 2 uncached (coarse-grained memory is refused)
    unsigned long long* h_status = nullptr;   // pinned: the kernels' time-out word, mirrored
};

struct OsArgs {
    double* buf; size_t count; unsigned long long seq; int n, r, op; size_t cap; int nbmax; int parity;
    char* peer[OS_MAXR]; size_t recv_off;
    unsigned long long* h_status; long long timeout_ticks;
};
static __device__ __forceinline__ void st_sys(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
static __device__ __forceinline__ double ld_sys(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// At most OS_MAXWG workgroups per collective, each walking the chunks blockIdx.x, blockIdx.x + gridDim.x, ... in the same order on every
// rank: first it hands out ALL its chunks (stores, fence, flags -- nothing to wait for), then it collects them (poll, ordered sum).
// Why the cap: a workgroup that polls holds a wave on every SIMD of its CU.  The 461 KB buffer of an ordinary bond is 29 chunks, but the
// two Label-on-B bonds move 4.6 MB = 282 chunks, and 282 polling workgroups put a wave on every SIMD of EVERY CU; where ranks share a GPU
// (the one-GPU test vehicle) a 768-lane workgroup of the peer's preceding kernel -- 3 waves of 168 registers per SIMD -- then fits
// nowhere, the peer never reaches its collective and both sides wait for ever (seen as "collective 30 timed out", the second all-reduce
// of the first Label-on-B bond: profiles/r06_oneshot_processes_root_cause.txt).  32 workgroups keep >= 7/8 of the chip free whatever
// the payload, and still move 4.6 MB in tens of microseconds.
#define OS_MAXWG 32
__global__ __launch_bounds__(256) void k_os_exchange(OsArgs A) {
    __shared__ int s_bad;
    const int tid = threadIdx.x, n = A.n, r = A.r;
    const int nb = (int)((A.count + OS_CHUNK - 1) / OS_CHUNK);
    if (tid == 0) s_bad = 0;
    // ---- 1. this rank's chunks into its slot of every rank's region (a broadcast: rank 0 alone writes), then the chunk's flag on every rank
    for (int chunk = blockIdx.x; chunk < nb; chunk += gridDim.x) {
        const size_t lo = (size_t)chunk * OS_CHUNK, hi = lo + OS_CHUNK < A.count ? lo + OS_CHUNK : A.count;
        if (A.op == 0 || r == 0) {
            for (int j = 0; j < n; ++j) {
                double* dst = reinterpret_cast<double*>(A.peer[j] + A.recv_off) + ((size_t)A.parity * n + r) * A.cap;
                for (size_t i = lo + tid; i < hi; i += 256) st_sys(dst + i, A.buf[i]);
            }
            __threadfence_system();                              // this lane's stores are visible system-wide ...
        }
        __syncthreads();                                         // ... and so are the whole workgroup's
        // (a broadcast posts the flags of EVERY rank too and waits for all of them: a root that waited for nobody could run two collectives
        // ahead and store into the slot -- same parity -- a slow rank is still reading; the flags make every collective a rendezvous)
        if (tid < n) {
            unsigned long long* fl = reinterpret_cast<unsigned long long*>(A.peer[tid]) + (size_t)r * A.nbmax + chunk;
            __hip_atomic_store(fl, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- 2. every rank's chunk (a broadcast: every rank's arrival) has landed here: ordered local sum (rank order: the same bits on
    //         every rank) / copy of rank 0's values
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    for (int chunk = blockIdx.x; chunk < nb; chunk += gridDim.x) {
        const size_t lo = (size_t)chunk * OS_CHUNK, hi = lo + OS_CHUNK < A.count ? lo + OS_CHUNK : A.count;
        if (tid < n && !s_bad) {
            const unsigned long long* fl = reinterpret_cast<const unsigned long long*>(A.peer[r]) + (size_t)tid * A.nbmax + chunk;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < A.seq) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > A.timeout_ticks) { s_bad = 1; break; }
            }
        }
        __syncthreads();
        if (s_bad) {
            // a peer never arrived: report, and POISON this chunk (and, without waiting again, every later one of this workgroup) -- the
            // stream carries on (pack kernels, CG vector kernels, the split), and whatever consumes the buffer before the host looks at the
            // status word must not see a plausible unsummed value: NaNs spread into every cost, norm and fingerprint downstream, and
            // every checked host synchronisation (SYNCK) fails on the status word
            if (tid == 0) { __hip_atomic_store(A.h_status, A.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            for (size_t i = lo + tid; i < hi; i += 256) A.buf[i] = qnan;
            continue;                                            // (s_bad stays set: uniform over the workgroup from here on)
        }
        __threadfence_system();                                  // (acquire for the lanes that did not poll)
        const double* src = reinterpret_cast<const double*>(A.peer[r] + A.recv_off) + (size_t)A.parity * n * A.cap;
        for (size_t i = lo + tid; i < hi; i += 256) {
            double s = ld_sys(src + i);
            if (A.op == 0) for (int j = 1; j < n; ++j) s += ld_sys(src + (size_t)j * A.cap + i);
            A.buf[i] = s;
        }
        __syncthreads();                                         // (s_bad is read above by every lane before any lane of the next round may set it)
    }
}

static size_t os_region_bytes(int n, size_t cap, int nbmax, size_t* recv_off) {
    size_t off = sizeof(unsigned long long) * ((size_t)n * nbmax + 8);
    off = (off + 4095) / 4096 * 4096;
    *recv_off = off;
    return off + sizeof(double) * 2 * (size_t)n * cap;
}

// size of the receive region tnml_oneshot_export allocates for a context of this configuration: [2 parities][nranks slots][mcap + tail]
// doubles + flags -- 37 MB per rank at maxm = 120 with 8 ranks, 460 MB at maxm = 300; NOT part of tnml_estimate_bytes (which does not know the
// transport): a driver that plans maxm against free memory (tnml_plan_maxm) subtracts it from its budget
int64_t tnml_oneshot_region_bytes(const tnml_config* cfg) {
    if (!cfg || cfg->nranks < 1 || cfg->maxm < 1) return -1;
    const bool bf = cfg->dtype == TNML_BF16 || cfg->dtype == TNML_BF16X3;
    const size_t Kmax = bf ? (size_t)(2 * cfg->maxm + 31) / 32 * 32 : (size_t)(2 * cfg->maxm + 15) / 16 * 16;      // (as tnml_create pads it)
    const size_t cap = (size_t)TNML_NL * Kmax * Kmax + TNML_TAILN;
    size_t off;
    return (int64_t)os_region_bytes(cfg->nranks, cap, (int)((cap + OS_CHUNK - 1) / OS_CHUNK), &off);
}

int tnml_oneshot_export(tnml_ctx* c, void* handle64) {
    if (!c || !handle64) return tnml_fail(c, "tnml_oneshot_export: null argument");
    if (c->comm || c->local || c->ipc) return tnml_fail(c, "tnml_oneshot_export: context already has a communicator");
    if (c->cfg.nranks < 1 || c->cfg.nranks > OS_MAXR) return tnml_fail(c, "tnml_oneshot_export: 1..%d ranks", OS_MAXR);
    static_assert(sizeof(hipIpcMemHandle_t) == TNML_ONESHOT_HANDLE_BYTES, "IPC handle size");
    HIPCK(c, hipSetDevice(c->cfg.device));
    IpcComm* ic = new IpcComm();
    ic->n = c->cfg.nranks; ic->r = c->cfg.rank;
    ic->cap = c->mcap + TNML_TAILN;
    ic->nbmax = (int)((ic->cap + OS_CHUNK - 1) / OS_CHUNK);
    ic->region_bytes = os_region_bytes(ic->n, ic->cap, ic->nbmax, &ic->recv_off);
    // fine-grained memory: peer stores over xGMI must become visible to this GPU's loads without a kernel boundary (coarse-grained
    // hipMalloc memory is only coherent at kernel boundaries); RCCL allocates its communication buffers the same way
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < ic->region_bytes) {
            const size_t want = ic->region_bytes;
            delete ic;
            return tnml_fail(c, "tnml_oneshot_export: the receive region needs %zu bytes, %zu are free on device %d -- tnml_estimate_bytes / tnml_plan_maxm do not "
                                "count it: plan maxm with tnml_oneshot_region_bytes subtracted from the budget", want, fr, c->cfg.device);
        }
    }
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, ic->region_bytes, hipDeviceMallocFinegrained) == hipSuccess) ic->mem_kind = 1;
    else if ((void)hipGetLastError(), hipExtMallocWithFlags(&p, ic->region_bytes, hipDeviceMallocUncached) == hipSuccess) ic->mem_kind = 2;
    else {
        // (no plain hipMalloc fallback: coarse-grained memory is coherent at kernel boundaries only -- a peer's stores and flags
        // could sit unseen in this GPU's L2 while the polling kernel runs)
        (void)hipGetLastError();
        const size_t want = ic->region_bytes;
        delete ic;
        return tnml_fail(c, "tnml_oneshot_export: cannot allocate %zu bytes of fine-grained (or uncached) device memory for the receive region; "
                            "the one-shot transport does not run on coarse-grained memory -- use tnml_comm_init (RCCL)", want);
    }
    ic->region = static_cast<char*>(p);
    if (hipMemset(p, 0, ic->recv_off) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); delete ic; return tnml_fail(c, "tnml_oneshot_export: memset failed"); }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(p); delete ic;
        return tnml_fail(c, "tnml_oneshot_export: hipIpcGetMemHandle failed (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    }
    if (hipHostMalloc((void**)&ic->h_status, 64) != hipSuccess) { (void)hipFree(p); delete ic; return tnml_fail(c, "tnml_oneshot_export: hipHostMalloc failed"); }
    ic->h_status[0] = 0;
    memcpy(handle64, &h, sizeof h);
    c->ipc = ic;
    c->bytes += (int64_t)ic->region_bytes;
    return 0;
}

int tnml_oneshot_connect(tnml_ctx* c, const void* handles) {
    if (!c || !handles) return tnml_fail(c, "tnml_oneshot_connect: null argument");
    IpcComm* ic = c->ipc;
    if (!ic) return tnml_fail(c, "tnml_oneshot_connect: call tnml_oneshot_export first");
    HIPCK(c, hipSetDevice(c->cfg.device));
    const char* hb = static_cast<const char*>(handles);
    for (int j = 0; j < ic->n; ++j) {
        if (j == ic->r) { ic->peer[j] = ic->region; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, hb + (size_t)j * sizeof h, sizeof h);
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            return tnml_fail(c, "tnml_oneshot_connect: hipIpcOpenMemHandle of rank %d's region failed (peer access between the two devices?)", j);
        }
        ic->peer[j] = static_cast<char*>(p); ic->opened[j] = true;
    }
    return 0;
}

void ipc_comm_release(tnml_ctx* c) {
    IpcComm* ic = c->ipc;
    if (!ic) return;
    c->ipc = nullptr;
    for (int j = 0; j < ic->n; ++j) if (ic->opened[j]) (void)hipIpcCloseMemHandle(ic->peer[j]);
    if (ic->region) (void)hipFree(ic->region);
    if (ic->h_status) (void)hipHostFree(ic->h_status);
    delete ic;
}
int ipc_comm_mem_kind(const tnml_ctx* c) { return c->ipc ? c->ipc->mem_kind : 0; }
int tnml_oneshot_mem_kind(tnml_ctx* c) { return c ? ipc_comm_mem_kind(c) : 0; }

// op: 0 = sum of doubles, 1 = copy of rank 0's values; in stream order, never blocks the host
int ipc_comm_exchange(tnml_ctx* c, double* buf, size_t count, int op) {
    IpcComm* ic = c->ipc;
    if (!ic->peer[ic->r]) return tnml_fail(c, "one-shot all-reduce: tnml_oneshot_connect was not called");
    if (count > ic->cap) return tnml_fail(c, "one-shot all-reduce: %zu elements exceed the slot capacity %zu", count, ic->cap);
    if (ic->h_status[0] != 0) return tnml_fail(c, "one-shot all-reduce: collective %llu timed out waiting for a peer (comm_timeout_s = %d)", ic->h_status[0], c->comm_timeout_s);
    if (count == 0) return 0;
    OsArgs a;
    a.buf = buf; a.count = count; a.seq = ++ic->seq; a.n = ic->n; a.r = ic->r; a.op = op; a.cap = ic->cap; a.nbmax = ic->nbmax;
    a.parity = (int)(a.seq & 1);
    for (int j = 0; j < OS_MAXR; ++j) a.peer[j] = j < ic->n ? ic->peer[j] : nullptr;
    a.recv_off = ic->recv_off; a.h_status = ic->h_status;
    a.timeout_ticks = (long long)c->comm_timeout_s * 100000000ll;       // wall_clock64: 100 MHz
    const int nb = (int)((count + OS_CHUNK - 1) / OS_CHUNK);
    static const bool trace = getenv("TNML_IPC_TRACE") != nullptr;       // debugging aid: the host-side sequence of collectives of every rank
    if (trace) fprintf(stderr, "[ipc rank %d] seq %llu op %d count %zu chunks %d\n", ic->r, a.seq, op, count, nb);
    hipLaunchKernelGGL(k_os_exchange, dim3(nb < OS_MAXWG ? nb : OS_MAXWG), dim3(256), 0, c->stream, a);
    HIPCK(c, hipGetLastError());
    return 0;
}
int ipc_comm_check(tnml_ctx* c) {
    if (c->ipc && c->ipc->h_status[0] != 0) return tnml_fail(c, "one-shot all-reduce: collective %llu timed out waiting for a peer (comm_timeout_s = %d)", c->ipc->h_status[0], c->comm_timeout_s);
    return 0;
}
