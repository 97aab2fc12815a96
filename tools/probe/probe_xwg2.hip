// probe_xwg2.hip -- the ONE-HOP exchange of k_sytrd_ro (eigh_mc.hip) in isolation: per step every workgroup publishes the 64 values of
// its rows as tagged 16-byte granules and every thread polls the one value of its row (n values in all), nothing else.  Variants of the
// memory path:
//   form 0: stores sc0 sc1 (write through to memory), agent-scope atomic loads (sc1)              -- what the kernel does
//   form 1: all workgroups on ONE XCD (grid 8 P, blockIdx % 8 == 0 works); plain stores (the L1 is write-through: they reach the XCD's L2),
//           buffer_inv sc0 (invalidate the CU's L1) + plain loads: the hand-off stays inside that L2
//   form 2: as 1 with buffer_inv sc1
//   form 3: as 1 but loads with sc0 and no invalidate
//   form 4: form 0's instructions on the one-XCD placement
// A poll that does not see its tag within 2^16 attempts counts as a timeout and ends the run (a path that never becomes visible).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(512) void k_x(u32x4* __restrict__ xb, int P, int n, int steps, unsigned tag0, long long* __restrict__ out, int work) {
    __shared__ int s_stop;
    const bool one_xcd = FORM >= 1;
    if (one_xcd && (blockIdx.x & 7)) return;
    const int p = one_xcd ? blockIdx.x >> 3 : blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { s_stop = 0; unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); atomicOr((unsigned long long*)(out + 2), 1ull << (x & 15)); }
    __syncthreads();
    long long bad = 0, timeouts = 0;
    double acc = 0.;
    for (int k = 0; k < steps; ++k) {
        const unsigned tag = tag0 + (unsigned)k;
        u32x4* buf = xb + (size_t)(k & 1) * 1024;
        // publish: rows i with (i >> 3) % P == p, by the first threads
        if (tid < 64) {
            const int i = 8 * (p + (tid >> 3) * P) + (tid & 7);
            if (i < n) {
                u64 bits = ((u64)k * 1000003ull + (u64)i + 1ull) * 0x9E3779B97F4A7C15ull;
                for (int w = 0; w < work; ++w) bits = bits * 6364136223846793005ull + 1442695040888963407ull;
                u32x4 g; g.x = (unsigned)bits; g.y = tag; g.z = (unsigned)(bits >> 32); g.w = tag;
                if (FORM == 0 || FORM == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(buf + i), "v"(g) : "memory");
                else                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(buf + i), "v"(g) : "memory");
            }
        }
        for (int i = tid; i < n; i += 512) {
            u32x4 g;
            int spin = 0;
            bool ok;
            do {
                if (FORM == 0 || FORM == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(buf + i) : "memory");
                else if (FORM == 1)         asm volatile("buffer_inv sc0\n\tglobal_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(buf + i) : "memory");
                else if (FORM == 2)         asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(buf + i) : "memory");
                else                        asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(buf + i) : "memory");
                ok = g.y == tag && g.w == tag;
            } while (!ok && ++spin < (1 << 16));
            if (!ok) { ++timeouts; s_stop = 1; }
            const u64 got = (u64)g.x | ((u64)g.z << 32);
            u64 ref = ((u64)k * 1000003ull + (u64)i + 1ull) * 0x9E3779B97F4A7C15ull;
            for (int w = 0; w < work; ++w) ref = ref * 6364136223846793005ull + 1442695040888963407ull;
            if (ok && got != ref) ++bad;
            acc += (double)(got >> 60) * 0.;
        }
        __syncthreads();
        if (s_stop) break;
    }
    if (bad) atomicAdd((unsigned long long*)out, (unsigned long long)bad);
    if (timeouts) atomicAdd((unsigned long long*)(out + 1), (unsigned long long)timeouts);
    if (acc != 0.) out[3] = 1;
}

int main() {
    u32x4* xb; long long* out;
    HC(hipMalloc(&xb, 16 * 2048)); HC(hipMemset(xb, 0, 16 * 2048));
    HC(hipMalloc(&out, 32)); HC(hipMemset(out, 0, 32));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    unsigned tag = 1;
    const int steps = 2000;
    for (int form = 0; form < 5; ++form)
        for (int P : {1, 4, 8, 13, 16})
            for (int n : {320, 600}) {
                if (8 * 8 * P < n) continue;                           // 64 rows per workgroup at most
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    HC(hipEventRecord(e0));
                    const int grid = form >= 1 ? 8 * P : P;
                    switch (form) {
                        case 0: hipLaunchKernelGGL(k_x<0>, dim3(grid), dim3(512), 0, 0, xb, P, n, steps, tag, out, 0); break;
                        case 1: hipLaunchKernelGGL(k_x<1>, dim3(grid), dim3(512), 0, 0, xb, P, n, steps, tag, out, 0); break;
                        case 2: hipLaunchKernelGGL(k_x<2>, dim3(grid), dim3(512), 0, 0, xb, P, n, steps, tag, out, 0); break;
                        case 3: hipLaunchKernelGGL(k_x<3>, dim3(grid), dim3(512), 0, 0, xb, P, n, steps, tag, out, 0); break;
                        default: hipLaunchKernelGGL(k_x<4>, dim3(grid), dim3(512), 0, 0, xb, P, n, steps, tag, out, 0); break;
                    }
                    HC(hipGetLastError());
                    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                    tag += steps + 7;
                    float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                long long h[4]; HC(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost)); HC(hipMemset(out, 0, 32));
                printf("form %d P=%2d n=%d: %.3f us/step   mismatches %lld  timeouts %lld  XCC ids seen 0x%llx\n", form, P, n, best * 1000.f / steps, h[0], h[1], h[2]);
                fflush(stdout);
            }
    return 0;
}
