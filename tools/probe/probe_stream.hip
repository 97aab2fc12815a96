// What does the HBM deliver to 4 streaming waves per CU (256 workgroups x 256 lanes) that read the Label-carrying environment
// [10][120][NTp] fp64 in tiles of TI images?  Access pattern = row segments of TI*8 bytes, VEC doubles per lane, RING row-groups in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int TI, int VEC, int RING, int NT_>
__global__ __launch_bounds__(256) void k(const double* __restrict__ EL, size_t lstride, int NTp, int ntiles, double* out) {
    const int lane = threadIdx.x & 63, sw = threadIdx.x >> 6;
    constexpr int LPR = TI / VEC;              // lanes per row segment
    constexpr int RPW = 64 / LPR;              // rows per wave-load
    constexpr int STEPS = 120 / (4 * RPW);     // row-groups per tile per wave (all 120 rows over 4 waves)
    const int col = (lane % LPR) * VEC, qs = lane / LPR;
    double acc[VEC] = {0.};
    typedef double dv __attribute__((ext_vector_type(VEC)));
    dv ring[RING][10];
    int issued = 0, used = 0;
    const int total = ((ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * STEPS;
    auto addr = [&](int idx, int l) {
        const int t = blockIdx.x + (idx / STEPS) * gridDim.x, k = idx % STEPS;
        return EL + (size_t)l * lstride + (size_t)((k * 4 + sw) * RPW + qs) * NTp + (size_t)t * TI + col;
    };
    // simple software ring with compiler-tracked loads would drain; use explicit pipelining by unrolling RING deep
    for (int base = 0; base < total + RING; base += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            const int idx = base + r;
            if (idx >= RING && idx - RING < total) {
#pragma unroll
                for (int l = 0; l < 10; ++l)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += ring[r][l][v];
            }
            if (idx < total) {
#pragma unroll
                for (int l = 0; l < 10; ++l) ring[r][l] = NT_ ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(addr(idx, l))) : *reinterpret_cast<const dv*>(addr(idx, l));
            }
        }
    }
    double s = 0; for (int v = 0; v < VEC; ++v) s += acc[v];
    if (s == 1.2345) out[0] = s;
}
template <int TI, int VEC, int RING, int NT_>
static void run(const char* tag, const double* EL, int NTp, double* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TI, VEC, RING, NT_>), dim3(256), dim3(256), 0, 0, EL, (size_t)120 * NTp, NTp, NTp / TI, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("%-58s %7.1f us  %5.2f TB/s  (%d KB in flight per CU)\n", tag, best * 1e3, 10.0 * 120 * NTp * 8 / best / 1e9, RING * 10 * VEC * 8 * 256 / 1024);
}
int main() {
    const int NTp = 60160;
    double *EL, *out; hipMalloc(&EL, (size_t)10 * 120 * NTp * 8); hipMemset(EL, 0, (size_t)10 * 120 * NTp * 8); hipMalloc(&out, 64);
    run<32, 1, 4, 1>("32-image tiles, 8 B/lane, ring 4, nt", EL, NTp, out);
    run<32, 1, 6, 1>("32-image tiles, 8 B/lane, ring 6, nt", EL, NTp, out);
    run<32, 2, 2, 1>("32-image tiles, 16 B/lane, ring 2, nt", EL, NTp, out);
    run<32, 2, 3, 1>("32-image tiles, 16 B/lane, ring 3, nt", EL, NTp, out);
    run<64, 1, 4, 1>("64-image tiles, 8 B/lane, ring 4, nt", EL, NTp, out);
    run<64, 1, 6, 1>("64-image tiles, 8 B/lane, ring 6, nt", EL, NTp, out);
    run<64, 2, 2, 1>("64-image tiles, 16 B/lane, ring 2, nt", EL, NTp, out);
    run<64, 2, 3, 1>("64-image tiles, 16 B/lane, ring 3, nt", EL, NTp, out);
    run<128, 2, 2, 1>("128-image tiles, 16 B/lane, ring 2, nt", EL, NTp, out);
    run<128, 2, 3, 1>("128-image tiles, 16 B/lane, ring 3, nt", EL, NTp, out);
    run<32, 1, 4, 0>("32-image tiles, 8 B/lane, ring 4, default policy", EL, NTp, out);
    run<64, 2, 3, 0>("64-image tiles, 16 B/lane, ring 3, default policy", EL, NTp, out);
    return 0;
}
