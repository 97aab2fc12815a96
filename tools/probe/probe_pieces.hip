// Hardware probe (not product code): how does the HBM read rate of the gradient GEMM's access pattern depend on the
// length of the contiguous piece it takes from each environment row?
//
// k_bgemm64 reads, per workgroup, 320 rows (10 labels x 32 links) of the Label-carrying environment [row][image] for
// its own image range, one chunk of images at a time: 256-byte pieces (32 images) in the register-staged kernel,
// 128-byte pieces (16 images) in the LDS-DMA variant, at a row stride of NTp*8 = 512 KB.  k_labeldot takes 1 KB pieces
// and streams at 6.4 TB/s; both forms of k_bgemm64 move 4.6-4.7 TB/s.  This probe replays the bgemm traversal (same
// grid: 4 column tiles x 64 image splits, 12 waves) with nothing but the loads, for pieces of 128 B ... 1 KB, with the
// same number of load instructions and bytes in flight per wave (16 x 1 KB) in every variant.
//
//   hipcc --offload-arch=gfx950 -O3 -o probe_pieces probe_pieces.hip && ./probe_pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int NTP = 65536, NQ = 128, NLAB = 10, SPLIT = 1024, BATCH = 16;

// PL lanes (16 B each) per row piece -> piece = 16*PL bytes = 2*PL images
template <int PL, int NT>
__global__ __launch_bounds__(768) void k_pieces(const double* __restrict__ E, double* __restrict__ out) {
    constexpr int RPI = 64 / PL;                  // rows per wave instruction
    constexpr int KT = 2 * PL;                    // images per chunk
    constexpr int NCH = SPLIT / KT;
    constexpr int GRP = 320 / RPI;                // instructions per chunk and workgroup
    constexpr int TOTAL = NCH * GRP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int by = blockIdx.y, split = blockIdx.z;
    const int rsub = lane / PL, col = (lane % PL) * 2;
    double s0 = 0., s1 = 0.;
    for (int t0 = wid; t0 < TOTAL; t0 += 12 * BATCH) {
        d2v v[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int t = t0 + 12 * i;
            v[i] = d2v{0., 0.};
            if (t < TOTAL) {
                const int ch = t / GRP, g = t % GRP;
                const int rloc = g * RPI + rsub, l = rloc >> 5, q = rloc & 31;
                const double* p = E + ((size_t)l * NQ + 32 * by + q) * NTP + (size_t)split * SPLIT + ch * KT + col;
                v[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p)) : *reinterpret_cast<const d2v*>(p);
            }
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) { s0 += v[i].x; s1 += v[i].y; }
    }
    if (s0 + s1 == 12345.678) out[0] = s0;        // keep the loads alive
}

template <int PL, int NT>
static void run(const double* E, double* out, hipStream_t st) {
    dim3 grid(1, 4, NTP / SPLIT), block(768);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_pieces<PL, NT>), grid, block, 0, st, E, out);
    CK(hipEventRecord(e0, st));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_pieces<PL, NT>), grid, block, 0, st, E, out);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)NLAB * NQ * NTP * 8.;
    printf("piece %4d B (%3d images)  %s loads: %7.1f us  %7.1f GB/s\n", 16 * PL, 2 * PL, NT ? "non-temporal" : "default     ", 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    const size_t n = (size_t)NLAB * NQ * NTP;
    double *E, *out;
    CK(hipMalloc(&E, n * sizeof(double)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(E, 0, n * sizeof(double)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("bgemm traversal of a [%d x %d][%d] fp64 environment (%.0f MB), 256 workgroups x 12 waves, 16 KB in flight per wave\n", NLAB, NQ, NTP, n * 8. / 1e6);
    run<8, 1>(E, out, st); run<16, 1>(E, out, st); run<32, 1>(E, out, st); run<64, 1>(E, out, st);
    run<8, 0>(E, out, st); run<16, 0>(E, out, st); run<32, 0>(E, out, st); run<64, 0>(E, out, st);
    return 0;
}
