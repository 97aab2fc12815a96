// where do the waves of a 768-lane (and 1024-lane) workgroup land?  prints SIMD id per wave index (HW_REG_HW_ID bits 5:4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4); unsigned h[4096];
    for (int nt : {768, 1024, 512}) {
        hipMemset(d, 0, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(8), dim3(nt), 100 * 1024, 0, d);
        hipMemcpy(h, d, 4096 * 4, hipMemcpyDeviceToHost);
        for (int b = 0; b < 3; ++b) { printf("%d lanes, block %d: simd of waves:", nt, b); for (int w = 0; w < nt / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf("  (cu %u)\n", (h[b * 16] >> 8) & 15); }
    }
    return 0;
}
