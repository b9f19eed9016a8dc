// Which SIMD does wave w of a 512-thread workgroup run on?  (k_potrf_step's two groups of four waves share the SIMDs pairwise if
// wave w and wave w + 4 land on the same one.)   hipcc --offload-arch=gfx950 -O2 -o tools/simd_map_probe tools/simd_map_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k_probe(unsigned* out) {
    __shared__ double big[19000];                 // ~152 KB: one workgroup per CU, like the step kernel
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id + (unsigned)(big[threadIdx.x] * 0.0);
}
int main() {
    unsigned* d; const int nb = 12;
    hipMalloc(&d, nb * 8 * sizeof(unsigned));
    hipLaunchKernelGGL(k_probe, dim3(nb), dim3(512), 0, 0, d);
    unsigned h[nb * 8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; ++b) {
        printf("wg %2d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd %u wave_slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
