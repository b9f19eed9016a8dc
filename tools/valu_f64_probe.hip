// VALU FP64 FMA peak probe for gfx950: is the vector pipe's attainable rate above the matrix pipe's (49 TF measured with
// tools/mfma_f64_probe.hip)?  Also runs both pipes together from the same waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: VALU only, 1: MFMA only, 2: both interleaved
__global__ void __launch_bounds__(256) k_probe(double* out, int iters, double seed) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
    const double x = seed * 0.5, y = seed * 0.25;
    d4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fma(a[i], x, y);
        }
        if (MODE != 0) {
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m1, 0, 0, 0);
            if (MODE == 1) {
                m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m2, 0, 0, 0);
                m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m3, 0, 0, 0);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    s += m0[0] + m1[1] + m2[2] + m3[3];
    if (s == 12345.678) out[0] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, double* d) {
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<MODE><<<blocks, 256>>>(d, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<MODE><<<blocks, 256>>>(d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double valu = (MODE != 1) ? waves * iters * 16.0 * 64 * 2 : 0.0;
    const double mfma = (MODE == 1) ? waves * iters * 4.0 * 2048 : (MODE == 2 ? waves * iters * 2.0 * 2048 : 0.0);
    printf("%-28s %d blocks/CU  %8.3f ms   VALU %6.1f TF   MFMA %6.1f TF   sum %6.1f TF\n", name, blocks_per_cu, ms,
           valu / ms * 1e-9, mfma / ms * 1e-9, (valu + mfma) / ms * 1e-9);
}

int main() {
    double* d;
    hipMalloc(&d, 64);
    for (int b : {1, 2, 4}) {
        run<0>("VALU fma f64", b, d);
        run<1>("MFMA f64 16x16x4", b, d);
        run<2>("VALU + MFMA interleaved", b, d);
    }
    return 0;
}
