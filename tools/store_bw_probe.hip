// Probe: what write bandwidth does this chip sustain?  k_gram_uf writes K_uf at 3.1 - 3.25 TB/s (N = 10^6) and is called store-bound; is that
// the ceiling of the memory system or of the kernel's store pattern?  Variants: hipMemsetAsync; a grid-stride kernel storing 16 bytes per
// lane (1 KB contiguous per wave instruction), plain / nontemporal; the Gram's pattern (a wave = 4 rows x 512 contiguous bytes, rows 4 KB
// apart); workgroups per CU 1 .. 8.   Build: hipcc --offload-arch=gfx950 -O3 -o store_bw_probe store_bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef double dv2 __attribute__((ext_vector_type(2)));
template <int NT>
__global__ void __launch_bounds__(256) k_store_linear(dv2* p, size_t n16, double v) {
    const dv2 x = {v, v + 1.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}
// the Gram's pattern: block (point block nb of 64 rows, column tile I of 64 doubles); thread (tm = tid & 15, tn = tid >> 4) stores
// 32 bytes at row nb 64 + 4 tn + b, column 64 I + 4 tm, b = 0 .. 3
__global__ void __launch_bounds__(256) k_store_gram(double* K, int Mp, size_t N, double v) {
    const int I = blockIdx.y * 64, tm = threadIdx.x & 15, tn = threadIdx.x >> 4;
    const size_t n0 = (size_t)blockIdx.x * 64;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const size_t n = n0 + tn * 4 + b;
        if (n < N) {
            double* dst = K + n * Mp + I + tm * 4;
            *reinterpret_cast<double2*>(dst) = make_double2(v, v + b);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(v + 2, v + 3);
        }
    }
}
// the same bytes with a row per wave: thread (lane = tid & 63, w = tid >> 6): 16 bytes at row nb 64 + 16 w + r (r = 0 .. 15), column 64 I .. : a wave
// instruction = 64 lanes x 16 B = ... only 512 B per row tile; so use a 128-column tile: lane -> column 2 lane of a 128-double (1 KB) run
__global__ void __launch_bounds__(256) k_store_rows(double* K, int Mp, size_t N, double v) {
    const int I = blockIdx.y * 128, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t n0 = (size_t)blockIdx.x * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t n = n0 + 16 * w + r;
        if (n < N) *reinterpret_cast<double2*>(K + n * Mp + I + 2 * lane) = make_double2(v, v + r);
    }
}
int main() {
    const size_t N = 1000000; const int Mp = 512;
    const size_t bytes = N * Mp * 8;
    double* d; CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) * 1e-12);
    };
    timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(d, 0, bytes, 0)); });
    for (int bpc : {1, 2, 4, 8, 16}) {
        char nm[64];
        snprintf(nm, 64, "linear 16 B/lane, %d blocks/CU", bpc);
        timeit(nm, [&] { k_store_linear<0><<<256 * bpc, 256>>>((dv2*)d, bytes / 16, 1.0); });
        snprintf(nm, 64, "linear nontemporal, %d blocks/CU", bpc);
        timeit(nm, [&] { k_store_linear<1><<<256 * bpc, 256>>>((dv2*)d, bytes / 16, 1.0); });
    }
    timeit("Gram pattern (64 x 64 tiles, 32 B per thread-row)", [&] { k_store_gram<<<dim3((unsigned)((N + 63) / 64), Mp / 64), 256>>>(d, Mp, N, 1.0); });
    timeit("rows pattern (64 x 128 tiles, 1 KB per wave-row)", [&] { k_store_rows<<<dim3((unsigned)((N + 63) / 64), Mp / 128), 256>>>(d, Mp, N, 1.0); });
    return 0;
}
