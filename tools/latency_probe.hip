// Dependent-issue latency of the instructions on the Cholesky pivot chain, gfx950, one wave: each kernel runs a long
// dependent chain of one instruction kind; cycles per link = elapsed / links at the measured shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(64) k_chain(double* out, long long* cyc, int iters, double seed) {
    double x = seed + threadIdx.x * 1e-3, y = 1.0 + seed * 1e-9;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) x = fma(x, y, 1e-9);                                   // v_fma_f64
            if (KIND == 1) x = __builtin_amdgcn_rcp(x) + 0.0 * y;                 // v_rcp_f64 (+ the compiler may fold the add)
            if (KIND == 2) x = __builtin_amdgcn_rsq(x);                           // v_rsq_f64
            if (KIND == 3) {                                                      // readlane -> VALU
                double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 5), __builtin_amdgcn_readlane(__double2loint(x), 5));
                x = fma(s, y, x);
            }
            if (KIND == 4) {                                                      // DPP quad broadcast -> VALU
                int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x55, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x55, 0xf, 0xf, true);
                x = fma(__hiloint2double(hi, lo), y, 1e-9);
            }
            if (KIND == 5) {                                                      // ds_bpermute -> VALU
                int a = 4 * ((threadIdx.x + 17) & 63);
                double s = __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(x)), __builtin_amdgcn_ds_bpermute(a, __double2loint(x)));
                x = fma(s, y, 1e-9);
            }
            if (KIND == 6) x = x * y;                                             // v_mul_f64
            if (KIND == 7) { float f = (float)x; f = fmaf(f, 1.0001f, 1e-9f); x = f; }   // cvt + f32 fma + cvt
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}

template <int KIND>
static int run(const char* name, double* d, long long* c, double seed) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_chain<KIND><<<1, 64>>>(d, c, 100, seed);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_chain<KIND><<<1, 64>>>(d, c, iters, seed);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long cy = 0;
    CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
    const double links = (double)iters * 16;
    printf("%-28s %8.1f ns/link   %8.1f counter ticks/link   (%.3f ms)\n", name, ms * 1e6 / links, cy / links, ms);
    return 0;
}

int main() {
    double* d; long long* c;
    CK(hipMalloc(&d, 64 * 8)); CK(hipMalloc(&c, 8));
    run<0>("v_fma_f64", d, c, 1.0);
    run<6>("v_mul_f64", d, c, 1.0);
    run<1>("v_rcp_f64 (+add)", d, c, 1.5);
    run<2>("v_rsq_f64", d, c, 1.5);
    run<3>("readlane x2 + fma", d, c, 1.0);
    run<4>("dpp x2 + fma", d, c, 1.0);
    run<5>("bpermute x2 + fma", d, c, 1.0);
    run<7>("cvt f64->f32, fma32, cvt back", d, c, 1.0);
    return 0;
}
