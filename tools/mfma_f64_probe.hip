// Probe for v_mfma_f64_16x16x4_f64 on gfx950: (1) A/B/C lane maps with exact integer data,
// (2) back-to-back issue rate -> measured FP64 MFMA peak (roofline denominator),
// (3) v_fma_f64 VALU rate for comparison.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// A is 16x4 (row-major A[i][k]), B is 4x16 (B[k][j]); hypothesis: lane l supplies A[l&15][l>>4], B[l>>4][l&15];
// D: lane l reg r holds D[row][col] with col = l&15 and row = ?  -> we dump raw registers and decode on the host.
__global__ void layout_kernel(const double* A, const double* B, double* Draw) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Draw[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_mfma(double* out, int iters) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    // (inline assembly, accumulators pinned in VGPRs: with the builtin the compiler copied every accumulator VGPR -> AGPR -> VGPR
    // around each round of this loop, 16 copies per MFMA, and the probe read 46 - 48 TFLOP/s where the instruction sustains 72 - 73)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) rate_valu(double* out, int iters) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    // ---- layout
    std::vector<double> A(64), B(64), D(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (i + 1) * (k == 0 ? 1 : (k == 1 ? 100 : (k == 2 ? 10000 : 1000000)));
    // B[k][j] picks k: B = one-hot weights so D[i][j] = sum_k A[i][k]*B[k][j]; use asymmetric B
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (j % 4 == k) ? (1 + j / 4) : 0;
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    std::vector<double> Dref(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; Dref[i * 16 + j] = s; }
    // test hypotheses
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int col = l & 15;
        int rowA = (l >> 4) + 4 * r;      // guide's f64 map
        int rowB = 4 * (l >> 4) + r;      // f32-style map
        if (D[l * 4 + r] != Dref[rowA * 16 + col]) okA = 0;
        if (D[l * 4 + r] != Dref[rowB * 16 + col]) okB = 0;
    }
    printf("layout: row=(lane>>4)+4*reg : %s ; row=4*(lane>>4)+reg : %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
    if (!okA && !okB) { for (int l = 0; l < 64; ++l) printf("lane %d: %g %g %g %g\n", l, D[l*4], D[l*4+1], D[l*4+2], D[l*4+3]); }
    // ---- rate
    int blocks = p.multiProcessorCount * 2;   // 2 blocks x 4 waves = 2 waves per SIMD
    double* dout; CK(hipMalloc(&dout, (size_t)blocks * 8 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 20000;
    auto timeit = [&](auto launch, const char* name, double flop_per_thread_block) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.3f ms  %8.2f TFLOP/s\n", name, ms, flop_per_thread_block / (ms * 1e-3) / 1e12);
    };
    for (int nb = 1; nb <= 8; nb *= 2) {
        int g = p.multiProcessorCount * nb;
        double fl = (double)g * 4 /*waves*/ * iters * 2048.0;
        char nm[64];
        snprintf(nm, 64, "mfma f64 acc=1 blk/CU=%d", nb); timeit([&] { rate_mfma<1><<<g, 256>>>(dout, iters); }, nm, fl * 1);
        snprintf(nm, 64, "mfma f64 acc=4 blk/CU=%d", nb); timeit([&] { rate_mfma<4><<<g, 256>>>(dout, iters); }, nm, fl * 4);
        snprintf(nm, 64, "mfma f64 acc=8 blk/CU=%d", nb); timeit([&] { rate_mfma<8><<<g, 256>>>(dout, iters); }, nm, fl * 8);
        snprintf(nm, 64, "valu fma f64 x16 blk/CU=%d", nb); timeit([&] { rate_valu<<<g, 256>>>(dout, iters); }, nm, (double)g * 256 * iters * 16 * 2.0);
    }
    return 0;
}
