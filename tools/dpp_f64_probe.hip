// Probe: is the FP64 outer product faster on the vector pipe than on the matrix pipe on gfx950?
// v_fmac_f64 is a VOP2 instruction and the DP ALU accepts ONE DPP control for 64-bit operands: row_newbcast:N (lane N of each
// 16-lane row as src0).  A 16 x 16 outer-product step is then 16 v_fmac_f64_dpp with NO operand traffic beyond the two doubles a
// lane loads (the matrix instruction v_mfma_f64_16x16x4_f64 attains 46 - 48 TFLOP/s on this chip, plain v_fma_f64 67:
// tools/valu_f64_probe.hip).  Measures (1) correctness of the lane map, (2) the register-only issue rate, (3) an LDS-fed 32 x 32
// block per wave with a four-way split of k over the DPP rows, against (4) the MFMA tile product fed from the same LDS panels.
// Build: hipcc --offload-arch=gfx950 -O3 -o dpp_f64_probe dpp_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#define FMAC_BC(acc, a, b, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b))
#define OUTER16(acc, a, b)                                                                                   \
    FMAC_BC(acc[0], a, b, 0); FMAC_BC(acc[1], a, b, 1); FMAC_BC(acc[2], a, b, 2); FMAC_BC(acc[3], a, b, 3);   \
    FMAC_BC(acc[4], a, b, 4); FMAC_BC(acc[5], a, b, 5); FMAC_BC(acc[6], a, b, 6); FMAC_BC(acc[7], a, b, 7);   \
    FMAC_BC(acc[8], a, b, 8); FMAC_BC(acc[9], a, b, 9); FMAC_BC(acc[10], a, b, 10); FMAC_BC(acc[11], a, b, 11); \
    FMAC_BC(acc[12], a, b, 12); FMAC_BC(acc[13], a, b, 13); FMAC_BC(acc[14], a, b, 14); FMAC_BC(acc[15], a, b, 15)

// (1) lane map: acc[n] of lane l = a[16 (l / 16) + n] * b[l]
__global__ void k_map(const double* a, const double* b, double* out) {
    const int l = threadIdx.x;
    double x = a[l], y = b[l];
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    OUTER16(acc, x, y);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = acc[i];
}

// (2) register-only rate: NB blocks of 16 accumulators
template <int NB>
__global__ void __launch_bounds__(256) k_rate(double* out, int iters, double seed) {
    double acc[NB][16];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0;
    double a = seed + threadIdx.x * 1e-9, b = seed - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NB; ++j) { OUTER16(acc[j], a, b); }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.678) out[0] = s;
}

// (3) LDS-fed: a wave owns a 32 x 32 block of C = A^T B (panels As[k][64], Bs[k][64], row stride LP), DPP row r takes k = 4 m + r
constexpr int KP = 32;          // k rows in the panels
constexpr int LP = 68;          // row stride (doubles): rows r, r+1, r+2, r+3 start 4 banks-of-8-bytes apart
__global__ void __launch_bounds__(256) k_lds_dpp(double* out, int iters, double seed) {
    __shared__ double As[KP * LP], Bs[KP * LP];
    for (int e = threadIdx.x; e < KP * LP; e += 256) { As[e] = seed + e * 1e-6; Bs[e] = seed - e * 1e-6; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 4, l = lane & 15;
    const int i0 = 32 * (wave >> 1), j0 = 32 * (wave & 1);
    double acc[4][16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int m = 0; m < KP / 4; ++m) {
            const double* pa = As + (4 * m + r) * LP + i0 + l;
            const double* pb = Bs + (4 * m + r) * LP + j0 + l;
            const double a0 = pa[0], a1 = pa[16], b0 = pb[0], b1 = pb[16];
            OUTER16(acc[0], a0, b0); OUTER16(acc[1], a0, b1); OUTER16(acc[2], a1, b0); OUTER16(acc[3], a1, b1);
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.678) out[0] = s;
}

// (4) the MFMA form of the same product: wave = 32 x 32 block as 2 x 2 MFMA tiles, operands P[k][i]
__global__ void __launch_bounds__(256) k_lds_mfma(double* out, int iters, double seed) {
    __shared__ double As[KP * LP], Bs[KP * LP];
    for (int e = threadIdx.x; e < KP * LP; e += 256) { As[e] = seed + e * 1e-6; Bs[e] = seed - e * 1e-6; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lk = lane >> 4, li = lane & 15;
    const int i0 = 32 * (wave >> 1), j0 = 32 * (wave & 1);
    d4 c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int m = 0; m < KP / 4; ++m) {
            const double* pa = As + (4 * m + lk) * LP + i0 + li;
            const double* pb = Bs + (4 * m + lk) * LP + j0 + li;
            const double a0 = pa[0], a1 = pa[16], b0 = pb[0], b1 = pb[16];
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
        }
    }
    double s = c00[0] + c01[1] + c10[2] + c11[3];
    if (s == 12345.678) out[0] = s;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d\n", p.gcnArchName, p.multiProcessorCount);
    {
        std::vector<double> a(64), b(64), o(1024);
        for (int i = 0; i < 64; ++i) { a[i] = i + 1; b[i] = 1000.0 + i; }
        double *da, *db, *dout;
        CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dout, 8192));
        CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
        k_map<<<1, 64>>>(da, db, dout); CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int l = 0; l < 64; ++l) for (int n = 0; n < 16; ++n) if (o[l * 16 + n] != a[16 * (l / 16) + n] * b[l]) ok = 0;
        printf("lane map acc[n](lane l) = a[16 (l / 16) + n] b[l]: %s\n", ok ? "MATCH" : "NO");
        if (!ok) for (int l = 0; l < 64; l += 5) printf("  lane %d: %g %g %g (b = %g)\n", l, o[l * 16], o[l * 16 + 1], o[l * 16 + 15], b[l]);
    }
    double* d; CK(hipMalloc(&d, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char* name, int bpc, double flop) {
        launch(100); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); launch(0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-34s %d blocks/CU %8.3f ms  %7.2f TFLOP/s\n", name, bpc, best, flop / (best * 1e-3) / 1e12);
    };
    const int cus = p.multiProcessorCount;
    for (int bpc : {1, 2, 4}) {
        const int g = cus * bpc;
        const int iters = 20000;
        const double per_outer = 16.0 * 64 * 2;         // flop of one OUTER16 of one wave
        timeit([&](int w) { k_rate<1><<<g, 256>>>(d, w ? w : iters, 1.0); }, "fmac_dpp regs, 16 acc", bpc, (double)g * 4 * iters * per_outer);
        timeit([&](int w) { k_rate<4><<<g, 256>>>(d, w ? w : iters, 1.0); }, "fmac_dpp regs, 64 acc", bpc, (double)g * 4 * iters * 4 * per_outer);
        const int it2 = 2000;
        timeit([&](int w) { k_lds_dpp<<<g, 256>>>(d, w ? w : it2, 1.0); }, "fmac_dpp LDS-fed 32x32/wave", bpc, (double)g * 4 * it2 * (KP / 4) * 4 * per_outer);
        timeit([&](int w) { k_lds_mfma<<<g, 256>>>(d, w ? w : it2, 1.0); }, "mfma LDS-fed 32x32/wave", bpc, (double)g * 4 * it2 * (KP / 4) * 4 * 2048.0);
    }
    return 0;
}
