// Micro-benchmark of the sequential 64 x 64 tile routines (one block, as on the critical path of the
// blocked Cholesky): potf2_tile and trsm_tile, with a validation of potf2_tile against a host Cholesky.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tile_bench tile_bench.hip
#include "../gaussianprocessnode_amd/csrc/sgp_kernels.hip.h"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace sgp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_potf2_only(double* A, int ld, int reps, int* info) {
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    __shared__ double colp[4 * DPB];
    __shared__ double rinv[TB];
    for (int it = 0; it < reps; ++it) {
        tile_g2s(lds, A, ld, 0, 0);
        __syncthreads();
        potf2_tile(lds, colp, rinv, info, 0, 64);
    }
    tile_s2g(lds, A + 64 * 64, ld, 0, 0);
}
__global__ void __launch_bounds__(256) k_trsm_only(double* A, int ld, int reps, int* info) {
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    __shared__ double colp[4 * DPB];
    __shared__ double rinv[TB];
    tile_g2s(lds, A, ld, 0, 0);
    __syncthreads();
    potf2_tile(lds, colp, rinv, info, 0, 64);
    double* X = lds + TB * LT;
    for (int it = 0; it < reps; ++it) {
        tile_g2s(X, A, ld, 0, 0);
        __syncthreads();
        trsm_tile(X, lds, colp, rinv);
        __syncthreads();
    }
    tile_s2g(X, A + 64 * 64, ld, 0, 0);
}
__global__ void __launch_bounds__(256) k_trtri_only(double* A, int ld, int reps, int* info) {
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    __shared__ double work[32 * 33];
    __shared__ double colp[4 * DPB];
    __shared__ double rinv[TB];
    tile_g2s(lds, A, ld, 0, 0);
    __syncthreads();
    potf2_tile(lds, colp, rinv, info, 0, 64);
    double* X = lds + TB * LT;
    for (int it = 0; it < reps; ++it) {
        trtri_tile(lds, rinv, X, work);
        __syncthreads();
    }
    tile_s2g(X, A + 64 * 64, ld, 0, 0);
}
__global__ void __launch_bounds__(256) k_copy_only(double* A, int ld, int reps) {
    __shared__ double lds[2 * TB * PS];
    for (int it = 0; it < reps; ++it) {
        tile_g2s(lds, A, ld, 0, 0);
        __syncthreads();
    }
    tile_s2g(lds, A + 64 * 64, ld, 0, 0);
}

int main() {
    const int n = 64;
    std::vector<double> A(2 * n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[j * n + i] = (i == j ? 2.0 : 0.0) + 1.0 / (1.0 + std::abs(i - j));
    double* dA; int* dInfo;
    CK(hipMalloc(&dA, 2 * n * n * 8)); CK(hipMalloc(&dInfo, 4)); CK(hipMemset(dInfo, 0, 4));
    CK(hipMemcpy(dA, A.data(), 2 * n * n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    auto run = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-24s %8.2f us per tile  (%6.0f cycles/pivot @2.4GHz)\n", name, 1e3 * ms / reps, 1e3 * ms / reps * 2400 / 64);
    };
    run("copy g2s only", [&] { k_copy_only<<<1, 256>>>(dA, n, reps); });
    run("potf2_tile (+copy)", [&] { k_potf2_only<<<1, 256>>>(dA, n, reps, dInfo); });
    run("trsm_tile (+copy)", [&] { k_trsm_only<<<1, 256>>>(dA, n, reps, dInfo); });
    run("trtri_tile", [&] { k_trtri_only<<<1, 256>>>(dA, n, reps, dInfo); });
    // validation of potf2_tile against a host Cholesky
    {
        CK(hipMemset(dInfo, 0, 4));
        k_potf2_only<<<1, 256>>>(dA, n, 1, dInfo); CK(hipDeviceSynchronize());
        std::vector<double> L(n * n), R(n * n);
        CK(hipMemcpy(L.data(), dA + n * n, n * n * 8, hipMemcpyDeviceToHost));
        int info; CK(hipMemcpy(&info, dInfo, 4, hipMemcpyDeviceToHost));
        R = std::vector<double>(A.begin(), A.begin() + n * n);
        for (int j = 0; j < n; ++j) {
            double d = R[j * n + j]; for (int k = 0; k < j; ++k) d -= R[k * n + j] * R[k * n + j];
            d = std::sqrt(d); R[j * n + j] = d;
            for (int i = j + 1; i < n; ++i) { double v = R[j * n + i]; for (int k = 0; k < j; ++k) v -= R[k * n + i] * R[k * n + j]; R[j * n + i] = v / d; }
        }
        double err = 0; int wi = -1, wj = -1;
        for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) { double e = std::abs(L[j * n + i] - R[j * n + i]); if (e > err) { err = e; wi = i; wj = j; } }
        printf("potf2 validation: info %d max |L - L_ref| = %.3e at (%d,%d); L[0..4][0..4]:\n", info, err, wi, wj);
        for (int i = 0; i < 5; ++i) { for (int j = 0; j < 5; ++j) printf(" %10.6f/%10.6f", L[j * n + i], i >= j ? R[j * n + i] : 0.0); printf("\n"); }
        printf("diag:"); for (int i = 0; i < 24; ++i) printf(" %.4f/%.4f", L[i * n + i], R[i * n + i]); printf("\n");
        printf("row 8:"); for (int j = 0; j <= 8; ++j) printf(" %.4f/%.4f", L[j * n + 8], R[j * n + 8]); printf("\n");
    }
    return 0;
}
