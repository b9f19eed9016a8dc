// Probe for a streaming SYRK WITHOUT LDS staging: every wave owns a whole 64 x 64 tile of Psi2 = K_uf^T-products over its share of
// the points (16 accumulators of v_mfma_f64_16x16x4_f64) and loads its operands from global memory straight into the registers the
// matrix instruction reads (lane (lk, li) of operand block t: K_uf[n0 + lk][64 I + 16 t + li] -- 16 lanes = one 128-byte line), P
// k-steps ahead.  No barrier, no LDS traffic, no cross-wave meeting in the loop; the waves of a workgroup add their partial tiles up
// in LDS once at the end.  Question: does it beat the LDS-staged kernels (k_syrk_stream16: 49 TFLOP/s at N = 10^6, 38 at T)?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o syrk_direct_probe syrk_direct_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int TB = 64;

__device__ __forceinline__ void tile_from_index(int t, int& I, int& J) {
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    I = i;
    J = t - i * (i + 1) / 2;
}

template <int P, bool DIAG>
__device__ __forceinline__ void stream_tile(d4 (&acc)[4][4], const double* __restrict__ pa, const double* __restrict__ pb, size_t step, int nt) {
    double a[P][4], b[P][4];
    auto load = [&](int p, int t) {
        const double* qa = pa + (size_t)t * step;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[p][u] = qa[16 * u];
        if constexpr (!DIAG) {
            const double* qb = pb + (size_t)t * step;
#pragma unroll
            for (int u = 0; u < 4; ++u) b[p][u] = qb[16 * u];
        }
    };
    if (nt <= 0) return;
    auto mma = [&](int p) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p][ti], DIAG ? a[p][tj] : b[p][tj], acc[ti][tj], 0, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < P; ++p) load(p, p < nt ? p : nt - 1);
    // main loop: whole rounds of P k-steps, NO branch inside (the compiler's s_waitcnt placement is exact only for straight-line
    // bodies: with a conditional product in the round it drained all loads at the loop header); the loads past the end are clamped
    // to the last k-step (harmless re-reads), the last nt mod P products follow behind the loop
    int t0 = 0;
    for (; t0 + P <= nt; t0 += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            mma(p);
            const int tn = t0 + p + P;
            load(p, tn < nt ? tn : nt - 1);
        }
    }
#pragma unroll
    for (int p = 0; p < P - 1; ++p)
        if (t0 + p < nt) mma(p);
}

// grid: ntiles x nchunks items (chunk-major, cut into 8 runs for the XCDs); WAVES waves per workgroup share an item's points
template <int WAVES, int P>
__global__ void __launch_bounds__(64 * WAVES) k_syrk_direct(const double* __restrict__ Kuf, double* __restrict__ slabs, int Mp, int64_t N,
                                                            int tile0, int ntiles, int chunk, int nchunks) {
    __shared__ __attribute__((aligned(16))) double lds[4 * TB * TB];
    const int nitems = ntiles * nchunks, per = (nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (item >= nitems) return;
    const int chunk_id = item / ntiles, tile_id = item % ntiles;
    int I, J;
    tile_from_index(tile0 + tile_id, I, J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int64_t cbeg = (int64_t)chunk_id * chunk;
    int64_t cend = cbeg + chunk;
    if (cend > N) cend = N;
    const int nfull = cend > cbeg ? (int)((cend - cbeg) / 4) : 0;          // (the probe's N is a multiple of 4)
    const int nt = nfull > wave ? (nfull - wave + WAVES - 1) / WAVES : 0;    // this wave's k-steps: wave, wave + WAVES, ...
    d4 acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
    const double* pa = Kuf + (size_t)(cbeg + 4 * wave + lk) * Mp + I * TB + li;
    const double* pb = Kuf + (size_t)(cbeg + 4 * wave + lk) * Mp + J * TB + li;
    const size_t step = (size_t)4 * WAVES * Mp;
    if (I == J) stream_tile<P, true>(acc, pa, pb, step, nt);
    else stream_tile<P, false>(acc, pa, pb, step, nt);
    // partial tiles -> LDS ([i][j], 64 doubles per row), four waves at a time; everyone sums in wave order
    double out[(TB * TB) / (64 * WAVES)];
#pragma unroll
    for (int e = 0; e < (TB * TB) / (64 * WAVES); ++e) out[e] = 0.0;
    for (int round = 0; round < WAVES / 4; ++round) {
        if (round) __syncthreads();
        if ((wave >> 2) == round) {
            double* my = lds + (wave & 3) * (TB * TB);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) my[(16 * ti + lk + 4 * r) * TB + 16 * tj + li] = acc[ti][tj][r];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < (TB * TB) / (64 * WAVES) / 2; ++e) {
            const int idx = 2 * (tid + 64 * WAVES * e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 x = *reinterpret_cast<const double2*>(lds + q * (TB * TB) + idx);
                out[2 * e] += x.x; out[2 * e + 1] += x.y;
            }
        }
    }
    double* dst = slabs + ((size_t)chunk_id * ntiles + tile_id) * (TB * TB);
#pragma unroll
    for (int e = 0; e < (TB * TB) / (64 * WAVES) / 2; ++e) {
        const int idx = 2 * (tid + 64 * WAVES * e);
        *reinterpret_cast<double2*>(dst + idx) = make_double2(out[2 * e], out[2 * e + 1]);
    }
}

__global__ void k_fill(double* K, size_t count) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x)
        K[e] = 1e-3 * (double)((e * 2654435761ull >> 7) & 1023) - 0.5;
}

template <int WAVES, int P>
static void run(const char* name, const double* dK, double* dS, int Mp, int64_t N, int tile0, int ntiles, int cus, bool check) {
    const int want = cus / ntiles > 0 ? cus / ntiles : 1;
    int64_t per = (N + want - 1) / want;
    per = (per + 4 * WAVES - 1) / (4 * WAVES) * (4 * WAVES);
    const int nchunks = (int)((N + per - 1) / per);
    const int nitems = ntiles * nchunks, grid = (nitems + 7) / 8 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { k_syrk_direct<WAVES, P><<<grid, 64 * WAVES>>>(dK, dS, Mp, N, tile0, ntiles, (int)per, nchunks); };
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double flop = (double)N * ntiles * 2.0 * TB * TB;      // executed (full tiles, diagonal ones included)
    printf("%-22s N=%-8lld tiles %2d chunks %3d x %5lld points: %9.1f us  %6.2f TFLOP/s executed\n", name, (long long)N, ntiles, nchunks,
           (long long)per, best * 1e3, flop / (best * 1e-3) * 1e-12);
    if (check) {
        std::vector<double> S((size_t)nitems * TB * TB), K((size_t)N * Mp);
        CK(hipMemcpy(S.data(), dS, S.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(K.data(), dK, K.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t : {0, ntiles / 2, ntiles - 1}) {
            int I = 0; while ((I + 1) * (I + 2) / 2 <= tile0 + t) ++I; const int J = tile0 + t - I * (I + 1) / 2;
            for (int i : {0, 17, 63}) for (int j : {0, 5, 62}) {
                double ref = 0, got = 0;
                for (int64_t n = 0; n < N; ++n) ref += K[n * Mp + I * TB + i] * K[n * Mp + J * TB + j];
                for (int c = 0; c < nchunks; ++c) got += S[((size_t)c * ntiles + t) * TB * TB + i * TB + j];
                worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 1e-300));
            }
        }
        printf("    check against a host sum: worst relative difference %.2e\n", worst);
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, Mp = 512;
    const int64_t NBIG = 1000000;
    double *dK, *dS;
    CK(hipMalloc(&dK, (size_t)NBIG * Mp * 8)); CK(hipMalloc(&dS, (size_t)4096 * TB * TB * 8));
    k_fill<<<4096, 256>>>(dK, (size_t)NBIG * Mp); CK(hipDeviceSynchronize());
    // T's group 0 (tile rows 5..7: tiles 15..35), the masked group's tiles on all CUs, all tiles; then N = 10^6
    run<4, 7>("4 waves, P = 7", dK, dS, Mp, 10000, 15, 21, cus, true);
    run<4, 5>("4 waves, P = 5", dK, dS, Mp, 10000, 15, 21, cus, false);
    run<8, 6>("8 waves, P = 6", dK, dS, Mp, 10000, 15, 21, cus, true);
    run<8, 4>("8 waves, P = 4", dK, dS, Mp, 10000, 15, 21, cus, false);
    run<4, 7>("4 waves, P = 7", dK, dS, Mp, 10000, 0, 15, 192, false);
    run<8, 6>("8 waves, P = 6", dK, dS, Mp, 10000, 0, 15, 192, false);
    run<4, 7>("4 waves, P = 7", dK, dS, Mp, 10000, 0, 36, cus, false);
    run<8, 6>("8 waves, P = 6", dK, dS, Mp, 10000, 0, 36, cus, false);
    run<4, 7>("4 waves, P = 7", dK, dS, Mp, NBIG, 0, 36, cus, false);
    run<4, 5>("4 waves, P = 5", dK, dS, Mp, NBIG, 0, 36, cus, false);
    run<8, 6>("8 waves, P = 6", dK, dS, Mp, NBIG, 0, 36, cus, false);
    run<8, 4>("8 waves, P = 4", dK, dS, Mp, NBIG, 0, 36, cus, false);
    return 0;
}
