// sgp_kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels of the sparse-GP VMP sweep.
//
// Everything is FP64 (cond(Lambda) ~ 7e8 at the kin40k operating point).
// Dense contractions run on v_mfma_f64_16x16x4_f64 through one LDS-panel tile routine
// (tile_mma_64x64); the lane maps below were verified on hardware with tools/mfma_f64_probe.hip:
//     A operand: lane l holds A[i = l & 15][k = l >> 4]        (16 x 4)
//     B operand: lane l holds B[k = l >> 4][j = l & 15]        (4 x 16)
//     C/D      : lane l, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
//
// Storage conventions (device):
//     * square matrices: column-major, leading dimension = padded size (multiple of TB = 64);
//       the pad block of an SPD matrix is the identity, of a statistics matrix zero.
//     * K_uf: column-major Mp x N (column n = K(Xu, x_n), the reference's Psi1_trans of point n).
//     * LDS operand panels: P[k][i] with row stride PS = 80 doubles, so that the two k-rows read by
//       one 32-lane group of ds_read_b64 fall on opposite halves of the 64 banks (conflict-free).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace sgp {

constexpr int TB = 64;          // tile edge of every blocked algorithm
constexpr int PS = 80;          // LDS panel row stride (doubles): 2*PS*2 dwords = 32 (mod 64)
constexpr int KB = 16;          // points (k extent) per LDS stage of the streaming SYRK
constexpr int MAXD = 32;        // max input dimension
constexpr int MAXO = 4;         // max outputs of a MultiSGP node
constexpr int LT = TB + 1;      // row stride of a 64 x 64 LDS tile

typedef double d4 __attribute__((ext_vector_type(4)));

// Parameters that change from sweep to sweep (theta, w): kept in device memory so that a captured
// hipGraph replays with fresh values (the graph's first node copies them from pinned host memory).
struct Params {
    double sigma2;
    double jitter;
    double E_logw;
    double n_override;          // unused (<0)
    double inv_ell[MAXD];
    double W[MAXO * MAXO];      // mean(q_w): scalar w_bar in W[0] for UniSGP, d_out x d_out for MultiSGP
    double prior_iso;           // prior precision on the diagonal when prior form = isotropic
    double pad[7];
};

__device__ __forceinline__ int64_t realtime_ticks() { return (int64_t)__builtin_amdgcn_s_memrealtime(); }

// Diagnostics (variant library built with -DSGP_SWEEP_TRACE, tools/sweep_trace.py): 100 MHz begin / end stamps of the kernels of
// the LAST sweep, taken inside the kernels -- rocprofv3's kernel trace slows the host's launches down until the GPU starves,
// so it cannot show how the streams of an overlapped sweep really interleave.  Slot map: 0 k_prep_xu (statistics), 1 k_prep_xu
// (K_uu chain), 2 k_gram_uf, 3 k_gram_uu, 4 k_trmv_mu_scan, 5 / 6 k_gemm32 (Sigma launch / K_uu^-1), 7 k_scalars, 8 k_join_wait,
// 16 + j Lambda-chain step j, 40 + j K_uu-chain step j, 64 + tile0 k_syrk_stream, 128 + row_lo k_assemble, 200 + j: the
// moment step j of the Lambda chain had its statistics (end of its wait).
constexpr int TRACE_SLOTS_N = 256;
__device__ long long g_sweep_trace[TRACE_SLOTS_N * 65];
#ifdef SGP_SWEEP_TRACE
__device__ __forceinline__ void trace_begin(int slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && slot < TRACE_SLOTS_N) {
        g_sweep_trace[slot * 65] = realtime_ticks();
        for (int e = 1; e < 65; ++e) g_sweep_trace[slot * 65 + e] = 0;
    }
}
__device__ __forceinline__ void trace_mark(int slot) {
    if ((threadIdx.x & 255) == 0 && slot < TRACE_SLOTS_N) g_sweep_trace[slot * 65] = realtime_ticks();
}
__device__ __forceinline__ void trace_end(int slot) {
    if ((threadIdx.x & 63) == 0 && slot < TRACE_SLOTS_N)      // every wave: the waves of a workgroup leave at different times
        atomicMax(&g_sweep_trace[slot * 65 + 1 + ((blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) & 63)], (long long)realtime_ticks());
}
#else
__device__ __forceinline__ void trace_begin(int) {}
__device__ __forceinline__ void trace_mark(int) {}
__device__ __forceinline__ void trace_end(int) {}
#endif
struct TraceScope {                     // begin at construction, end at every exit of the kernel (nothing in the product build)
    int slot;
    __device__ __forceinline__ explicit TraceScope(int s) : slot(s) { trace_begin(s); }
    __device__ __forceinline__ ~TraceScope() { trace_end(slot); }
};

// deterministic workgroup sum (wave butterflies, then the waves in order); `red` = one double of LDS per wave
__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// ------------------------------------------------------------------------------------------------
// timestamp marker: one thread writes the 100 MHz constant clock (bench.py's per-phase durations)
// ------------------------------------------------------------------------------------------------
// Phase stamps.  One record of STAMP_STRIDE int64 per phase slot: [0] begin, [1] end, [2 .. 65] exit ticks spread over
// 64 addresses.  Every workgroup stamping the SAME address was measured to cost 18 us on the K_uf kernel and 9 us on the
// streaming SYRK (1256 / 1152 same-address 64-bit atomics serialise at ~14 ns each, and the entry atomic sits in front of
// the workgroup's first loads): the entry is therefore stamped by workgroup 0 alone (the first dispatched), the exits by
// every workgroup but onto 64 different addresses, folded into `end` by stamp_accumulate.
constexpr int STAMP_STRIDE = 2 + 64;
__device__ __forceinline__ void stamp_enter(int64_t* rec) {
    if (rec && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        atomicMin(reinterpret_cast<long long*>(rec), (long long)realtime_ticks());
}
__device__ __forceinline__ void stamp_exit(int64_t* rec) {
    if (rec && threadIdx.x == 0)
        atomicMax(reinterpret_cast<long long*>(rec + 2 + ((blockIdx.x + 7 * blockIdx.y) & 63)), (long long)realtime_ticks());
}
// end of a sweep (one wave): fold the spread exits, close the sweep stamp, totals[i] += end_i - begin_i, totals[count] += 1.
// Slot 5 is derived: the idle time between the end of slot 7 (LOCAL) and the begin of slot 3 (FINISH1).  Slot numbers are
// include/sgp_hip.h's SGP_T_*.
// Runs at the very end of the sweep's last kernel, i.e. on the critical path of back-to-back sweeps: every load is issued
// before the first store (one memory round trip), lane i then owns slot i (one read-modify-write of totals[i] per lane, in
// parallel).  The first version walked the slots one after another -- a load, a fold and a store per slot, then 8 more
// dependent round trips by lane 0 -- and kept the GPU ~9 us per sweep between k_scalars' exit stamp and the next sweep.
struct StampFold { long long b, e, tot; };
// part 1 (any time after the sweep's other kernels have finished -- k_scalars calls it first thing, so that the round trip
// hides behind its own reductions): lane i < 8 gets slot i's begin, folded end and running total
__device__ __forceinline__ StampFold stamp_fold_load(const int64_t* stamps, const int64_t* totals, int lane) {
    constexpr int NSLOTS = 8;
    long long ex[NSLOTS];
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) ex[i] = stamps[i * STAMP_STRIDE + 2 + lane];
    const int slot = lane & (NSLOTS - 1);
    StampFold f;
    f.b = stamps[slot * STAMP_STRIDE];
    f.e = stamps[slot * STAMP_STRIDE + 1];
    f.tot = totals[slot];
    long long mine = 0;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        long long m = ex[i];
        for (int o = 32; o > 0; o >>= 1) {
            const long long other = __shfl_xor(m, o);
            m = other > m ? other : m;
        }
        mine = (slot == i) ? m : mine;
    }
    f.e = mine > f.e ? mine : f.e;
    return f;
}
// part 2 (the very end of the sweep's last kernel): close the sweep, store
__device__ __forceinline__ void stamp_fold_finish(StampFold f, int64_t* stamps, int64_t* totals, int lane) {
    constexpr int NSLOTS = 8, SWEEP = 0, FINISH1 = 3, FINISH2 = 4, GAP = 5, LOCAL = 7;
    constexpr long long UNSET = 0x7fffffffffffffffLL;
    const long long now = realtime_ticks();
    const int slot = lane & (NSLOTS - 1);
    long long b = f.b, e = f.e;
    if (slot == SWEEP || slot == FINISH2) e = now;        // this kernel is the end of both (its own exit atomic may be in flight)
    const long long local_end = __shfl(e, LOCAL), finish1_begin = __shfl(b, FINISH1);
    if (slot == GAP) { b = local_end; e = (finish1_begin != UNSET) ? finish1_begin : local_end; }
    if (lane < NSLOTS) {
        int64_t* last = totals + NSLOTS + 1;                  // this sweep's (begin, end) pairs, for sgp_get_timestamps
        last[2 * slot] = b;
        last[2 * slot + 1] = e;
        if (e > b && b != UNSET) totals[slot] = f.tot + (e - b);
        // ready for the next sweep: the begins are taken with atomicMin, so they start from "unset"; the exits are folded
        // with max and time only grows, so they need no reset
        stamps[slot * STAMP_STRIDE] = UNSET;
        stamps[slot * STAMP_STRIDE + 1] = 0;
    }
    if (lane == 0) totals[NSLOTS] += 1;
}
__device__ __forceinline__ void stamp_accumulate(int64_t* stamps, int64_t* totals, int lane) {
    stamp_fold_finish(stamp_fold_load(stamps, totals, lane), stamps, totals, lane);
}

// ------------------------------------------------------------------------------------------------
// Device words between streams.  Every wait on one is BOUNDED (a waiter that spins forever can hang the GPU for everyone on
// the host) and a wait that gives up says so: it ORs its bit into the handle's sync-status word (dInfo[3]), which the
// getters turn into SGP_ERR_HIP -- what the waiter was protecting is then not to be trusted.
// ------------------------------------------------------------------------------------------------
enum { SYNC_LATE_DONE = 1,        // the K_uu chain started before the previous sweep's done word (its outputs were still being read)
       SYNC_LATE_GRAD_START = 2,  // the K_uu half of the theta gradient started before the sweep's done word
       SYNC_LATE_KINV = 4,        // the Sigma launch gave up waiting for K_uu^-1 (the traces are NaN)
       SYNC_LATE_GRAD_JOIN = 8,   // the gradient's finishing kernel gave up waiting for its K_uu half
       SYNC_LATE_COLUMN = 16 };   // a Lambda-chain step gave up waiting for its tile column of the statistics
constexpr int JOIN_SPIN_LIMIT = 1 << 21;    // default number of polls (>= ~0.5 us each) before a waiter gives up
__device__ __forceinline__ bool join_ready(const long long* w, long long need) {
    return __hip_atomic_load((const __attribute__((address_space(1))) long long*)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
}
// one thread's bounded wait; false = gave up (and the bit is set)
__device__ __forceinline__ bool spin_until(const long long* w, long long need, int limit, int* status, int bit) {
    int it = 0;
    while (!join_ready(w, need)) {
        if (++it >= limit) {
            if (status) atomicOr(status, bit);
            return false;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Xu (D x M, AoS) -> Xus (SoA, scaled by 1/ell, padded to Mp with zeros)
// ------------------------------------------------------------------------------------------------
// `hP` is the handle's PINNED host copy of the parameters, read over the bus by this first kernel of a launch sequence
// and mirrored into device memory (`dP`) for every later kernel -- one node less than a separate H2D copy in front of it.
// Also resets the Cholesky status word of the sequence (`info_reset`) and, on the main stream, the phase stamps.
__global__ void k_prep_xu(const double* __restrict__ Xu, double* __restrict__ Xus, const Params* __restrict__ hP,
                          Params* __restrict__ dP, int* __restrict__ info_reset, int M, int Mp, int D, int64_t* stamps,
                          int nslots, int sweep_slot, const long long* wait_word, long long wait_need,
                          const long long* gate_word, long long gate_need, int spin_limit, int* sync_status) {
    TraceScope trace(stamps ? 0 : 1);
    if (gate_word) {
        // ... and the chain's workgroups (each takes a whole CU's LDS) stay off the CUs until the sweep's streaming SYRK has
        // its single resident round on them (its last workgroup sets the gate when it starts): that grid is sized for ALL CUs.
        // (Scheduling only: giving up here costs time, not correctness, so no status bit.)
        if (threadIdx.x == 0) spin_until(gate_word, gate_need, spin_limit, nullptr, 0);
        __syncthreads();
    }
    if (wait_word) {
        // first kernel of the K_uu chain: the previous sweep's last kernel on the other stream still reads what this chain
        // overwrites (see k_scalars).  Giving up is reported (SYNC_LATE_DONE): the previous sweep's scalars may be wrong.
        if (threadIdx.x == 0) spin_until(wait_word, wait_need, spin_limit, sync_status, SYNC_LATE_DONE);
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        if (stamps) {                                               // first kernel of a sweep: reset the phase stamps
            for (int e = threadIdx.x; e < nslots * STAMP_STRIDE; e += blockDim.x) {
                const int i = e / STAMP_STRIDE, f = e % STAMP_STRIDE;
                int64_t v = 0;
                if (f == 0) v = (i == sweep_slot || i == nslots - 1) ? realtime_ticks() : 0x7fffffffffffffffLL;   // sweep and LOCAL begin here
                stamps[e] = v;
            }
        }
        const double* src = reinterpret_cast<const double*>(hP);
        double* dst = reinterpret_cast<double*>(dP);
        for (int e = threadIdx.x; e < (int)(sizeof(Params) / sizeof(double)); e += blockDim.x) dst[e] = src[e];
        if (info_reset && threadIdx.x == 0) *info_reset = 0;
    }
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Mp) return;
    for (int d = 0; d < D; ++d) Xus[(size_t)d * Mp + m] = (m < M) ? Xu[(size_t)m * D + d] * hP->inv_ell[d] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// K_uu (+ jitter I), padded with the identity.  One 64 x 64 tile per block, 16 entries per thread.
// ------------------------------------------------------------------------------------------------
// Two forms.  k_gram_uu_lds (coordinate panels in 32 KB of LDS) is what a sweep whose SYRK fills the chip launches: it is queued
// behind the gate, i.e. while that SYRK holds every byte of LDS, and therefore gets a CU only as SYRK workgroups leave -- which is
// the point: FP64 vector work beside the SYRK takes the FP64 pipe from its MFMAs, and that SYRK is on the sweep's critical path
// (round 4, profiles/r04_ab_log.txt [9]: the LDS-free form beside it, 51 instead of 42 us for group 0's launch).  k_gram_uu (no
// LDS; thread = row i with its D coordinates in registers, a column's coordinates wave-uniform scalar loads) is for the small
// problems, where the K_uu chain starts with the sweep and nothing is there to wait for (C1: 20 300 instead of 17 500 sweeps/s).
__global__ void __launch_bounds__(256) k_gram_uu_lds(const double* __restrict__ Xus, double* __restrict__ Kuu,
                                                 const Params* __restrict__ P, int M, int Mp, int D) {
    __shared__ double ui[MAXD * TB];
    __shared__ double uj[MAXD * TB];
    TraceScope trace(3);
    const int I = blockIdx.x * TB, J = blockIdx.y * TB;
    for (int t = threadIdx.x; t < D * TB; t += 256) {
        int d = t / TB, r = t % TB;
        ui[t] = Xus[(size_t)d * Mp + I + r];
        uj[t] = Xus[(size_t)d * Mp + J + r];
    }
    __syncthreads();
    const int i = threadIdx.x & 63;
    const int jg = threadIdx.x >> 6;
    const double s2 = P->sigma2, jit = P->jitter;
    for (int jj = 0; jj < 16; ++jj) {
        int j = jg * 16 + jj;
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) { double t = ui[d * TB + i] - uj[d * TB + j]; d2 = fma(t, t, d2); }
        int gi = I + i, gj = J + j;
        double v;
        if (gi < M && gj < M) v = s2 * exp(-0.5 * d2) + (gi == gj ? jit : 0.0);
        else v = (gi == gj) ? 1.0 : 0.0;
        Kuu[(size_t)gj * Mp + gi] = v;
    }
}

__global__ void __launch_bounds__(256) k_gram_uu(const double* __restrict__ Xus, double* __restrict__ Kuu,
                                                 const Params* __restrict__ P, int M, int Mp, int D) {
    TraceScope trace(3);
    const int I = blockIdx.x * TB, J = blockIdx.y * TB;
    const int i = threadIdx.x & 63;
    const int jg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double ui[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) ui[d] = (d < D) ? Xus[(size_t)d * Mp + I + i] : 0.0;
    const double s2 = P->sigma2, jit = P->jitter;
    const int gi = I + i;
    for (int jj = 0; jj < 16; ++jj) {
        const int gj = J + jg * 16 + jj;
        const double* uj = Xus + gj;                                   // (wave-uniform address)
        double d2 = 0.0;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < D) { const double t = ui[d] - uj[(size_t)d * Mp]; d2 = fma(t, t, d2); }
        double v;
        if (gi < M && gj < M) v = s2 * exp(-0.5 * d2) + (gi == gj ? jit : 0.0);
        else v = (gi == gj) ? 1.0 : 0.0;
        Kuu[(size_t)gj * Mp + gi] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// K_uf tile 64 (m) x 64 (n) per block; each thread a 4 x 4 micro-tile.  Also the per-block partial of
// B = K_uf * Yw  (Yw = omega .* y, n x d_out), reduced deterministically later by k_assemble.
//   X   : D x N AoS (one point per column), unscaled;  Yw : N x d_out column-major
//   Kuf : Mp x N column-major;  bpart : [nblk][d_out][Mp]
// ------------------------------------------------------------------------------------------------
// DCAP: capacity of the LDS coordinate panels (8 for D <= 8 -- 18 KB of LDS per workgroup instead of 42 KB, i.e. 8
// instead of 3 resident workgroups per CU for this store-bound kernel -- else MAXD)
template <int DCAP>
__global__ void __launch_bounds__(256) k_gram_uf(const double* __restrict__ Xus, const double* __restrict__ X,
                                                 const double* __restrict__ Yw, double* __restrict__ Kuf,
                                                 double* __restrict__ bpart, const Params* __restrict__ P,
                                                 int M, int Mp, int D, int64_t N, int d_out, int64_t* stamps,
                                                 int64_t* sweep_begin) {
    __shared__ double us[DCAP * TB];
    TraceScope trace(2);
    if (sweep_begin) {            // first kernel of a sweep whose parameters were already resident (no k_prep_xu in front)
        stamp_enter(sweep_begin + 0 * STAMP_STRIDE);          // SGP_T_SWEEP
        stamp_enter(sweep_begin + 7 * STAMP_STRIDE);          // SGP_T_LOCAL
    }
    stamp_enter(stamps);
    __shared__ double xs[DCAP * TB];
    __shared__ double ys[MAXO * TB];
    __shared__ double red[16 * TB];
    const int I = blockIdx.y * TB;                           // m tile on grid.y, point block on grid.x (no 65535 limit)
    const int64_t n0 = (int64_t)blockIdx.x * TB;
    for (int t = threadIdx.x; t < D * TB; t += 256) {
        int d = t / TB, r = t % TB;
        us[t] = Xus[(size_t)d * Mp + I + r];
    }
    for (int t = threadIdx.x; t < D * TB; t += 256) {       // coalesced AoS read, transposed into SoA
        int p = t / D, d = t % D;
        int64_t n = n0 + p;
        xs[d * TB + p] = (n < N) ? X[(size_t)n * D + d] * P->inv_ell[d] : 0.0;
    }
    for (int t = threadIdx.x; t < d_out * TB; t += 256) {
        int o = t / TB, p = t % TB;
        int64_t n = n0 + p;
        ys[t] = (n < N) ? Yw[(size_t)o * N + n] : 0.0;
    }
    __syncthreads();
    const int tm = threadIdx.x & 15, tn = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int d = 0; d < D; ++d) {
        double u[4], x[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { u[a] = us[d * TB + tm * 4 + a]; x[a] = xs[d * TB + tn * 4 + a]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) { double t = u[a] - x[b]; acc[a][b] = fma(t, t, acc[a][b]); }
    }
    const double s2 = P->sigma2;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        int64_t n = n0 + tn * 4 + b;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int m = I + tm * 4 + a;
            double v = (m < M && n < N) ? s2 * exp(-0.5 * acc[a][b]) : 0.0;
            acc[a][b] = v;
        }
        if (n < N) {
            double* dst = Kuf + (size_t)n * Mp + I + tm * 4;
            *reinterpret_cast<double2*>(dst) = make_double2(acc[0][b], acc[1][b]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(acc[2][b], acc[3][b]);
        }
    }
    // partial B for this block's 64 points: reduce the 16 n-groups through LDS in a fixed order
    for (int o = 0; o < d_out; ++o) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < 4; ++b) s = fma(acc[a][b], ys[o * TB + tn * 4 + b], s);
            red[tn * TB + tm * 4 + a] = s;
        }
        __syncthreads();
        if (threadIdx.x < TB) {
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 16; ++g) s += red[g * TB + threadIdx.x];
            bpart[((size_t)blockIdx.x * d_out + o) * Mp + I + threadIdx.x] = s;
        }
    }
    stamp_exit(stamps);
}

// ------------------------------------------------------------------------------------------------
// MFMA tile routine: acc(64 x 64, 4 waves x (2 x 2) tiles of 16 x 16) += sum_k As[k][i] * Bs[k][j]
// ------------------------------------------------------------------------------------------------
struct Acc4 { d4 t[2][2]; };

__device__ __forceinline__ void acc_zero(Acc4& a) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) a.t[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
}

// kcount must be a multiple of 4.  As/Bs point at LDS panels with row stride PS.
__device__ __forceinline__ void tile_mma(Acc4& acc, const double* As, const double* Bs, int kcount, int lane, int wr, int wc) {
    const int li = lane & 15, lk = lane >> 4;
    const double* ap = As + lk * PS + wr * 32 + li;
    const double* bp = Bs + lk * PS + wc * 32 + li;
#pragma unroll 4
    for (int k = 0; k < kcount; k += 4) {
        double a0 = ap[0], a1 = ap[16];
        double b0 = bp[0], b1 = bp[16];
        acc.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.t[0][0], 0, 0, 0);
        acc.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.t[0][1], 0, 0, 0);
        acc.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.t[1][0], 0, 0, 0);
        acc.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.t[1][1], 0, 0, 0);
        ap += 4 * PS;
        bp += 4 * PS;
    }
}

// tile_mma with a B panel that is stored as it arrives from a matrix whose rows run along the contraction index (K_uf: a point's
// kernel values): Bs[j][k], row stride KMS = 66 doubles.  The kernels that build their B panel from K_uf (k_quadform_fused,
// k_theta_grad_uf) read 32-byte pieces of K_uf rows -- 16 threads per row, thread g the k-entries 4 g .. 4 g + 3 -- and store them
// with two 16-byte LDS stores; a 16-lane MFMA operand read (16 rows at one k) then hits 16 different bank pairs (66 x 2 dwords =
// 4 mod 64).  Rounds 3 / 4a transposed the pieces on the way in (four 8-byte stores with an XOR swizzle): PMC counted 58 % of the
// LDS's active cycles as bank conflicts (profiles/r04_ab_log.txt [21], [25]).
constexpr int KMS = 66;
__device__ __forceinline__ void tile_mma_bk(Acc4& acc, const double* As, const double* Bs, int kcount, int lane, int wr, int wc) {
    const int li = lane & 15, lk = lane >> 4;
    const double* ap = As + lk * PS + wr * 32 + li;
    const double* bp = Bs + (wc * 32 + li) * KMS + lk;
#pragma unroll 4
    for (int k = 0; k < kcount; k += 4) {
        double a0 = ap[0], a1 = ap[16];
        double b0 = bp[0], b1 = bp[16 * KMS];
        acc.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.t[0][0], 0, 0, 0);
        acc.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.t[0][1], 0, 0, 0);
        acc.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.t[1][0], 0, 0, 0);
        acc.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.t[1][1], 0, 0, 0);
        ap += 4 * PS;
        bp += 4;
    }
}

// The same for a DIAGONAL tile of a symmetric product (As and Bs hold the same rows, Bs weighted): only the 10 sub-tiles (R, C),
// R >= C, of the 4 x 4 grid of 16 x 16 blocks are formed, dealt 3 : 2 : 2 : 3 over the waves --
//   wave 0: (0,0) (1,0) (1,1)   wave 1: (2,0) (2,1)   wave 2: (3,0) (3,1)   wave 3: (2,2) (3,2) (3,3)
// (accumulators t[0][0], t[1][0], t[1][1] for waves 0 / 3, t[0][0], t[0][1] for waves 1 / 2): 3 instead of 4 MFMAs per wave and
// k-step, and no flop on the upper halves, which the algorithm does not contain.
__device__ __forceinline__ void tile_mma_diag(Acc4& acc, const double* As, const double* Bs, int kcount, int lane, int wave) {
    const int li = lane & 15, lk = lane >> 4;
    if (wave == 0 || wave == 3) {
        const int base = (wave == 3) ? 32 : 0;
        const double* ap = As + lk * PS + base + li;
        const double* bp = Bs + lk * PS + base + li;
#pragma unroll 4
        for (int k = 0; k < kcount; k += 4) {
            const double a0 = ap[0], a1 = ap[16], b0 = bp[0], b1 = bp[16];
            acc.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.t[0][0], 0, 0, 0);
            acc.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.t[1][0], 0, 0, 0);
            acc.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.t[1][1], 0, 0, 0);
            ap += 4 * PS;
            bp += 4 * PS;
        }
    } else {
        const double* ap = As + lk * PS + 16 * (wave + 1) + li;
        const double* bp = Bs + lk * PS + li;
#pragma unroll 4
        for (int k = 0; k < kcount; k += 4) {
            const double a0 = ap[0], b0 = bp[0], b1 = bp[16];
            acc.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.t[0][0], 0, 0, 0);
            acc.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.t[0][1], 0, 0, 0);
            ap += 4 * PS;
            bp += 4 * PS;
        }
    }
}

// element (i, j) of the 64 x 64 tile owned by (lane, wave, ti, tj, r)
__device__ __forceinline__ int acc_row(int lane, int wr, int ti, int r) { return wr * 32 + ti * 16 + (lane >> 4) + 4 * r; }
__device__ __forceinline__ int acc_col(int lane, int wc, int tj) { return wc * 32 + tj * 16 + (lane & 15); }

// panel[k][i] = G[(row0 + i) + (col0 + k) * ld]  for i < 64, k < kcount   (contiguous along i in memory)
// (all global loads are issued before the first LDS store: one memory latency per panel instead of one per pass)
__device__ __forceinline__ void load_panel_n(double* panel, const double* __restrict__ G, size_t ld, int row0, int col0,
                                             int kcount, int tid) {
    if (kcount == TB) {
        double2 v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int t = tid + 256 * u, k = t >> 4, rq = t & 15;
            const double* src = G + (size_t)(col0 + k) * ld + row0 + rq * 4;
            v0[u] = *reinterpret_cast<const double2*>(src);
            v1[u] = *reinterpret_cast<const double2*>(src + 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int t = tid + 256 * u, k = t >> 4, rq = t & 15;
            double* dst = panel + k * PS + rq * 4;
            dst[0] = v0[u].x; dst[1] = v0[u].y; dst[2] = v1[u].x; dst[3] = v1[u].y;
        }
        return;
    }
    for (int t = tid; t < kcount * 16; t += 256) {
        int k = t >> 4, rq = t & 15;
        const double* src = G + (size_t)(col0 + k) * ld + row0 + rq * 4;
        double2 v0 = *reinterpret_cast<const double2*>(src);
        double2 v1 = *reinterpret_cast<const double2*>(src + 2);
        double* dst = panel + k * PS + rq * 4;
        dst[0] = v0.x; dst[1] = v0.y; dst[2] = v1.x; dst[3] = v1.y;
    }
}
// panel[k][i] = G[(row0 + k) + (col0 + i) * ld]  (contiguous along k in memory: transposing load)
__device__ __forceinline__ void load_panel_t(double* panel, const double* __restrict__ G, size_t ld, int row0, int col0,
                                             int kcount, int tid) {
    if (kcount == TB) {                         // (all global loads before the first LDS store: one memory latency, not four)
        double2 v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, g = t & 15, i = t >> 4;
            const double* src = G + (size_t)(col0 + i) * ld + row0 + g * 4;
            v0[u] = *reinterpret_cast<const double2*>(src);
            v1[u] = *reinterpret_cast<const double2*>(src + 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, g = t & 15, i = t >> 4;
            panel[(g * 4 + 0) * PS + i] = v0[u].x;
            panel[(g * 4 + 1) * PS + i] = v0[u].y;
            panel[(g * 4 + 2) * PS + i] = v1[u].x;
            panel[(g * 4 + 3) * PS + i] = v1[u].y;
        }
        return;
    }
    const int kq = kcount >> 2;                 // groups of 4 consecutive k
    for (int t = tid; t < kq * 64; t += 256) {
        int g = t % kq, i = t / kq;
        const double* src = G + (size_t)(col0 + i) * ld + row0 + g * 4;
        double2 v0 = *reinterpret_cast<const double2*>(src);
        double2 v1 = *reinterpret_cast<const double2*>(src + 2);
        panel[(g * 4 + 0) * PS + i] = v0.x;
        panel[(g * 4 + 1) * PS + i] = v0.y;
        panel[(g * 4 + 2) * PS + i] = v1.x;
        panel[(g * 4 + 3) * PS + i] = v1.y;
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming SYRK over the points: slab[c][t] (64 x 64, stored [i][j] row-major) =
//     sum_{n in chunk c} K[I+i, n] * omega_n * K[J+j, n]      for the lower tile t = (I, J), I >= J.
// Split over the point axis so that >= 1 block per CU exists even for M = 512 (36 tiles);
// partial slabs are summed in a fixed order by k_assemble (bitwise reproducible, no atomics).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_from_index(int t, int& I, int& J) {
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    I = i;
    J = t - i * (i + 1) / 2;
}

// One work item of the streaming SYRK: the 64 x 64 tile (I, J) summed over the points of chunk `chunk_id`, into `out` ([i][j]
// row-major; a diagonal tile as the full symmetric tile).
template <bool DIAG>
__device__ __forceinline__ void syrk_item(const double* __restrict__ Kuf, const double* __restrict__ omega, double* __restrict__ out,
                                          double* lds, int Mp, int64_t N, int I, int J, int chunk_id, int chunk, bool write_through) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int64_t nbeg = (int64_t)chunk_id * chunk;
    int64_t nend = nbeg + chunk;
    if (nend > N) nend = N;
    const int stages = nend > nbeg ? (int)((nend - nbeg + KB - 1) / KB) : 0;
    // staging map: thread -> (point p = tid >> 4, row quad rq = tid & 15): 32 B of one K_uf column
    const int p = tid >> 4, rq = tid & 15;
    Acc4 acc;
    acc_zero(acc);
    // (gload only REQUESTS stage s + 1; the values are first touched in lstore, behind stage s's products: the weighting by omega
    // used to sit in gload, and the s_waitcnt vmcnt(0) it needs then lands in FRONT of the MFMAs -- every stage sat out its whole
    // memory latency before its products)
    double2 ra[2], rb[2];
    double rw = 1.0;
    const bool offdiag = DIAG ? false : (I != J);         // (a diagonal tile reads its rows once)
    auto gload = [&](int s) {
        int64_t n = nbeg + (int64_t)s * KB + p;
        ra[0] = ra[1] = rb[0] = rb[1] = make_double2(0.0, 0.0);
        if (n < nend) {
            const double* src = Kuf + (size_t)n * Mp + I * TB + rq * 4;
            ra[0] = *reinterpret_cast<const double2*>(src);
            ra[1] = *reinterpret_cast<const double2*>(src + 2);
            if (omega) rw = omega[n];
            if (offdiag) {
                const double* sb = Kuf + (size_t)n * Mp + J * TB + rq * 4;
                rb[0] = *reinterpret_cast<const double2*>(sb);
                rb[1] = *reinterpret_cast<const double2*>(sb + 2);
            }
        }
    };
    auto lstore = [&](int buf) {
        double* A = lds + buf * (2 * KB * PS) + p * PS + rq * 4;
        double* B = A + KB * PS;
        double2 b0 = offdiag ? rb[0] : ra[0], b1 = offdiag ? rb[1] : ra[1];
        if (omega) { b0.x *= rw; b0.y *= rw; b1.x *= rw; b1.y *= rw; }
        *reinterpret_cast<double2*>(A) = ra[0];
        *reinterpret_cast<double2*>(A + 2) = ra[1];
        *reinterpret_cast<double2*>(B) = b0;
        *reinterpret_cast<double2*>(B + 2) = b1;
    };
    if (stages > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        if (s + 1 < stages) gload(s + 1);
        const double* A = lds + buf * (2 * KB * PS);
        if constexpr (DIAG) tile_mma_diag(acc, A, A + KB * PS, KB, lane, wave);
        else tile_mma(acc, A, A + KB * PS, KB, lane, wr, wc);
        if (s + 1 < stages) lstore(buf ^ 1);
        __syncthreads();
    }
#ifdef SGP_EXP_QUARTER_SLABS      // timing experiment only (results are wrong): what would a quarter of the slab traffic buy?
    if (chunk_id & 3) return;
#endif
    if constexpr (!DIAG) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // write_through (A/B switch SGP_SYRK_WT, default off): agent-scope stores, past this XCD's write-back L2, so
                    // that the chain steps running beside this launch do not flush its slab lines at their kernel boundaries.
                    // Measured at T, 4 alternations x 1000 sweeps on one box: 4118-4135 sweeps/s in all three modes -- no effect
                    // on the overlapped sweep; alone, the single plain-order launch is slower with it (63.6 vs 59.5 us).
                    double* dst = out + acc_row(lane, wr, ti, r) * TB + acc_col(lane, wc, tj);
                    if (write_through)
                        __hip_atomic_store((__attribute__((address_space(1))) double*)dst, acc.t[ti][tj][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        *dst = acc.t[ti][tj][r];
                }
    } else {
        // the wave's sub-tiles (tile_mma_diag), each also at its mirror position: the slab is the full symmetric tile.
        // Accumulator slot s of the wave holds sub-tile (R0 + (s > 0), C0 + (s > 1)) for waves 0 / 3, (wave + 1, s) for waves 1 / 2
        const int row = lane >> 4, col = lane & 15;
        const bool corner = (wave == 0 || wave == 3);
        const int b = (wave == 3) ? 2 : 0;
#pragma unroll
        for (int slot = 0; slot < 3; ++slot) {
            if (!corner && slot == 2) break;
            const d4 v = corner ? (slot == 0 ? acc.t[0][0] : (slot == 1 ? acc.t[1][0] : acc.t[1][1])) : (slot == 0 ? acc.t[0][0] : acc.t[0][1]);
            const int R = corner ? b + (slot > 0) : wave + 1, C = corner ? b + (slot > 1) : slot;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[(16 * R + row + 4 * r) * TB + 16 * C + col] = v[r];
                if (R != C) out[(16 * C + col) * TB + 16 * R + row + 4 * r] = v[r];
            }
        }
    }
}

// One launch covers the lower tiles of the tile rows [row_lo, row_lo + nrows) of Psi2 (a whole matrix: row_lo = 0, nrows = T; the
// overlapped sweep launches the rows in groups, last rows first, see sgp_api.hip): tiles [tile0, tile0 + ntiles) of the row-major
// triangle, the point axis split into `nchunks` chunks of `chunk` points.  `slabs` is the launch's own slab area
// [nchunks][ntiles] of 64 x 64.  Diagonal tiles form 10 of their 16 sub-tiles (tile_mma_diag): no flop on the halves the
// algorithm does not contain.  (Their workgroups finish a quarter earlier; giving them 4/3 longer chunks to rebalance the round
// was measured -- 30 x 336 points off the diagonal, 23 x 448 on it -- and lost, 69 vs 61 us: the items of one point range then
// no longer sit next to each other in the XCD map below, and the diagonal items' K_uf reads miss the L2 their chunk-mates filled.)
struct SyrkGeom {
    int row_lo, nrows;              // tile rows
    int tile0, ntiles;              // their lower tiles
    int chunk, nchunks;             // split of the point axis
    int write_through;              // slabs stored past the L2 (launches that run beside the factorisation chains)
    int wide;                       // 1: launched as k_syrk_direct (one 512-thread workgroup per CU, no LDS staging; `chunk` a multiple of 4 points)
};
__global__ void __launch_bounds__(256) k_syrk_stream(const double* __restrict__ Kuf, const double* __restrict__ omega,
                                                     double* __restrict__ slabs, int Mp, int64_t N, SyrkGeom g, int64_t* stamps,
                                                     long long* gate, long long gate_value) {
    __shared__ double lds[2 * 2 * KB * PS];           // [buf][panel A|B][KB][PS]
    TraceScope trace(64 + g.tile0);
    stamp_enter(stamps);
    // the last workgroup of this launch's single resident round is on a CU: whoever waited for that (the K_uu chain's first
    // kernel) may take the CUs that are left
    if (gate && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(gate, gate_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // XCD-aware block -> (tile, chunk) map: workgroups are dealt round-robin over the 8 XCDs, so ids congruent mod 8
    // share an L2.  All tiles of one point-chunk read the same K_uf columns: the work items, ordered chunk-major, are cut
    // into 8 contiguous runs, one per XCD, so that a chunk's columns are fetched into (at most two) L2s once.  Needs the
    // item count to be a multiple of 8 (the host picks nchunks accordingly whenever there is enough work; speed only).
    const int nitems = g.ntiles * g.nchunks;
    int item = blockIdx.x;
    if ((nitems & 7) == 0) item = (blockIdx.x & 7) * (nitems >> 3) + (blockIdx.x >> 3);
    const int chunk_id = item / g.ntiles, tile_id = item % g.ntiles;
    int I, J;
    tile_from_index(g.tile0 + tile_id, I, J);
    const int chunk = g.chunk;
    const size_t slab = (size_t)chunk_id * g.ntiles + tile_id;
    double* out = slabs + slab * (TB * TB);
#ifdef SGP_SYRK_DIAG_SKIP
    // two specialisations of the whole stage loop: with the choice inside the loop the kernel needed 148 VGPRs (2 waves per
    // SIMD, i.e. two workgroups per CU and a second round: 95 instead of 58 us)
    if (I == J) syrk_item<true>(Kuf, omega, out, lds, Mp, N, I, J, chunk_id, chunk, g.write_through != 0);
    else syrk_item<false>(Kuf, omega, out, lds, Mp, N, I, J, chunk_id, chunk, g.write_through != 0);
#else
    syrk_item<false>(Kuf, omega, out, lds, Mp, N, I, J, chunk_id, chunk, g.write_through != 0);
#endif
    stamp_exit(stamps);
}

// ------------------------------------------------------------------------------------------------
// The same work item WITHOUT LDS staging (round 4, `wide` launches: wherever the SYRK fills the chip).  Every wave owns the WHOLE
// 64 x 64 tile over its share of the item's points -- 16 accumulators of v_mfma_f64_16x16x4_f64 -- and loads its operands from
// global memory straight into the registers the matrix instruction reads: lane (lk, li) of operand block u supplies
// K_uf[n0 + lk][64 I + 16 u + li], i.e. 16 lanes = one 128-byte line of a K_uf row, SYRK_P k-steps (of 4 points) ahead.  No
// barrier, no LDS traffic and no meeting of waves inside the loop: a wave waits for nothing but its own loads, and those are
// SYRK_P - 1 steps old when it needs them.  The eight waves of a workgroup (two per SIMD: one fills the other's issue gaps) take the
// chunk's k-steps round-robin, add their partial tiles up in LDS at the end in fixed order (two rounds of four tiles: 128 KB) and
// ONE slab leaves the CU, as with round 4's first form of this launch (k_syrk_stream16: four LDS-staged wave groups per CU, each
// meeting on an LDS counter per 16-point stage).
// Why: tools/dpp_f64_probe.hip showed that LDS-fed v_mfma_f64 attains 63 / 72 / 73 TFLOP/s at 1 / 2 / 4 waves per SIMD on this
// chip -- rounds 1-3 had priced the matrix pipe at 46 - 48 from a probe whose loop the compiler had filled with accumulator
// copies -- while k_syrk_stream16 ran at 49 (N = 10^6) and 38 (T): timing variants without its group meetings / global loads / LDS
// stores gained 7 % each and 20 % together (gpurun_out/r4r), and what was left was still the staging structure.  This kernel:
// 71 TFLOP/s executed at N = 10^6 (4.14 instead of 5.36 ms), T's group launches 38.7 / 36.9 instead of 44.3 / 42.5 us
// (tools/syrk_direct_probe.hip, profiles/r04_ab_log.txt [14]-[16]).
// The k-step loop has NO branch inside: the compiler's s_waitcnt placement is exact only for a straight-line round (with a
// conditional product in it, it drained all loads at the loop header); loads past the wave's last k-step are clamped to that step
// (harmless re-reads), the last nt mod SYRK_P products follow behind the loop, a ragged last k-step (chunk end not a multiple of
// 4) is formed by the wave whose turn it is, unpipelined, with zeros for the missing points.
// ------------------------------------------------------------------------------------------------
constexpr int SYRK_WAVES = 8;
constexpr int SYRK_P = 5;               // k-steps in flight (per-point weights: 4 -- the weights and the weighted operands need the registers)
constexpr int SYRK_DIRECT_THREADS = 64 * SYRK_WAVES;
template <bool DIAG, bool WEIGHTED>
__device__ __forceinline__ void syrk_direct_stream(d4 (&acc)[4][4], const double* __restrict__ pa, const double* __restrict__ pb,
                                                   const double* __restrict__ pw, size_t step, int nt) {
    constexpr int P = WEIGHTED ? 4 : SYRK_P;
    double a[P][4], b[P][4], w[P];
    auto load = [&](int p, int t) {
        const double* qa = pa + (size_t)t * step;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[p][u] = qa[16 * u];
        if constexpr (!DIAG) {
            const double* qb = pb + (size_t)t * step;
#pragma unroll
            for (int u = 0; u < 4; ++u) b[p][u] = qb[16 * u];
        }
        if constexpr (WEIGHTED) w[p] = pw[(size_t)t * 4 * SYRK_WAVES];
    };
    auto mma = [&](int p) {
        double bw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bw[u] = DIAG ? a[p][u] : b[p][u];
            if constexpr (WEIGHTED) bw[u] *= w[p];
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p][ti], bw[tj], acc[ti][tj], 0, 0, 0);
    };
    if (nt <= 0) return;
#pragma unroll
    for (int p = 0; p < P; ++p) load(p, p < nt ? p : nt - 1);
    int t0 = 0;
    for (; t0 + P <= nt; t0 += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            __builtin_amdgcn_sched_barrier(0);        // (products and loads stay in the order written: see quadform_stream)
            mma(p);
            __builtin_amdgcn_sched_barrier(0);
            const int tn = t0 + p + P;
            load(p, tn < nt ? tn : nt - 1);
        }
    }
#pragma unroll
    for (int p = 0; p < P - 1; ++p)
        if (t0 + p < nt) mma(p);
}

__global__ void __launch_bounds__(SYRK_DIRECT_THREADS) k_syrk_direct(const double* __restrict__ Kuf, const double* __restrict__ omega,
                                                                     double* __restrict__ slabs, int Mp, int64_t N, SyrkGeom g,
                                                                     int64_t* stamps, long long* gate, long long gate_value) {
    __shared__ __attribute__((aligned(16))) double lds[4 * TB * TB];
    TraceScope trace(64 + g.tile0);
    stamp_enter(stamps);
    // XCD-aware block -> item map for ANY item count: the items, ordered chunk-major, are cut into 8 runs of `per` items, one per
    // XCD (workgroups are dealt round-robin over the XCDs); the grid has 8 * per blocks, the surplus ones leave
    const int nitems = g.ntiles * g.nchunks, per = (nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    // the last workgroup of this launch's single resident round is on a CU: whoever waited for that (the K_uu chain's first
    // kernel) may take the CUs that are left
    if (gate && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(gate, gate_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (item >= nitems) return;                       // (whole workgroup: no barrier is left behind)
    const int chunk_id = item / g.ntiles, tile_id = item % g.ntiles;
    int I, J;
    tile_from_index(g.tile0 + tile_id, I, J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int64_t cbeg = (int64_t)chunk_id * g.chunk;
    int64_t cend = cbeg + g.chunk;
    if (cend > N) cend = N;
    const int nfull = cend > cbeg ? (int)((cend - cbeg) >> 2) : 0;                          // whole k-steps of the chunk
    const int nt = nfull > wave ? (nfull - wave + SYRK_WAVES - 1) / SYRK_WAVES : 0;          // this wave's: wave, wave + 8, ...
    d4 acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
    {
        const int64_t n0 = cbeg + 4 * wave + lk;        // the lane's point of the wave's first k-step
        const double* pa = Kuf + (size_t)n0 * Mp + I * TB + li;
        const double* pb = Kuf + (size_t)n0 * Mp + J * TB + li;
        const double* pw = omega ? omega + n0 : nullptr;
        const size_t step = (size_t)4 * SYRK_WAVES * Mp;
        if (I == J) {
            if (omega) syrk_direct_stream<true, true>(acc, pa, pb, pw, step, nt);
            else syrk_direct_stream<true, false>(acc, pa, pb, pw, step, nt);
        } else {
            if (omega) syrk_direct_stream<false, true>(acc, pa, pb, pw, step, nt);
            else syrk_direct_stream<false, false>(acc, pa, pb, pw, step, nt);
        }
    }
    if (((cend - cbeg) & 3) != 0 && cend > cbeg && wave == (nfull % SYRK_WAVES)) {          // the ragged last k-step
        const int64_t n = cbeg + 4 * (int64_t)nfull + lk;
        const bool in = n < cend;
        const double* qa = Kuf + (size_t)(in ? n : cend - 1) * Mp + I * TB + li;
        const double* qb = Kuf + (size_t)(in ? n : cend - 1) * Mp + J * TB + li;
        const double wn = in ? (omega ? omega[n] : 1.0) : 0.0;
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { av[u] = in ? qa[16 * u] : 0.0; bv[u] = qb[16 * u] * wn; }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bv[tj], acc[ti][tj], 0, 0, 0);
    }
    // the eight partial tiles meet in LDS ([i][j], 64 doubles per row), four at a time, summed in wave order; thread t owns the
    // entries 2 (t + 512 e), 2 (t + 512 e) + 1 of the slab (16-byte stores, 1 KB runs per wave)
    constexpr int NE = (TB * TB) / SYRK_DIRECT_THREADS / 2;
    double2 out[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) out[e] = make_double2(0.0, 0.0);
#pragma unroll
    for (int round = 0; round < SYRK_WAVES / 4; ++round) {
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
        for (int e = 0; e < NE; ++e) {
            const int idx = 2 * (tid + SYRK_DIRECT_THREADS * e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 x = *reinterpret_cast<const double2*>(lds + q * (TB * TB) + idx);
                out[e].x += x.x; out[e].y += x.y;
            }
        }
    }
    double* dst = slabs + ((size_t)chunk_id * g.ntiles + tile_id) * (TB * TB);
#pragma unroll
    for (int e = 0; e < NE; ++e) *reinterpret_cast<double2*>(dst + 2 * (tid + SYRK_DIRECT_THREADS * e)) = out[e];
    stamp_exit(stamps);
}

// ------------------------------------------------------------------------------------------------
// Sum the SYRK slabs and the B partials into the packed statistics buffer (the all-reduce payload):
//   stats = [Psi2 (Mp x Mp, full symmetric) | B (Mp x d_out) | scalars]
// ------------------------------------------------------------------------------------------------
// B = sum of the per-block partials of k_gram_uf: workgroup `bid` of `nb` takes the (row block, output) pairs bid, bid + nb, ...;
// 4 threads per entry -- one per wave, so that a wave reads 512-byte runs -- walk the partials with a stride of 4, up to 32
// independent loads in flight (round 3 had 16: three dependent round trips for the 157 partial rows of N = 10 000, the longest
// thing k_assemble did), and are combined in a fixed order.  The four waves meet through a few hundred bytes of GLOBAL scratch
// (`bscratch`, written, workgroup barrier, read back by wave 0 on the same CU), not through LDS: any LDS in a launch that runs
// beside a SYRK keeps a fourth SYRK workgroup off every CU that still holds one of its blocks, and the next group's SYRK then
// needs a second round (measured: +15 us on the statistics).  (Four adjacent lanes per entry and shuffles instead: 430 instead
// of 180 us for the 15 625 partial rows of N = 10^6.)
__device__ __forceinline__ void sum_b_pairs(const double* __restrict__ bpart, double* __restrict__ B, double* __restrict__ bscratch,
                                            int Mp, int T, int nblk, int d_out, int bid, int nb) {
    const int tid = threadIdx.x, m = tid & 63, part = tid >> 6;
    for (int pair = bid; pair < T * d_out; pair += nb) {
        const int Ib = pair % T, o = pair / T;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        int b = part;
        for (; b + 124 < nblk; b += 128) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = bpart[((size_t)(b + 4 * u) * d_out + o) * Mp + Ib * TB + m];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc[u & 3] += v[u];
        }
        for (; b + 28 < nblk; b += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = bpart[((size_t)(b + 4 * u) * d_out + o) * Mp + Ib * TB + m];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] += v[u];
        }
        for (; b < nblk; b += 4) acc[0] += bpart[((size_t)b * d_out + o) * Mp + Ib * TB + m];
        double* red = bscratch + (size_t)pair * (4 * TB);
        red[part * TB + m] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();                                     // (waits for the stores: a workgroup-scope release)
        if (part == 0) B[(size_t)o * Mp + Ib * TB + m] = (red[m] + red[TB + m]) + (red[2 * TB + m] + red[3 * TB + m]);
    }
}
__global__ void __launch_bounds__(256) k_assemble(const double* __restrict__ slabs, const double* __restrict__ bpart,
                                                  const double* __restrict__ data_scalars, double* __restrict__ stats,
                                                  int Mp, int T, SyrkGeom g, int nblk, int d_out,
                                                  int nscal, int do_b, int64_t* stamps, int* __restrict__ info_reset,
                                                  long long* start_word, long long start_value, int packed,
                                                  double* __restrict__ bscratch) {
    // grid (rows, T + do_b, 4 or 16): blocks (x, y < T, z) sum rows [64 z / grid.z, ...) of the slab tile (I, J) = (row_lo + x, y),
    // I >= J -- four entries per thread and 48 loads in flight (grid.z = 4: few chunks) or one entry per thread (grid.z = 16: many
    // chunks), see below -- and write both mirror images.
    // `slabs` / `g`: the slab area and geometry of this launch's tile rows (k_syrk_stream).  Blocks with y == T (do_b) sum the
    // B partials and copy the data scalars.
    // packed (data-sharded sweeps): `stats` is the exchange buffer [lower tiles, row-major triangle, 64 x 64 column-major each |
    // B | scalars] -- what the ranks sum-all-reduce (1.18 MB at M = 512 instead of the 2.10 MB of the full symmetric matrix);
    // k_unpack_stats expands the reduced buffer into the layout the rest of the sweep reads.
    // NO LDS on purpose: in the overlapped sweep this kernel runs while the NEXT group's SYRK already holds every byte of LDS on
    // its CUs; a block that needs none fits beside those workgroups.  (Adjacent LANES down a column -- 32-byte runs both ways -- made
    // the loads irregular across the wave and the kernel twice as slow in round 3, 15 instead of 7.7 us; the layout below keeps the
    // lanes along a row and gives each THREAD four rows.)
    // (Publishing the results to a kernel that is ALREADY running on another stream from inside this one was tried -- a counter
    // bumped by every block: with __threadfence() each block writes the whole L2 back, which the next group's SYRK keeps filling
    // with dirty slab lines (70 us for this kernel instead of 5); with write-through stores 13 us for 21 tiles.  The kernel
    // boundary does it once: k_join_set behind this launch sets the group's word.)
    TraceScope trace(128 + g.row_lo);
    const int row_lo = g.row_lo;
    // start_word (may be nullptr): this launch has started, i.e. the SYRK in front of it on this stream has drained -- the masked
    // statistics stream lets its next SYRK onto the chip at that moment (k_join_wait), not earlier (it would share the CUs with
    // the SYRK this launch sums up) and not much later (the CUs it is sized for would be taken by the chains' workgroups)
    if (start_word && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(start_word, start_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int I = row_lo + blockIdx.x, J = blockIdx.y, z = blockIdx.z;
    const int tid = threadIdx.x;
    if (J < T && I >= J && gridDim.z == 16) {
        // MANY chunks (small problems: a few tiles, the point axis cut into hundreds of 16-point chunks): one entry per thread, sixteen
        // blocks per tile -- the chunk loop is the long part and wants as many threads as there are entries.  Thread = (row 4 z +
        // (tid >> 6), column j = tid & 63); the column side goes out as scattered 8-byte stores (a few KB here).
        const int il = tid >> 6, j = tid & 63;
        const int t = I * (I + 1) / 2 + J - g.tile0;
        const double* base = slabs + (size_t)t * (TB * TB) + (z * 4 + il) * TB + j;
        const size_t cstride = (size_t)g.ntiles * (TB * TB);
        const int nchunks = g.nchunks;
        double s = 0.0;
        int c = 0;
        for (; c + 24 <= nchunks; c += 24) {                 // 24 loads in flight; fixed summation order: chunk 0, 1, 2, ...
            double v[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) v[u] = base[(size_t)(c + u) * cstride];
#pragma unroll
            for (int u = 0; u < 24; ++u) s += v[u];
        }
        for (; c + 4 <= nchunks; c += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = base[(size_t)(c + u) * cstride];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; c < nchunks; ++c) s += base[(size_t)c * cstride];
        if (packed) stats[(size_t)(I * (I + 1) / 2 + J) * (TB * TB) + j * TB + z * 4 + il] = s;
        else {
            stats[(size_t)(J * TB + j) * Mp + I * TB + z * 4 + il] = s;
            if (I != J) stats[(size_t)(I * TB + z * 4 + il) * Mp + J * TB + j] = s;
        }
    } else if (J < T && I >= J) {
        // FEW chunks (grid.z == 4; the sweeps whose SYRK fills the chip: 12 chunks per tile with the 16-wave SYRK).
        // thread = (column j = tid & 63, FOUR consecutive rows 4 zr .. 4 zr + 3, zr = 4 z + (tid >> 6)): a wave reads one 512-byte
        // slab row per (row, chunk) -- coalesced, every load of a batch independent -- and owns 4 x 64 entries whose images in the
        // statistics are 32-byte runs down a column (two 16-byte stores per thread) and, mirrored, 512-byte runs along a row.
        // (Round 3 had one entry per thread: the column side went out as 8-byte stores scattered over 64 lines per instruction.)
        const int zr = 4 * z + (tid >> 6), j = tid & 63;
        const int t = I * (I + 1) / 2 + J - g.tile0;
        const double* base = slabs + (size_t)t * (TB * TB) + (size_t)(4 * zr) * TB + j;
#ifdef SGP_EXP_QUARTER_SLABS
        const size_t cstride = (size_t)g.ntiles * (TB * TB) * 4;
        const int nchunks = (g.nchunks + 3) / 4;
#else
        const size_t cstride = (size_t)g.ntiles * (TB * TB);
        const int nchunks = g.nchunks;
#endif
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        int c = 0;
        for (; c + 12 <= nchunks; c += 12) {                 // 48 loads in flight; fixed summation order: chunk 0, 1, 2, ...
            double v[12][4];
#pragma unroll
            for (int u = 0; u < 12; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[u][r] = base[(size_t)(c + u) * cstride + r * TB];
#pragma unroll
            for (int u = 0; u < 12; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] += v[u][r];
        }
        for (; c + 4 <= nchunks; c += 4) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[u][r] = base[(size_t)(c + u) * cstride + r * TB];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] += v[u][r];
        }
        for (; c < nchunks; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += base[(size_t)c * cstride + r * TB];
        double* col = packed ? stats + (size_t)(I * (I + 1) / 2 + J) * (TB * TB) + (size_t)j * TB + 4 * zr
                             : stats + (size_t)(J * TB + j) * Mp + I * TB + 4 * zr;
        *reinterpret_cast<double2*>(col) = make_double2(s[0], s[1]);
        *reinterpret_cast<double2*>(col + 2) = make_double2(s[2], s[3]);
        if (!packed && I != J)
#pragma unroll
            for (int r = 0; r < 4; ++r) stats[(size_t)(I * TB + 4 * zr + r) * Mp + J * TB + j] = s[r];
    }
    // B = sum of the per-block partials (sum_b_pairs), by the blocks of the extra grid row.  (Two ways of taking this off the path in
    // front of the Lambda chain were built and measured in round 4 -- B summed on the masked stream ahead of this launch, and B
    // summed by extra workgroups of the chain's step 0 -- and neither paid: profiles/r04_ab_log.txt [8], [12].)
    if (do_b && J == T) {
        double* B = stats + (packed ? (size_t)(T * (T + 1) / 2) * (TB * TB) : (size_t)Mp * Mp);
        const int bid = blockIdx.x * gridDim.z + z, nb = gridDim.x * gridDim.z;
        sum_b_pairs(bpart, B, bscratch, Mp, T, nblk, d_out, bid, nb);
        if (bid == 0)
            for (int e = tid; e < nscal; e += 256) B[(size_t)Mp * d_out + e] = data_scalars[e];
    }
    if (info_reset && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *info_reset = 0;
    stamp_exit(stamps);
}

// the reduced exchange buffer (k_assemble, packed) -> the packed statistics layout [Psi2 full symmetric | B | scalars]
__global__ void __launch_bounds__(256) k_unpack_stats(const double* __restrict__ pack, double* __restrict__ stats, int Mp, int T,
                                                      int tail, int tile0, int count) {
    // grid (count + 1): block b < count copies lower tile tile0 + b to both mirror positions; the last block copies B and the
    // scalars (tail > 0: the launch that carries them -- the whole buffer, or the first group of an overlapped data-sharded sweep)
    const int ntiles = T * (T + 1) / 2, tid = threadIdx.x;
    if ((int)blockIdx.x == count) {
        for (int e = tid; e < tail; e += 256) stats[(size_t)Mp * Mp + e] = pack[(size_t)ntiles * (TB * TB) + e];
        return;
    }
    const int t = tile0 + (int)blockIdx.x;
    int I, J;
    tile_from_index(t, I, J);
    __shared__ double tile[TB * LT];
    const double* src = pack + (size_t)t * (TB * TB);
    for (int e = tid; e < TB * TB; e += 256) {
        const int j = e >> 6, i = e & 63;                    // element (i, j) of the tile: column-major in the exchange buffer
        const double v = src[e];
        stats[(size_t)(J * TB + j) * Mp + I * TB + i] = v;
        tile[i * LT + j] = v;
    }
    if (I == J) return;
    __syncthreads();
    for (int e = tid; e < TB * TB; e += 256) {
        const int i = e >> 6, j = e & 63;                    // the mirror image: 64 consecutive j per row i
        stats[(size_t)(I * TB + i) * Mp + J * TB + j] = tile[i * LT + j];
    }
}

// ------------------------------------------------------------------------------------------------
// Lambda = Lambda0 + W (x) Psi2 ,  xi = xi0 + vec(B W)   (GPnode/UniSGPnode.jl:62-63 summed over the
// N messages of :144-173; MultiSGP: GPnode/MultiSGPnode.jl:306-307).  Q = d_out * M, padded to Qp with I.
// prior_form: 1 = dense precision Lambda0 (Qp x Qp) + xi0, 2 = isotropic precision P->prior_iso, xi0 = 0.
// ------------------------------------------------------------------------------------------------
// (Lam may be Lambda0 and xi may be xi0 when rev == 0: every entry is read and written by the same thread -- the posterior carry
// updates the prior in place)
__global__ void __launch_bounds__(256) k_form_lambda(const double* __restrict__ stats, const double* Lambda0,
                                                     const double* xi0, double* Lam, double* xi, const Params* __restrict__ P,
                                                     int M, int Mp, int d_out, int Q, int Qp, int prior_form, int rev, int64_t* stamps,
                                                     int* __restrict__ info_reset) {
    stamp_enter(stamps);
    if (info_reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *info_reset = 0;
    const int gi = blockIdx.x * TB + (threadIdx.x & 63);
    const int jg = threadIdx.x >> 6;
    const double* Psi2 = stats;
    const double* B = stats + (size_t)Mp * Mp;
    for (int jj = 0; jj < 16; ++jj) {
        int gj = blockIdx.y * TB + jg * 16 + jj;
        double v;
        if (gi < Q && gj < Q) {
            int a = gi / M, i = gi % M, b = gj / M, j = gj % M;
            double prior = (prior_form == 1) ? Lambda0[(size_t)gj * Qp + gi] : (gi == gj ? P->prior_iso : 0.0);
            v = prior + P->W[a + b * d_out] * Psi2[(size_t)j * Mp + i];
        } else v = (gi == gj) ? 1.0 : 0.0;
        // rev: write P Lambda P (index reversal).  Its lower Cholesky factor L' gives chol(Sigma_v).U = P L'^-1 P for free,
        // from which Uv follows by a rank-1 update instead of a third factorisation (see k_cholupdate).
        if (rev) Lam[(size_t)(Qp - 1 - gj) * Qp + (Qp - 1 - gi)] = v;
        else Lam[(size_t)gj * Qp + gi] = v;
    }
    if (blockIdx.y == 0 && threadIdx.x < TB) {
        double v = 0.0;
        if (gi < Q) {
            int a = gi / M, i = gi % M;
            v = (prior_form == 1) ? xi0[gi] : 0.0;
            for (int e = 0; e < d_out; ++e) v = fma(B[(size_t)e * Mp + i], P->W[e + a * d_out], v);
        }
        xi[gi] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky (lower), tile 64, ONE launch per block step (look-ahead by redundancy):
//   k_potrf_step(j): block (a, b) of the trailing matrix first applies the rank-64 update of step j-1
//   (A_ik -= L_i,j-1 L_k,j-1^T on the matrix cores).  Blocks of the first trailing column (b = 0) then
//   continue with step j without leaving the kernel: each of them ALSO factors the diagonal tile A_jj itself in
//   LDS (redundant compute instead of an inter-block hand-off; its rank-64 update arrives as a tile formed by
//   the previous launch, see syrk_slice) and solves X L_jj^T = A_ij for its own tile; the block on the
//   diagonal writes L_jj (and, with Winv, L_jj^-1).
// Thread layout of the sequential parts: the triangular solves give 4 adjacent lanes (q = tid & 3) one row
// r = tid >> 2, lane q keeping the row's entries c = q (mod 4) in registers (no reductions; the quad exchanges
// the pivot entry with a DPP quad broadcast); the pivot runs of the factorisation use lane = 16 q + r over the
// full symmetric 16 x 16 block (DPP row broadcast + one ds_bpermute per pivot, see potf2_tile).  Two workgroup
// barriers per 16 pivots in the factorisation; the panel solve's 16 x 16 blocks (trsm_update / trsm_solve) are wave-local and
// placed in those eight intervals where their wave's SIMD is free of factoring work.
// `info` receives (global column + 1) of the first non-positive pivot (LAPACK convention).
// ------------------------------------------------------------------------------------------------

template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// lane N of each 16-lane DPP row to the whole row (row_newbcast)
template <int N>
__device__ __forceinline__ double row_bcast(double v) {
    constexpr int ctrl = 0x150 + N;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// 1/sqrt(d): v_rsq_f64 seed (measured ~2^-25 relative on gfx950) + two Newton steps (rounding-limited).  Off the pivot
// chain: potf2_tile evaluates it once per 16 pivots, for all of them in parallel.
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// One 16-pivot triangular solve  x D^T = b  for the lane's row (4 lanes per row, lane q holds columns q, q + 4, ...), in the
// scaled form z_c = x_c / D_cc:  z_c <- (b_c / D_cc) - sum_{k < c} z_k (D_ck / D_cc), so that a pivot is one quad broadcast
// and the FMAs -- no multiply by the reciprocal and no select on its dependent chain.  Dp (LDS, DPS doubles per row) is the
// block prepared by potf2_tile: Dp[c][k] = D_ck / D_cc below the diagonal, ZERO on and above it (so the reads need no
// masks: finished columns are left alone); ri = 1 / diag(D).  All operands are fetched into registers before the loop: with
// the reads inside it the compiler exec-masks each of them (~35 instructions and a wait per pivot, the loop is bound by
// instruction issue).  On exit x holds the solution.
constexpr int DPS = 17;             // row stride of a prepared 16 x 16 block (odd: the four rows a quad reads differ in bank)
constexpr int DPB = 16 * DPS;       // doubles per block
template <class F>
__device__ __forceinline__ void solve16(double (&x)[4], const double* Dp, const double* ri, int q, F&& per_pivot) {
    double dk[16][4];
    static_for<16>([&](auto kc) {
        constexpr int k = decltype(kc)::value, ki = k >> 2;
#pragma unroll
        for (int i = ki; i < 4; ++i) dk[k][i] = Dp[(4 * i + q) * DPS + k];
    });
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] *= ri[4 * i + q];
    static_for<16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int kq = k & 3, ki = k >> 2;
        const double v = quad_bcast<kq>(x[ki]);
#pragma unroll
        for (int i = ki; i < 4; ++i) x[i] = fma(-v, dk[k][i], x[i]);
        per_pivot(kc);
    });
}

// Factor the 64 x 64 tile S (LDS, S[r][c], stride LT, lower part valid) in place: on exit S holds L (strict upper part
// zero), rinv[c] = 1 / L_cc and Dp (4 DPB doubles of LDS) the four diagonal 16 x 16 blocks in the form solve16 reads.
// Blocked by 16 columns, wave w owns rows 16 w .. 16 w + 15:
//   (1) every wave w > cb subtracts the contribution of the block columns to the left from its 16 x 16 block (w, cb) (one
//       MFMA product, K = 16 cb, wave-local);
//   (2) wave cb factors its diagonal 16 x 16 block: entries in registers, the pivot row / column exchanged by DPP row
//       broadcast and ds_bpermute -- all inside ONE wave, so the 16 pivots need no workgroup barrier;
//   (3) after a barrier the waves below solve their 16 x 16 block against it (rows independent, in registers) and at once
//       subtract its square from their own diagonal block (w, w).
// Two workgroup barriers per 16 pivots instead of one per pivot.
// The four column blocks are a RUNTIME loop (one copy of the 16 unrolled pivots, not four): the step kernel's code was 210 KB
// with everything unrolled, several times the instruction cache, and the pivot chain -- one instruction every few cycles, no
// reuse -- then runs at the rate instructions arrive from the L2, which a streaming SYRK on the other CUs keeps busy.
#ifdef SGP_POTF2_LEFT_LOOKING          // A/B switch: the diagonal blocks updated left-looking, in front of their pivot runs
constexpr int POTF2_LEFT = 1;
#else
constexpr int POTF2_LEFT = 0;
#endif
// acc += A[16 x 16 K] B^T for K = 16 chunks: rows of A at ap, of B at bp (both LDS, [row][k], stride LT, already offset by the
// lane's row and k), operands of all chunks first, four independent accumulators
template <int CHUNKS>
__device__ __forceinline__ void mma_left(d4 (&acc)[4], const double* ap, const double* bp) {
    double av[4 * CHUNKS], bv[4 * CHUNKS];
#pragma unroll
    for (int s4 = 0; s4 < 4 * CHUNKS; ++s4) { av[s4] = ap[4 * s4]; bv[s4] = bp[4 * s4]; }
#pragma unroll
    for (int s4 = 0; s4 < 4 * CHUNKS; ++s4) acc[s4 & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc[s4 & 3], 0, 0, 0);
}
__device__ __forceinline__ void mma_left_n(d4 (&acc)[4], const double* ap, const double* bp, int chunks) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = (d4){0.0, 0.0, 0.0, 0.0};
    if (chunks == 1) mma_left<1>(acc, ap, bp);
    else if (chunks == 2) mma_left<2>(acc, ap, bp);
    else if (chunks == 3) mma_left<3>(acc, ap, bp);
}
// `idle(iv)`: what a wave does that has nothing to do in barrier interval iv (2 cb: the pivot run of block cb -- every wave but
// cb; 2 cb + 1: the solve phase behind it -- the waves <= cb).  k_potrf_step gives such waves blocks of the panel solve.
template <class Idle>
__device__ __forceinline__ void potf2_tile(double* S, double* Dp, double* rinv, int* info, int col_base, int n_valid, Idle&& idle) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = 16 * wave;
    const int li = lane & 15, lk = lane >> 4;
    const int rr = lane >> 2, q = lane & 3;
#pragma unroll 1
    for (int cb = 0; cb < 4; ++cb) {
        if (cb > 0 && wave > cb - POTF2_LEFT) {                          // (the pivot wave's own block is already up to date, see (3))
            d4 acc[4];
            mma_left_n(acc, S + (r0 + li) * LT + lk /* A[i][k] = L[r0 + i][k] */, S + (16 * cb + li) * LT + lk /* B[k][j] = L[16 cb + j][k] */, cb);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                S[(r0 + lk + 4 * r) * LT + 16 * cb + li] -= (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
        }
        if (wave == cb) {
            // Pivot-phase coordinates: lane = 16 pq + pr holds row pr, columns 4 i + pq -- of the FULL symmetric block, so
            // that the pivot ROW (the same numbers as the pivot column) is where a DPP row broadcast can reach it.
            const int pr = lane & 15, pq = lane >> 4;
            double a[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * i + pq;
                a[i] = (c <= pr) ? S[(r0 + pr) * LT + 16 * cb + c] : S[(r0 + c) * LT + 16 * cb + pr];
                lo[i] = 0.0;
            }
            // The loop is bound by what this one wave can issue (~7 cycles per instruction) and by the LDS crossbar
            // (ds_bpermute: ~75 cycles, DPP: ~14), so everything that can wait does: the column is kept unnormalised (lo),
            // lane k keeps pivot k, and 1 / sqrt(d), the scaling of the columns and the failure test happen once after the
            // 16 pivots, for all of them in parallel.  On the dependent chain: pivot -> v_rcp_f64 -> e = 1 - d r ->
            // w = e + e^2 -> 1 / d = r (1 + w) (to rounding) -> rank-1 update -> next pivot.
            // A non-positive pivot is reported and NOT patched: what follows it in the factor is NaN / garbage (LAPACK
            // leaves it undefined).  Finished columns and the rows above the pivot receive garbage updates; nothing reads
            // them again.
            double dsave = 1.0;
            static_for<16>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int kq = k & 3, ki = k >> 2;
                // A[k][4 i + pq]: register a[i] of lane (pq, k) -- lane k of this lane's own DPP row
                double y[4];
#pragma unroll
                for (int i = ki; i < 4; ++i) y[i] = row_bcast<k>(a[i]);
                // A[pr][k]: register a[ki] of lane (kq, pr) -- another DPP row, through the LDS crossbar
                // (v_permlane32_swap + v_permlane16_swap can do this without LDS, but measured slower: 9.0 vs 7.6 us per tile)
                const int src = 4 * (16 * kq + pr);
                const double x = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(a[ki])),
                                                  __builtin_amdgcn_ds_bpermute(src, __double2loint(a[ki])));
                const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[ki]), 16 * kq + k),
                                                  __builtin_amdgcn_readlane(__double2loint(a[ki]), 16 * kq + k));
                dsave = (lane == k) ? d : dsave;
                // 1 / d = r / (1 - e) = r (1 + e + e^2 + ...), e = 1 - d r ~ 2^-25: two terms past r are exact to rounding
                const double r = __builtin_amdgcn_rcp(d);
                const double e = fma(-d, r, 1.0);
                const double w = fma(e, e, e);
                const double dinv = fma(r, w, r);
                lo[ki] = (pq == kq && pr >= k) ? x : lo[ki];  // L[pr][k] sqrt(d_k); rows above the pivot stay zero
                // The update is formed as (x y) / d, not (x / d) y: the product is the same number in entry (r, c) and in its
                // mirror image (c, r), so the block stays BITWISE symmetric and this elimination is the Cholesky recurrence.
                // With (x / d) y the two triangles drift apart by rounding, the factor is then that of an LU (column part
                // kept, row part discarded), L L^T = A holds only to cond * eps, and the Schur complements of an
                // ill-conditioned K_uu (cond 1e9) came out 1e5 times less accurate (measured: 1e-7 instead of 1e-12).
#pragma unroll
                for (int i = ki; i < 4; ++i) a[i] = fma(-dinv, x * y[i], a[i]);
            });
            const unsigned long long failed = __ballot(lane < 16 && !(dsave > 0.0));
            if (failed != 0ull && lane == 0) {
                const int bad = __builtin_ctzll(failed);
                if (col_base + 16 * cb + bad < n_valid) atomicCAS(info, 0, col_base + 16 * cb + bad + 1);
            }
            const double ri = rsqrt_nr(dsave);
            if (lane < 16) rinv[16 * cb + lane] = ri;
            __builtin_amdgcn_wave_barrier();                  // (same wave: the LDS queue keeps the order; this keeps the compiler's)
            const double rrow = rinv[16 * cb + pr];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * i + pq;
                const double l = lo[i] * rinv[16 * cb + c];
                S[(r0 + pr) * LT + 16 * cb + c] = l;
                Dp[cb * DPB + pr * DPS + c] = (c < pr) ? l * rrow : 0.0;
            }
        } else {
            idle(2 * cb);
        }
        __syncthreads();
        if (wave > cb) {
            double x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = S[(r0 + rr) * LT + 16 * cb + 4 * i + q];
            solve16(x, Dp + cb * DPB, rinv + 16 * cb, q, [](auto) {});
#pragma unroll
            for (int i = 0; i < 4; ++i) S[(r0 + rr) * LT + 16 * cb + 4 * i + q] = x[i];
            // The wave's OWN diagonal block loses L(w, cb) L(w, cb)^T right away -- its rows only, no other wave involved --
            // so that a wave that becomes the pivot wave starts its 16 pivots with nothing left to subtract: the
            // left-looking form put 4 cb dependent MFMAs (~200 cycles each) in front of every pivot run.
            __builtin_amdgcn_wave_barrier();
            if constexpr (!POTF2_LEFT) {
                const double* dp = S + (r0 + li) * LT + 16 * cb + lk;    // A[i][k] = L[r0 + i][16 cb + k] = B[k][i]
                double dv[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) dv[s4] = dp[4 * s4];
                d4 g[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    g[s4] = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[s4], dv[s4], (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    S[(r0 + lk + 4 * r) * LT + r0 + li] -= (g[0][r] + g[1][r]) + (g[2][r] + g[3][r]);
            }
        } else {
            if (wave < cb) {
                // rows of finished waves: the strict upper part of this block column is zero
#pragma unroll
                for (int i = 0; i < 4; ++i) S[(r0 + rr) * LT + 16 * cb + 4 * i + q] = 0.0;
            }
            idle(2 * cb + 1);
        }
        __syncthreads();
    }
}

// One 16 x 16 block of the solve X L^T = B, in place (X: LDS tile, stride LT; S holds L as potf2_tile left it, with Dp and
// rinv = 1 / diag(L)): column block cb of the 16 rows of row block rb -- first the contribution of the column blocks to its
// left (one MFMA product, K = 16 cb, four independent accumulators: a dependent v_mfma_f64 chain costs ~200 cycles per link),
// then the 16 x 16 triangle with 4 lanes per row in registers (solve16).  Wave-local; any wave may take any row block.  The
// unit of the scheduled solve group (k_potrf_step).
// The two halves on their own: the update needs the column blocks to the left of cb solved (and L's block row cb, final one
// phase before D_cb is), the substitution needs D_cb -- so a block's update can run an interval ahead of its substitution.
__device__ __forceinline__ void trsm_update(double* X, const double* S, int rb, int cb) {       // cb >= 1
    const int lane = threadIdx.x & 63;
    const int r0 = 16 * rb;
    const int li = lane & 15, lk = lane >> 4;
    d4 acc[4];
    mma_left_n(acc, X + (r0 + li) * LT + lk, S + (16 * cb + li) * LT + lk, cb);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        X[(r0 + lk + 4 * r) * LT + 16 * cb + li] -= (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
}
__device__ __forceinline__ void trsm_solve(double* X, const double* Dp, const double* rinv, int rb, int cb) {
    const int lane = threadIdx.x & 63;
    const int r0 = 16 * rb;
    const int rr = lane >> 2, q = lane & 3;
    double x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = X[(r0 + rr) * LT + 16 * cb + 4 * i + q];
    solve16(x, Dp + cb * DPB, rinv + 16 * cb, q, [](auto) {});
#pragma unroll
    for (int i = 0; i < 4; ++i) X[(r0 + rr) * LT + 16 * cb + 4 * i + q] = x[i];
}
// The solve work of one row block in solve interval it (0 .. 3 = I4 .. I7): the substitution of column block `it` behind its
// update -- except that block 1's update runs ahead, behind block 0's substitution in I4: the interval of a pivot run has
// room, the phase interval I5 (0.85 us) has none for more than a substitution.  (Block 3's update ahead in I6 as well:
// I6 grows by more than I7 shrinks.)
__device__ __forceinline__ void trsm_interval(double* X, const double* S, const double* Dp, const double* rinv, int rb, int it) {
    if (it >= 2) trsm_update(X, S, rb, it);
    trsm_solve(X, Dp, rinv, rb, it);
    if (it == 0) trsm_update(X, S, rb, 1);
}
// K-slice c (columns 16 c .. 16 c + 15 of the solved tile X) of the lower 16 x 16 tile (R, C) of X X^T, formed transposed
// (A operand = the column tile) so that the stores run along Dn's columns (Dn: 64 x 64, column-major, ld 64; the next step
// subtracts it from its diagonal tile instead of recomputing the product in front of its factorisation)
__device__ __forceinline__ void syrk_slice(d4& g, const double* X, int R, int C, int c) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double* pa = X + (16 * C + li) * LT + 16 * c + lk;
    const double* pb = X + (16 * R + li) * LT + 16 * c + lk;
    double av[4], bv[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) { av[k4] = pa[4 * k4]; bv[k4] = pb[4 * k4]; }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) g = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k4], bv[k4], g, 0, 0, 0);
}
__device__ __forceinline__ void syrk_store(const d4& g, double* __restrict__ Dn, int R, int C) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) Dn[(16 * C + lk + 4 * r) * TB + 16 * R + li] = g[r];
}

// Invert the 64 x 64 lower-triangular tile S (LDS, S[r][c], stride LT; rinv[c] = 1 / L_cc) into Wt (LDS, same layout,
// strict upper part zero).  Recursive doubling at 16-column granularity, like the matrix-level launch_trtri:
//   (1) wave w inverts its own diagonal 16 x 16 block (4 lanes per row, right-to-left pivots, in registers, wave-local);
//   (2) 16-blocks:  W21 = -W22 (L21 W11) for the pairs (0,1) and (2,3)      -- waves 1 and 3, two MFMA products each;
//   (3) 32-blocks:  W21 = -W22 (L21 W11), a 2 x 2 grid of 16 x 16 MFMA tiles -- one per wave, K = 32.
// T (>= 32 * 33 doubles of LDS) holds the intermediate L21 W11.
__device__ __forceinline__ void trtri_tile(const double* S, const double* rinv, double* Wt, double* T) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int rr = lane >> 2, q = lane & 3;
    for (int e = tid; e < TB * LT; e += 256) Wt[e] = 0.0;
    __syncthreads();
    {   // (1) diagonal 16 x 16 blocks
        const int b0 = 16 * wave;
        double w[4], wo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { w[i] = (4 * i + q == rr) ? 1.0 : 0.0; wo[i] = 0.0; }
        // operands into registers first, unconditionally (see solve16): L_kc for c = 4 i + q, i <= k / 4, and 1 / L_kk
        double lk_[16][4], rv[16];
        static_for<16>([&](auto kc) {
            constexpr int k = decltype(kc)::value, ki = k >> 2;
            rv[k] = rinv[b0 + k];
#pragma unroll
            for (int i = 0; i <= ki; ++i) lk_[k][i] = S[(b0 + k) * LT + b0 + 4 * i + q];
        });
        static_for<16>([&](auto kc) {
            constexpr int k = 15 - decltype(kc)::value;
            constexpr int kq = k & 3, ki = k >> 2;
            const double v = quad_bcast<kq>(w[ki]) * rv[k];                      // W_rk (zero for k > r)
            if (q == kq) wo[ki] = v;
#pragma unroll
            for (int i = 0; i <= ki; ++i) w[i] = fma(-v, lk_[k][i], w[i]);       // (c >= k: column k is finished, zeros above)
        });
#pragma unroll
        for (int i = 0; i < 4; ++i) Wt[(b0 + rr) * LT + b0 + 4 * i + q] = wo[i];
    }
    __syncthreads();
    if (wave & 1) {   // (2) pairs of 16-blocks: rows of block `wave`, columns of block `wave - 1`
        const int rb = 16 * wave, cb = 16 * (wave - 1);
        double* Tw = T + (wave >> 1) * (16 * 17);
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)                                           // T = L21 W11
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[(rb + li) * LT + cb + 4 * s4 + lk],
                                                       Wt[(cb + 4 * s4 + lk) * LT + cb + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Tw[(lk + 4 * r) * 17 + li] = acc[r];
        __builtin_amdgcn_wave_barrier();
        acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)                                           // W21 = -W22 T
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[(rb + li) * LT + rb + 4 * s4 + lk], Tw[(4 * s4 + lk) * 17 + li],
                                                       acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Wt[(rb + lk + 4 * r) * LT + cb + li] = -acc[r];
    }
    __syncthreads();
    {   // (3) the 32-blocks: tile (ti, tj) of W21, rows 32 + 16 ti, columns 16 tj
        const int ti = wave >> 1, tj = wave & 1;
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4)                                           // T = L21 W11, K = 32
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[(32 + 16 * ti + li) * LT + 4 * s4 + lk],
                                                       Wt[(4 * s4 + lk) * LT + 16 * tj + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * ti + lk + 4 * r) * 33 + 16 * tj + li] = acc[r];
        __syncthreads();
        acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4)                                           // W21 = -W22 T
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[(32 + 16 * ti + li) * LT + 32 + 4 * s4 + lk],
                                                       T[(4 * s4 + lk) * 33 + 16 * tj + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Wt[(32 + 16 * ti + lk + 4 * r) * LT + 16 * tj + li] = -acc[r];
    }
    __syncthreads();
}

// coalesced copies between a column-major global tile and an LDS tile S[r][c]
struct TileRegs { double v[16]; };
// register u of thread tid holds element (r, c) = (2 (e & 31) + (u & 1), e >> 5), e = tid + 256 (u >> 1): two consecutive
// rows per thread, so that a tile is 8 sixteen-byte loads per thread instead of 16 eight-byte ones
__device__ __forceinline__ void tile_g2r(TileRegs& t, const double* __restrict__ A, size_t ld, int row0, int col0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (threadIdx.x & 255) + 256 * u, c = e >> 5, r = (e & 31) * 2;
        const double2 w = *reinterpret_cast<const double2*>(A + (size_t)(col0 + c) * ld + row0 + r);
        t.v[2 * u] = w.x;
        t.v[2 * u + 1] = w.y;
    }
}
// Lambda = Lambda0 + W (x) Psi2 in index-reversed order, evaluated on the fly: step 0 of the Lambda factorisation reads its
// tiles through this instead of from memory, so that no separate k_form_lambda launch (and no round trip of Lambda through
// HBM) sits in front of the chain.  stats == nullptr: the matrix is already in A.
// Overlapped sweep (sgp_api.hip, sweep_overlapped): the statistics arrive in GROUPS of tile columns of P Lambda P (= tile rows of
// Psi2, last rows first) while the factorisation is already running.  Tile column c is formed -- added into the matrix -- by the
// workgroups of step form_step[c] <= c, which first wait (bounded) until col_words[col_group[c]] >= col_need (the sweep's number,
// written by a k_join_set behind the group's k_assemble);
// until then the tile in A holds only the (negative) rank-64 updates that steps 1 .. collected for it.  col_words == nullptr:
// the statistics are complete before step 0, which forms every tile (form_step all zero).
constexpr int LAM_MAX_GROUPS = 8;
constexpr int LAM_MAX_COLS = 64;            // CU_MAXQ / TB
struct LamForm {
    const double* stats;
    const double* Lambda0;
    const double* xi0;
    double* xi;
    const Params* P;
    int M, Mp, d_out, Q, prior_form;
    int64_t* stamps;
    const long long* col_words;
    long long col_need;
    unsigned char form_step[LAM_MAX_COLS], col_group[LAM_MAX_COLS];      // col_group 0xff: in stream order, nothing to wait for
    int spin_limit;
    int* sync_status;
    int trace_chain;            // diagnostics (SGP_SWEEP_TRACE): 1 = this launch belongs to the Lambda chain (its slots), 0 = K_uu chain
};
// Straight-line on purpose, with the two uniform choices (dense prior? several outputs?) made OUTSIDE the 16-entry loop:
// any branch inside it splits the loop body into basic blocks, the compiler then waits for each load before the next is
// issued, and step 0 was measured 13-15 us longer than a step that just loads its tiles.
// BYPASS: the workgroup had to wait for its statistics (they were written while this kernel was already running): read them
// past this XCD's L2, which was invalidated at kernel start, i.e. possibly before the producer's write-back
__device__ __forceinline__ double load_agent(const double* p) {
    return __hip_atomic_load((const __attribute__((address_space(1))) double*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool DENSE, bool MULTI, bool BYPASS = false>
__device__ __forceinline__ double lambda_entry(const LamForm& f, int gi, int gj, int Qp, double prior_iso, double w00) {
    const bool inside = gi < f.Q && gj < f.Q;
    const int si = inside ? gi : 0, sj = inside ? gj : 0;              // safe indices for the pad entries
    int a = 0, i = si, b = 0, j = sj;
    if constexpr (MULTI) { a = si / f.M; i = si % f.M; b = sj / f.M; j = sj % f.M; }
    double psi;
    if constexpr (BYPASS) psi = load_agent(f.stats + (size_t)j * f.Mp + i); else psi = f.stats[(size_t)j * f.Mp + i];
    const double diag = (gi == gj) ? 1.0 : 0.0;
    double prior, w;
    if constexpr (DENSE) prior = f.Lambda0[(size_t)sj * Qp + si]; else prior = diag * prior_iso;
    if constexpr (MULTI) w = f.P->W[a + b * f.d_out]; else w = w00;
    // arithmetic mask instead of a select on the loaded values: with `inside ? fma(w, psi, prior) : diag` the compiler sinks
    // the loads into a per-entry branch (s_cbranch_execz + load + s_waitcnt vmcnt(0), sixteen times over) -- seen in the ISA
    const double m = inside ? 1.0 : 0.0;
    return fma(m * w, psi, fma(m, prior, (1.0 - m) * diag));
}
// tile (row0.., col0..) of P Lambda P: entry (r, c) is Lambda[Qp-1-r][Qp-1-c]
template <bool DENSE, bool MULTI, bool BYPASS>
__device__ __forceinline__ void tile_form_r_impl(TileRegs& t, const LamForm& f, int Qp, int row0, int col0) {
    const double prior_iso = BYPASS ? load_agent(&f.P->prior_iso) : f.P->prior_iso, w00 = BYPASS ? load_agent(&f.P->W[0]) : f.P->W[0];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = (threadIdx.x & 255) + 256 * (u >> 1), c = e >> 5, r = (e & 31) * 2 + (u & 1);      // the layout of tile_g2r
        t.v[u] = lambda_entry<DENSE, MULTI, BYPASS>(f, Qp - 1 - (row0 + r), Qp - 1 - (col0 + c), Qp, prior_iso, w00);
    }
}
__device__ __forceinline__ void tile_form_r(TileRegs& t, const LamForm& f, int Qp, int row0, int col0, bool bypass = false) {
    const bool dense = f.prior_form == 1, multi = f.d_out > 1;
    if (bypass && !multi) {                 // (the overlapped sweep is UniSGP only)
        if (dense) tile_form_r_impl<true, false, true>(t, f, Qp, row0, col0); else tile_form_r_impl<false, false, true>(t, f, Qp, row0, col0);
        return;
    }
    if (dense) { if (multi) tile_form_r_impl<true, true, false>(t, f, Qp, row0, col0); else tile_form_r_impl<true, false, false>(t, f, Qp, row0, col0); }
    else       { if (multi) tile_form_r_impl<false, true, false>(t, f, Qp, row0, col0); else tile_form_r_impl<false, false, false>(t, f, Qp, row0, col0); }
}
__device__ __forceinline__ void tile_r2s(double* S, const TileRegs& t) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = (threadIdx.x & 255) + 256 * (u >> 1), c = e >> 5, r = (e & 31) * 2 + (u & 1);
        S[r * LT + c] = t.v[u];
    }
}
__device__ __forceinline__ void tile_g2s(double* S, const double* __restrict__ A, size_t ld, int row0, int col0) {
    TileRegs t;
    tile_g2r(t, A, ld, row0, col0);
    tile_r2s(S, t);
}
// The tile a Cholesky step starts from: what A holds for it (load_old) plus, in the step that forms it, Lambda's tile
// (do_form) -- see LamForm.  Neither: the tile has not received anything yet (zeros).
__device__ __forceinline__ void tile_fetch(TileRegs& t, bool do_form, bool load_old, const double* __restrict__ A, const LamForm& f,
                                           int ld, int row0, int col0, bool bypass) {
    if (do_form) {
        tile_form_r(t, f, ld, row0, col0, bypass);
        if (load_old) {
            TileRegs o;
            tile_g2r(o, A, ld, row0, col0);
#pragma unroll
            for (int u = 0; u < 16; ++u) t.v[u] += o.v[u];
        }
    } else if (load_old) {
        tile_g2r(t, A, ld, row0, col0);
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) t.v[u] = 0.0;
    }
}
// (all LDS reads first, then eight 16-byte stores per thread: this is the last thing a panel block of a Cholesky step does)
__device__ __forceinline__ void tile_s2g(const double* S, double* __restrict__ A, size_t ld, int row0, int col0) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (threadIdx.x & 255) + 256 * u, c = e >> 5, r = (e & 31) * 2;
        v[u] = make_double2(S[r * LT + c], S[(r + 1) * LT + c]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (threadIdx.x & 255) + 256 * u, c = e >> 5, r = (e & 31) * 2;
        *reinterpret_cast<double2*>(A + (size_t)(col0 + c) * ld + row0 + r) = v[u];
    }
}
// the same for the transposed tile: A(r, c) = S[c][r]
__device__ __forceinline__ void tile_s2g_t(const double* S, double* __restrict__ A, size_t ld, int row0, int col0) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (threadIdx.x & 255) + 256 * u, c = e >> 5, r = (e & 31) * 2;
        v[u] = make_double2(S[c * LT + r], S[c * LT + r + 1]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (threadIdx.x & 255) + 256 * u, c = e >> 5, r = (e & 31) * 2;
        *reinterpret_cast<double2*>(A + (size_t)(col0 + c) * ld + row0 + r) = v[u];
    }
}
__device__ __forceinline__ void tile_sub_acc(double* S, const Acc4& acc, int lane, int wr, int wc) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                S[acc_row(lane, wr, ti, r) * LT + acc_col(lane, wc, tj)] -= acc.t[ti][tj][r];
}

// Hazard handled here: every block of the panel column reads the UNFACTORED diagonal tile A_jj from global memory, so
// the diagonal block must not overwrite it in place during the same launch.  It parks L_jj in `scratch` (one tile) and
// the next step's block (0, 0) moves it into place (kernel boundary = all readers done).  The last step has no readers.
// While the panel blocks solve their tiles, the otherwise idle diagonal block also inverts L_jj (Winv != nullptr): the
// diagonal tiles of L^-1 that the recursive triangular inverse starts from, without a launch of their own.
constexpr int PS32 = 48;        // LDS panel row stride for 32-wide panels: 2 * PS32 dwords = 32 (mod 64)

// panel[k][i] = G[(row0 + i) + (col0 + k) * ld], i < 32, k < 64
__device__ __forceinline__ void load_panel32_n(double* panel, const double* __restrict__ G, size_t ld, int row0, int col0, int tid) {
    double2 v0[2], v1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int t = tid + 256 * u, k = t >> 3, rq = t & 7;
        const double* src = G + (size_t)(col0 + k) * ld + row0 + rq * 4;
        v0[u] = *reinterpret_cast<const double2*>(src);
        v1[u] = *reinterpret_cast<const double2*>(src + 2);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int t = tid + 256 * u, k = t >> 3, rq = t & 7;
        double* dst = panel + k * PS32 + rq * 4;
        dst[0] = v0[u].x; dst[1] = v0[u].y; dst[2] = v1[u].x; dst[3] = v1[u].y;
    }
}
// panel[k][i] = G[(row0 + k) + (col0 + i) * ld], i < 32, k < 64  (transposing load: lanes walk the 32 columns so that
// the four LDS rows written by one instruction land on different banks)
__device__ __forceinline__ void load_panel32_t(double* panel, const double* __restrict__ G, size_t ld, int row0, int col0, int tid) {
    double2 v0[2], v1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int t = tid + 256 * u, i = t & 31, g = t >> 5;
        const double* src = G + (size_t)(col0 + i) * ld + row0 + g * 4;
        v0[u] = *reinterpret_cast<const double2*>(src);
        v1[u] = *reinterpret_cast<const double2*>(src + 2);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int t = tid + 256 * u, i = t & 31, g = t >> 5;
        panel[(g * 4 + 0) * PS32 + i] = v0[u].x;
        panel[(g * 4 + 1) * PS32 + i] = v0[u].y;
        panel[(g * 4 + 2) * PS32 + i] = v1[u].x;
        panel[(g * 4 + 3) * PS32 + i] = v1[u].y;
    }
}


// ------------------------------------------------------------------------------------------------
// Inverse factor W = L^-1 row by row, riding along the Cholesky steps: once step i has finished, block row i of W is
//   W_ic = - W_ii * sum_{k=c}^{i-1} L_ik W_kc ,   c < i      (W_ii: the diagonal tile's inverse, trtri_tile)
// and needs only finished things -- columns < i of L, rows < i of W.  Its tiles are computed by EXTRA workgroups of the
// next step's launch (k_potrf_step, blockIdx >= the step's own tile count): the steps are latency-bound on a handful of
// CUs, so the inverse costs one short extra launch (for the last row) instead of a phase of its own.
// One workgroup = the 64 x 32 half h of tile (i, c): per k one 64 x 64 x 32 product on the matrix cores (wave w: rows
// 16 w .., two 16 x 16 tiles, K split over two accumulator sets), then the product with W_ii, then a coalesced store.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void winv_half_mma(d4& c0a, d4& c1a, d4& c0b, d4& c1b, const double* As, const double* Bs, int lane,
                                              int wave) {
    const int li = lane & 15, lk = lane >> 4;
    const double* ap = As + lk * PS + wave * 16 + li;
    const double* bp = Bs + lk * PS32 + li;
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 2) {
        const double a0 = ap[(4 * k4) * PS], b00 = bp[(4 * k4) * PS32], b01 = bp[(4 * k4) * PS32 + 16];
        const double a1 = ap[(4 * k4 + 4) * PS], b10 = bp[(4 * k4 + 4) * PS32], b11 = bp[(4 * k4 + 4) * PS32 + 16];
        c0a = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b00, c0a, 0, 0, 0);
        c1a = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b01, c1a, 0, 0, 0);
        c0b = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b10, c0b, 0, 0, 0);
        c1b = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b11, c1b, 0, 0, 0);
    }
}

// mode 0: the whole tile half at once.  The two-launch pipeline used by the Cholesky steps splits it:
// mode 1 (one launch early): T' = sum_{k=c}^{i-2} L_ik W_kc, parked in W's own (i, c) tile;
// mode 2 (finish):           W_ic = - W_ii (T' + L_{i,i-1} W_{i-1,c})   -- two products instead of up to i - c + 1.
__device__ __forceinline__ void winv_row_tile(const double* __restrict__ L, double* __restrict__ W, int ld, int i, int c, int h,
                                              double* lds, int mode, double* lds2 = nullptr) {
    double* As = lds;                                     // As[kk][r], 64 x 64, stride PS
    double* Bs = lds + TB * PS;                           // Bs[kk][jj], 64 x 32, stride PS32
    double* Os = Bs;                                      // Os[jj][r], stride LT: staging for coalesced tile-half I/O
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int col0 = c * TB + 32 * h;
    const d4 Z = (d4){0.0, 0.0, 0.0, 0.0};
    d4 t0a = Z, t1a = Z, t0b = Z, t1b = Z;
    if (mode == 2 && lds2) {
        // The finishing step with EVERY operand requested before the first wait (round 4): the parked partial sum straight into
        // the accumulator layout, the two panels of the one remaining product L_{i,i-1} W_{i-1,c}, and W_ii into a second panel
        // buffer -- one memory round trip where the general path below has three (partial -> panels -> W_ii).  This is the launch
        // behind the last Cholesky step: nothing hides it, the sweep waits for it (7.0 - 7.4 us per workgroup before).
        double* As2 = lds2;                               // As2[kk][r] = W_ii[r][kk], stride PS
        const int k = i - 1;
        double2 a0[4], a1[4], w0[4], w1[4], b0[2], b1[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, kk = t >> 4, rq = t & 15;
            const double* sa = L + (size_t)(k * TB + kk) * ld + i * TB + rq * 4;
            const double* sw = W + (size_t)(i * TB + kk) * ld + i * TB + rq * 4;
            a0[u] = *reinterpret_cast<const double2*>(sa); a1[u] = *reinterpret_cast<const double2*>(sa + 2);
            w0[u] = *reinterpret_cast<const double2*>(sw); w1[u] = *reinterpret_cast<const double2*>(sw + 2);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + 256 * u, ii = t & 31, g = t >> 5;
            const double* sb = W + (size_t)(col0 + ii) * ld + k * TB + g * 4;
            b0[u] = *reinterpret_cast<const double2*>(sb); b1[u] = *reinterpret_cast<const double2*>(sb + 2);
        }
        if (c <= i - 2) {                                 // (the parked partial: T' of mode 1, in W's own (i, c) tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * TB + wave * 16 + lk + 4 * r;
                t0a[r] = W[(size_t)(col0 + li) * ld + row];
                t1a[r] = W[(size_t)(col0 + 16 + li) * ld + row];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, kk = t >> 4, rq = t & 15;
            double* da = As + kk * PS + rq * 4;
            double* dw = As2 + kk * PS + rq * 4;
            da[0] = a0[u].x; da[1] = a0[u].y; da[2] = a1[u].x; da[3] = a1[u].y;
            dw[0] = w0[u].x; dw[1] = w0[u].y; dw[2] = w1[u].x; dw[3] = w1[u].y;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + 256 * u, ii = t & 31, g = t >> 5;
            Bs[(g * 4 + 0) * PS32 + ii] = b0[u].x;
            Bs[(g * 4 + 1) * PS32 + ii] = b0[u].y;
            Bs[(g * 4 + 2) * PS32 + ii] = b1[u].x;
            Bs[(g * 4 + 3) * PS32 + ii] = b1[u].y;
        }
        __syncthreads();
        winv_half_mma(t0a, t1a, t0b, t1b, As, Bs, lane, wave);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + lk + 4 * r;
            Bs[row * PS32 + li] = t0a[r] + t0b[r];
            Bs[row * PS32 + 16 + li] = t1a[r] + t1b[r];
        }
        __syncthreads();
        d4 o0a = Z, o1a = Z, o0b = Z, o1b = Z;
        winv_half_mma(o0a, o1a, o0b, o1b, As2, Bs, lane, wave);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + lk + 4 * r;
            Os[li * LT + row] = -(o0a[r] + o0b[r]);
            Os[(16 + li) * LT + row] = -(o1a[r] + o1b[r]);
        }
        __syncthreads();
        for (int e = tid; e < 32 * TB; e += 256) {
            const int jj = e >> 6, r = e & 63;
            W[(size_t)(col0 + jj) * ld + i * TB + r] = Os[jj * LT + r];
        }
        return;
    }
    const int kbeg = (mode == 2) ? i - 1 : c, kend = (mode == 1) ? i - 1 : i;      // [kbeg, kend)
    if (mode == 2 && c <= i - 2) {                        // resume from the parked partial sum
        for (int e = tid; e < 32 * TB; e += 256) {
            const int jj = e >> 6, r = e & 63;
            Os[jj * LT + r] = W[(size_t)(col0 + jj) * ld + i * TB + r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + lk + 4 * r;
            t0a[r] = Os[li * LT + row];
            t1a[r] = Os[(16 + li) * LT + row];
        }
    }
    for (int k = kbeg; k < kend; ++k) {
        __syncthreads();
        load_panel_n(As, L, ld, i * TB, k * TB, TB, tid);             // As[kk][r]  = L[i*64 + r, k*64 + kk]
        load_panel32_t(Bs, W, ld, k * TB, col0, tid);                 // Bs[kk][jj] = W[k*64 + kk, col0 + jj]
        __syncthreads();
        winv_half_mma(t0a, t1a, t0b, t1b, As, Bs, lane, wave);
    }
    __syncthreads();
    d4 o0 = Z, o1 = Z;
    if (mode == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] = t0a[r] + t0b[r]; o1[r] = t1a[r] + t1b[r]; }
    } else {
        load_panel_n(As, W, ld, i * TB, i * TB, TB, tid);             // As[kk][r]  = W_ii[r][kk]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + lk + 4 * r;
            Bs[row * PS32 + li] = t0a[r] + t0b[r];
            Bs[row * PS32 + 16 + li] = t1a[r] + t1b[r];
        }
        __syncthreads();
        d4 o0a = Z, o1a = Z, o0b = Z, o1b = Z;
        winv_half_mma(o0a, o1a, o0b, o1b, As, Bs, lane, wave);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] = -(o0a[r] + o0b[r]); o1[r] = -(o1a[r] + o1b[r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + lk + 4 * r;
        Os[li * LT + row] = o0[r];
        Os[(16 + li) * LT + row] = o1[r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * TB; e += 256) {
        const int jj = e >> 6, r = e & 63;
        W[(size_t)(col0 + jj) * ld + i * TB + r] = Os[jj * LT + r];
    }
}

// ------------------------------------------------------------------------------------------------
// Sigma = W^T W rides along as well: Sigma(I,J) = sum_{k >= I} W(k,I)^T W(k,J), and block row k of W is final one launch
// after step k.  Extra workgroups of launch j add row j - 2's contribution W(j-2,I)^T W(j-2,J) to every lower tile (I, J),
// I <= j - 2, of the accumulator Sacc (first contribution of a tile: plain store), one 64 x 64 x 64 product each.  The
// product launch after the factorisation (k_gemm32 mode 0) then only adds the last TWO block rows (the launch behind the last step
// carries no Sigma workgroups, see k_potrf_step) and runs its epilogue.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sigma_row_tile(const double* __restrict__ W, double* __restrict__ Sacc, int ld, int i, int I,
                                               int J, double* panels, double* tile) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    double* As = panels;
    double* Bs = panels + TB * PS;
    load_panel_t(As, W, ld, i * TB, I * TB, TB, tid);              // As[kk][r] = W[i*64 + kk, I*64 + r]
    if (I != J) load_panel_t(Bs, W, ld, i * TB, J * TB, TB, tid);  // Bs[kk][c] = W[i*64 + kk, J*64 + c]
    if (I == i) {
        for (int e = tid; e < TB * LT; e += 256) tile[e] = 0.0;    // first contribution to this tile
    } else {
        tile_g2s(tile, Sacc, ld, I * TB, J * TB);
    }
    __syncthreads();
    Acc4 acc;
    acc_zero(acc);
    tile_mma(acc, As, (I != J) ? Bs : As, TB, lane, wr, wc);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.t[ti][tj][r] = -acc.t[ti][tj][r];
    tile_sub_acc(tile, acc, lane, wr, wc);                        // tile += product
    __syncthreads();
    tile_s2g(tile, Sacc, ld, I * TB, J * TB);
}

// ------------------------------------------------------------------------------------------------
// The forward solve t = W (P xi) = L^-1 (P xi) rides along too (one workgroup per launch): block i of t is
//   t_i = W_ii (xi'_i - sum_{k<i} L_ik t_k),   xi'_m = xi[ld - 1 - m],
// computable one launch after step i (W_ii) -- so t is complete when the factorisation is, and mu = P W^T t (and p = P t for
// the closed-form Uv) follow with ONE mat-vec launch instead of two.
// ------------------------------------------------------------------------------------------------
// NP waves (4: a 256-thread launch; 8: all 512 threads of a k_potrf_step workgroup): wave `part` takes 64 / NP of the 64 columns of
// every block.  Round 4: eight waves, four block columns' loads in flight (the whole of row 7 at M = 512 in two rounds), W_ii requested
// with the first of them -- in the launch behind the last step this one workgroup was the last to leave (6.7 us against 5.2).
template <int NP>
__device__ __forceinline__ void tvec_role(const double* __restrict__ L, const double* __restrict__ W,
                                          const double* __restrict__ xi, double* __restrict__ t, int ld, int i, double* lds) {
    constexpr int CP = TB / NP;         // columns per wave
    double* red = lds;                  // [NP][64]
    double* rvec = lds + NP * TB;       // [64]
    const int tid = threadIdx.x, r = tid & 63, part = tid >> 6;
    const double* wb = W + (size_t)(i * TB + CP * part) * ld + i * TB + r;       // W_ii: lower triangular, zeros above
    double wv[CP];
#pragma unroll
    for (int u = 0; u < CP; ++u) wv[u] = wb[(size_t)u * ld];
    double acc = 0.0;
    for (int k = 0; k < i; k += 4) {            // four block columns' loads in flight; the FMAs keep the order k, u.  (A short last batch
        double v[4][CP], tv[4][CP];             // is padded with clamped loads that are multiplied away: no one-column rounds.)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kq = (k + q < i) ? k + q : i - 1;
            const double live = (k + q < i) ? 1.0 : 0.0;
            const double* base = L + (size_t)(kq * TB + CP * part) * ld + i * TB + r;
#pragma unroll
            for (int u = 0; u < CP; ++u) { v[q][u] = base[(size_t)u * ld]; tv[q][u] = t[kq * TB + CP * part + u] * live; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < CP; ++u) acc = fma(v[q][u], tv[q][u], acc);
    }
    auto tree = [&](int row) {          // fixed order over the waves
        double a[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) a[q] = red[q * TB + row];
#pragma unroll
        for (int w = 1; w < NP; w <<= 1)
#pragma unroll
            for (int q = 0; q + w < NP; q += 2 * w) a[q] += a[q + w];
        return a[0];
    };
    red[part * TB + r] = acc;
    __syncthreads();
    if (part == 0) rvec[r] = xi[ld - 1 - (i * TB + r)] - tree(r);
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < CP; ++u) s = fma(wv[u], rvec[CP * part + u], s);
    __syncthreads();
    red[part * TB + r] = s;
    __syncthreads();
    if (part == 0) t[i * TB + r] = tree(r);
}

// xi = xi0 + vec(B W) of the Lambda chain's step 0 (one workgroup, 256 threads)
__device__ __forceinline__ void form_xi(const LamForm& form, int ld, int tid, bool bypass = false) {
    const double* B = form.stats + (size_t)form.Mp * form.Mp;
    for (int gi = tid; gi < ld; gi += 256) {
        double v = 0.0;
        if (gi < form.Q) {
            const int aa = gi / form.M, i = gi % form.M;
            v = (form.prior_form == 1) ? form.xi0[gi] : 0.0;
            for (int e = 0; e < form.d_out; ++e) {
                const double* bp = B + (size_t)e * form.Mp + i;
                const double* wp = &form.P->W[e + aa * form.d_out];
                v = fma(bypass ? load_agent(bp) : *bp, bypass ? load_agent(wp) : *wp, v);
            }
        }
        form.xi[gi] = v;
    }
}
// wait (every wave's lane 0 polls; bounded) until tile-column group g of the statistics is complete; returns whether the
// wave had to wait -- its reads of the statistics then go past the L2 (see load_agent)
__device__ __forceinline__ bool wait_stat_group(const LamForm& f, int g) {
    int late = 0;
    if ((threadIdx.x & 63) == 0) {
        const long long* w = f.col_words + g;
        if (!join_ready(w, f.col_need)) {
            late = 1;
            spin_until(w, f.col_need, f.spin_limit, f.sync_status, SYNC_LATE_COLUMN);
        }
    }
    return __builtin_amdgcn_readfirstlane(late) != 0;
}

// Diagnostics (build with -DSGP_STEP_TRACE): 100 MHz stamps of the workgroup that owns tile (j + 1, j) of the Lambda chain,
// slot 64 j + 32 g + e for event e of group g (0 factoring, 1 solve) of step j < 8; read back with sgp_get_step_trace.
// SGP_STEP_TRACE_A: the panel block that is traced (1 = the one that also forms the next diagonal tile's update).
__device__ long long g_step_trace[8 * 64];
#ifndef SGP_STEP_TRACE_A
#define SGP_STEP_TRACE_A 1
#endif
#ifdef SGP_STEP_TRACE
#define STEP_TRACE(e) do { if (tv_t && a == SGP_STEP_TRACE_A && b == 0 && tw == 0 && lane == 0 && wave == 0 && j < 8) g_step_trace[j * 64 + (xgroup ? 32 : 0) + (e)] = realtime_ticks(); } while (0)
#else
#define STEP_TRACE(e) do { } while (0)
#endif

// Workgroups are launched with PSTEP_THREADS = 512 threads.  The blocks of the panel column use the second half (the "solve
// group", waves 4 .. 7): below the diagonal it carries the block's own tile -- its rank-64 update, then its triangular solve
// column block by column block, scheduled around the factoring group's eight barrier intervals (see the schedule in the kernel)
// -- while waves 0 .. 3 factor the diagonal tile; in the diagonal block (when the inverse factor is wanted) it runs the same
// solve on the identity, i.e. inverts L_jj.  Everywhere else waves 4 .. 7 leave at once (ended waves do not count at a barrier).
constexpr int PSTEP_THREADS = 512;
// TWINS: the block that owns tile (j + 1, j) also forms the next diagonal tile's update X X^T from its solved tile (160 MFMAs
// that can only start once column blocks of X are final, i.e. in the last three barrier intervals) and was the last to leave
// in every step (in-kernel exit stamps: +1.6 us after the other panel blocks, and the next step waits for the launch).
// POTRF_TWINS further workgroups solve the same tile redundantly and share the ten lower 16 x 16 tiles of X X^T with it.
#ifndef POTRF_TWINS
#define POTRF_TWINS 2
#endif
__host__ __device__ constexpr int potrf_twins(int Tn, int j) { return (Tn - j >= 2) ? POTRF_TWINS : 0; }
// A twin reads tile (j + 1, j) of the matrix at its start; the owner overwrites that tile with L at its end -- and nothing orders
// the start of one workgroup against the end of another (a twin may wait for a free CU).  So a twin counts itself into a word
// once its tile is in LDS, and the owner stores only when all twins have (it polls early: the answer is normally there long
// before it is needed; bounded, and a give-up is reported through `info` like any other, as -1).  Two words behind the three
// scratch tiles, by the step's parity; the diagonal block of step j clears the word of step j + 1.
constexpr int POTRF_SCRATCH = 3 * TB * TB + 64 + 64;   // doubles of scratch per factorisation chain
// ... [3 TB^2 + 64 + j]: sum of log L_cc over the 64 pivots of step j, written by the step's diagonal workgroup (1 / L_cc is in
// LDS there anyway): the sweep's closing kernel adds T numbers up instead of fetching 512 diagonal entries of each factor, one
// cache line apiece, and taking their logarithms on the sweep's critical path
constexpr int POTRF_LOGDET = 3 * TB * TB + 64;
__global__ void __launch_bounds__(PSTEP_THREADS) k_potrf_step(double* __restrict__ A, int ld, int j, int Tn, int* __restrict__ info,
                                                    int n_valid, double* __restrict__ scratch, double* __restrict__ Winv,
                                                    double* __restrict__ Sacc, const double* __restrict__ tv_xi,
                                                    double* __restrict__ tv_t, LamForm form) {
    // LDS: two MFMA operand panels (2 x 64 x PS) and two 64 x 64 tiles.  The panels stay valid while the diagonal tile is
    // factored: the waves that idle during the pivot runs use them for the block's own rank-64 update.
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    __shared__ __attribute__((aligned(16))) double tiles[2 * TB * LT];
    __shared__ double dprep[4 * DPB];                     // the diagonal tile's 16 x 16 blocks as solve16 reads them
    __shared__ double rinv[TB];
    TraceScope trace((form.trace_chain ? 16 : 40) + j);
    const bool xgroup = threadIdx.x >= 256;               // the solve group
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    {
        const int npot = (Tn - j) * (Tn - j + 1) / 2 + potrf_twins(Tn, j);   // this step's own tiles (and the twins of the
        if ((int)blockIdx.x >= npot) {                    // block below the diagonal); the workgroups beyond them work on
            int e = blockIdx.x - npot;                    // the inverse factor (winv_row_tile): finish block row j - 1,
            const int nfin = (j >= 2) ? 2 * (j - 1) : 0;  // then pre-accumulate block row j; Sigma = W^T W collects the
            const int npre = (j < Tn) ? nfin : 0;         // contribution of block row j - 2 (sigma_row_tile); and one
                                                          // workgroup advances the forward solve t = W (P xi) (tvec_role)
            // (not in the launch behind the last step, j == Tn: nothing hides that launch, and a 64^3 product per workgroup --
            // 64 dependent-issue MFMAs per wave, 2.8 us -- made its Sigma workgroups the last to leave, 7.7 us against the 5.1 of
            // the inverse factor's; the product launch behind it, whose workgroups are a quarter of the size, adds the last TWO
            // block rows instead)
            const int nsig = (Sacc && j >= 2 && j < Tn) ? (j - 1) * j / 2 : 0;
            if (e >= nfin + npre + nsig) { tvec_role<8>(A, Winv, tv_xi, tv_t, ld, j - 1, lds); return; }     // (all eight waves)
            if (xgroup) return;
            if (e < nfin) winv_row_tile(A, Winv, ld, j - 1, e >> 1, e & 1, lds, 2, tiles);
            else if (e < nfin + npre) { e -= nfin; winv_row_tile(A, Winv, ld, j, e >> 1, e & 1, lds, 1); }
            else if (e < nfin + npre + nsig) {
                int I, J;
                tile_from_index(e - nfin - npre, I, J);   // I >= J, I <= j - 2
                sigma_row_tile(Winv, Sacc, ld, j - 2, I, J, lds, tiles);
            }
            return;
        }
    }
    int a, b, tw = 0;                                     // tw > 0: a twin of the block (1, 0), see POTRF_TWINS
    {
        const int nown = (Tn - j) * (Tn - j + 1) / 2;
        if ((int)blockIdx.x >= nown) { a = 1; b = 0; tw = blockIdx.x - nown + 1; }
        else tile_from_index(blockIdx.x, a, b);           // a >= b, tile (j + a, j + b) of the matrix
    }
    const int i0 = (j + a) * TB, k0 = (j + b) * TB, j0 = j * TB;
    double* P0 = lds;
    double* P1 = lds + TB * PS;
    double* S = tiles;                                    // diagonal tile A_jj (panel column blocks only)
    double* X = tiles + TB * LT;                          // this block's own tile
    Acc4 accX;
    acc_zero(accX);
    const bool panel = (b == 0);
    // the diagonal workgroup of a chain that also wants W = L^-1 keeps its second half too: it inverts L_jj WHILE the first half
    // factors it (below) -- with the inverse as a phase of its own after the factorisation the diagonal workgroup was the last
    // to leave in half of the launches (in-kernel exit stamps: 21.8 us against the panel workgroups' 18.6), and the next step
    // waits for the whole launch (A/B on one box, 3 x 1000 sweeps each: 4021 against 3990 sweeps/s)
    const bool diag_inv = panel && a == 0 && Winv != nullptr;
    if (xgroup && !(panel && (a != 0 || diag_inv))) return;
    // What this workgroup's tile column starts from (see LamForm): formed here (do_form), already in A (load_old), or neither yet
    const bool lam = form.stats != nullptr;
    const int fstep = lam ? (int)form.form_step[j + b] : -1;
    const bool do_form = lam && fstep == j;
    const bool load_old = !lam || fstep < j || j >= 2;
    const bool xi_duty = lam && j == 0 && blockIdx.x == gridDim.x - 1;     // (step 0 has no extra workgroups)
    if (!lam && j == 0 && !panel) return;       // step 0 of a matrix that is already in A: no update to apply to the trailing tiles
    bool bypass = false;
    if (lam && form.col_words) {
        if (xi_duty && form.col_group[0] != 0xff) bypass = wait_stat_group(form, form.col_group[0]);   // B and the scalars arrive with the first group
        if (do_form && form.col_group[j + b] != 0xff) bypass = wait_stat_group(form, form.col_group[j + b]) || bypass;
        if (j == 0 && !do_form) {
            // step 0 of an overlapped sweep: this tile column has no statistics yet and there is no update to collect
            if (xi_duty) form_xi(form, ld, tid, bypass);
            return;
        }
        if (blockIdx.x == 0 && !xgroup) trace_mark(200 + j);
    }
    // scratch: tile 0 parks L_jj; tiles 1 and 2 (alternating with the step's parity: a late workgroup of step j + 1 may
    // still read one while step j + 1's owner of tile (j + 2, j + 1) writes the other) carry the next diagonal tile's
    // update L_{j+1,j} L_{j+1,j}^T from the workgroup that solved L_{j+1,j} to the next launch (syrk_slice / syrk_store)
    double* Dn_out = scratch + (size_t)(1 + (j & 1)) * TB * TB;
    const double* Dn_in = scratch + (size_t)(2 - (j & 1)) * TB * TB;
    // (the diagonal block moves the previous step's L_{j-1,j-1} from `scratch` into place: see move_prev below)
    auto move_prev = [&]() {
        const int q0 = (j - 1) * TB;
        for (int e = tid; e < TB * TB; e += 256) A[(size_t)(q0 + (e >> 6)) * ld + q0 + (e & 63)] = scratch[e];
    };
    // the tiles this block updates are fetched into registers now, so that their latency hides behind the MFMA phase
    TileRegs rX, rS, rD;
    if (lam && j == 0) stamp_enter(form.stamps);
    STEP_TRACE(0);
    if (panel) {
        // ---- a block of the panel column, the diagonal one included: two groups of four waves ----
        // (ONE copy of every inlined piece -- potf2_tile, the solve block, the tile formation -- for all of them: see potf2_tile
        // on what the code size of this kernel costs)
        long long* tw_words = reinterpret_cast<long long*>(scratch + 3 * TB * TB);
        if (!xgroup) {
            // factoring group: the diagonal tile A_jj (minus the rank-64 update the previous launch formed), factored by every
            // panel block itself -- no inter-block hand-off -- while the solve group works on the block's own tile
            if (a != 0 || diag_inv) __builtin_amdgcn_s_setprio(3);   // (the latency-bound group goes first where both want the same SIMD)
            tile_fetch(rS, do_form, load_old, A, form, ld, j0, j0, bypass);
            if (j > 0) {
                tile_g2r(rD, Dn_in, TB, 0, 0);
#pragma unroll
                for (int u = 0; u < 16; ++u) rS.v[u] -= rD.v[u];
            }
            if (xi_duty) form_xi(form, ld, tid, bypass);
            if (a == 0 && tid == 0) tw_words[(j + 1) & 1] = 0;       // the twins' word of the next step (see POTRF_SCRATCH)
            tile_r2s(S, rS);
            STEP_TRACE(1);
            __syncthreads();
            STEP_TRACE(2);
            if (a == 0 && j > 0 && !diag_inv) move_prev();            // (no second half in this block: nobody else to do it)
            // What this group's waves do while they have nothing to factor (potf2_tile's `idle`): wave 0 is free in every interval
            // behind its own pivot run -- a fifth worker of the panel solve (T0r2 in I4, T1r3 in I5, T2r3 in I6; the blocks the
            // solve group's waves 2 and 3 cannot take there, see below: they run two to a SIMD with a latency-bound solve of
            // that group, which costs neither much).  Waves 1 and 2 are free from I6 on: in the block below the diagonal and
            // its twins they form the workgroup's share of the next diagonal tile's update X X^T (tile t of the ten lower
            // 16 x 16 tiles belongs to workgroup t mod (1 + twins)): K-slice c once column block c of X is final in all rows.
            const bool has_solve = (a != 0) || diag_inv;
            const int ntwf = 1 + potrf_twins(Tn, j);
            int fcount = 0, fR[3], fC[3];
            d4 fg[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                fg[q] = (d4){0.0, 0.0, 0.0, 0.0};
                fR[q] = fC[q] = 0;
                const int t = tw + ntwf * ((wave - 1) + 2 * q);      // the q-th tile of wave 1 / wave 2
                if (a == 1 && (wave == 1 || wave == 2) && t < 10) { tile_from_index(t, fR[q], fC[q]); fcount = q + 1; }
            }
            auto fsyrk = [&](int c) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < fcount) syrk_slice(fg[q], X, fR[q], fC[q], c);
            };
            auto idle = [&](int iv) {
                if (!has_solve) return;
                if (wave == 0 && iv >= 4 && iv <= 6) trsm_interval(X, S, dprep, rinv, iv == 4 ? 2 : 3, iv - 4);
                if (iv == 6) { fsyrk(0); fsyrk(1); }
                else if (iv == 7) fsyrk(2);
            };
            potf2_tile(S, dprep, rinv, info, j0, n_valid, idle);
            STEP_TRACE(3);
            if (a == 0 && wave == 3) {                     // (the diagonal workgroup is not the one the launch waits for)
                double lg = -log(rinv[lane]);
                for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o);
                if (lane == 0) scratch[POTRF_LOGDET + j] = lg;
            }
            if (fcount > 0) {
                fsyrk(3);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < fcount) syrk_store(fg[q], Dn_out, fR[q], fC[q]);
            }
            if (a == 0) {
                // The diagonal block must not overwrite A_jj in place: the other panel blocks read it during this launch.  It parks
                // L_jj in `scratch`; the next step's diagonal block moves it into place.  The last step has no readers.
                if (j == Tn - 1) tile_s2g(S, A, ld, j0, j0);
                else tile_s2g(S, scratch, TB, 0, 0);                  // column-major tile
            }
            STEP_TRACE(4);
            return;
        }
        // solve group.  Below the diagonal: the block's own tile (j + a, j) -- its rank-64 update L_{i,j-1} L_{j,j-1}^T, then the
        // solve X L_jj^T = A_ij.  In the diagonal block (diag_inv): the same solve on the identity, W_jj = X^T = L_jj^-1 (row r of
        // X is zero left of column r -- exactly: every term is 0 * finite), instead of an inversion behind the factorisation.
        const bool below = (a != 0);
        if (below) {
            tile_fetch(rX, do_form, load_old, A, form, ld, i0, j0, bypass);
            if (j > 0) {
                const int p0 = (j - 1) * TB;
                load_panel_n(P0, A, ld, i0, p0, TB, tid);         // L_{i, j-1}
                load_panel_n(P1, A, ld, k0, p0, TB, tid);         // L_{j, j-1}
            }
            tile_r2s(X, rX);
        } else {
            for (int e = tid; e < TB * LT; e += 256) X[e] = (e / LT == e % LT) ? 1.0 : 0.0;
        }
        STEP_TRACE(1);
        __syncthreads();
        STEP_TRACE(2);
        // a twin counts itself in BEHIND the barrier: only then has EVERY wave's share of tile (j + 1, j) arrived in LDS (a wave
        // stores its registers as soon as its own loads are back; the owner overwrites the tile once the count is complete)
        if (below && tw > 0 && tid == 0)
            __hip_atomic_fetch_add((__attribute__((address_space(1))) long long*)(tw_words + (j & 1)), 1LL, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        // The diagonal block moves the previous step's L tile into place BEHIND its first barrier: the copy's loads miss every
        // cache, and in front of the barrier the factoring group waited 1 - 2 us for a tile nobody reads during this launch
        // (the step trace showed exactly that).  `scratch` is overwritten only after the factorisation, eight barriers on.
        if (!below && j > 0) move_prev();
        // SIMD-AWARE SCHEDULE.  Wave w of the factoring group and wave w of this group share a SIMD (tools/simd_map_probe.hip:
        // always, whatever the SIMD's number), and FP64 MFMAs and FP64 vector instructions share its pipe: a 64-cycle MFMA of
        // this group in front of a dependent FMA of the pivot chain delays the chain by all of it.  Measured (step trace, this
        // group's work compiled out): the factorisation takes 9.0 us alone and 13.3 with this group working in every interval.
        // So in each of the eight barrier intervals of potf2_tile the waves whose SIMD carries the critical work stay idle:
        //   I0 run 0 (factoring wave 0 pivots)          -> wave 0 idle
        //   I1 phase 0 (waves 1, 2, 3 solve + update)   -> waves 1, 2, 3 idle
        //   I2 run 1                                    -> wave 1 idle        I3 phase 1 (waves 2, 3)  -> waves 2, 3 idle
        //   I4 run 2                                    -> wave 2 idle        I5 phase 2 (wave 3), I6 run 3 -> wave 3 idle
        // and the work moves: the tile's own rank-64 update (four K-slices per wave, I0 .. I3) and the sixteen 16 x 16 blocks of
        // the triangular solve (I4 .. I7; X is in LDS, so any wave can take any row block) go where a SIMD is free.
        {
            // K-slices of the own update per wave and interval, a nibble each (immediates: a table in memory costs a scalar-cache
            // miss per interval): wave 0: 0 2 2 0, wave 1: 3 0 0 1, waves 2 and 3: 2 0 2 0
            const unsigned nslice = (wave == 0) ? 0x0220u : (wave == 1) ? 0x1003u : 0x0202u;
            int done = 0;
#pragma unroll 1
            for (int it = 0; it < 4; ++it) {
                if (below && j > 0) {
                    const int n = (nslice >> (4 * it)) & 15;
                    for (int q = 0; q < n; ++q) {
                        const int sl = done++;
                        const int li = lane & 15, lk = lane >> 4;
                        const double* ap = P0 + (16 * sl + lk) * PS + wr * 32 + li;
                        const double* bp = P1 + (16 * sl + lk) * PS + wc * 32 + li;
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const double a0 = ap[0], a1 = ap[16], b0 = bp[0], b1 = bp[16];
                            accX.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, accX.t[0][0], 0, 0, 0);
                            accX.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, accX.t[0][1], 0, 0, 0);
                            accX.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, accX.t[1][0], 0, 0, 0);
                            accX.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, accX.t[1][1], 0, 0, 0);
                            ap += 4 * PS; bp += 4 * PS;
                        }
                    }
                    if (done == 4) { tile_sub_acc(X, accX, lane, wr, wc); done = 5; }
                }
                STEP_TRACE(3 + 2 * it);
                __syncthreads();
                STEP_TRACE(4 + 2 * it);
            }
        }
        // I4 .. I7: column block it - 4 of the wave's own 16 rows -- except where the wave's SIMD carries the factorisation (wave 2
        // in I4, wave 3 in I5 and I6): those blocks are taken by wave 0 of the factoring group (see `idle` there), and wave 3
        // picks its rows up again at T3.
        const bool next = (a == 1);
        const int tw_need = (next && tw == 0) ? potrf_twins(Tn, j) : 0;      // the owner of the tile the twins read, see POTRF_SCRATCH
        bool tw_ready = true;
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            // (the owner looks at the twins' word early -- the answer is normally there long before it is needed, and a look costs
            // a trip to memory that would otherwise sit in front of the tile's store)
            if (it == 3 && tw_need > 0 && lane == 0) tw_ready = join_ready(tw_words + (j & 1), tw_need);
            const bool off = (wave == 2 && it == 0) || (wave == 3 && (it == 1 || it == 2));
            if (!off) trsm_interval(X, S, dprep, rinv, wave, it);
            STEP_TRACE(13 + 2 * it);
            __syncthreads();
            STEP_TRACE(14 + 2 * it);
        }
        STEP_TRACE(11);
        if (!below) tile_s2g_t(X, Winv, ld, j0, j0);
        else if (tw == 0) {
            // (the twins store nothing; the owner of the tile only once every twin has read what it overwrites, see POTRF_SCRATCH)
            if (tw_need > 0 && lane == 0 && !tw_ready) {
                int it = 0;
                while (!join_ready(tw_words + (j & 1), tw_need)) {
                    if (++it >= (1 << 21)) { atomicMin(info, -1); break; }   // (the twins never read it: the factor is not to be trusted)
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            tile_s2g(X, A, ld, i0, j0);
        }
        STEP_TRACE(12);
        return;
    }
    // ---- a trailing tile: A_ik -= L_{i,j-1} L_{k,j-1}^T, through LDS for coalesced global access ----
    // (the Lambda chain forms its tiles instead of loading them; step 0's last workgroup also writes xi)
    tile_fetch(rX, do_form, load_old, A, form, ld, i0, k0, bypass);
    if (xi_duty) form_xi(form, ld, tid, bypass);
    if (j > 0) {
        const int p0 = (j - 1) * TB;
        load_panel_n(P0, A, ld, i0, p0, TB, tid);         // L_{i, j-1}
        if (a != b) load_panel_n(P1, A, ld, k0, p0, TB, tid);   // L_{k, j-1}
        __syncthreads();
        tile_mma(accX, P0, (a != b) ? P1 : P0, TB, lane, wr, wc);
        __syncthreads();
    }
    tile_r2s(X, rX);
    __syncthreads();
    tile_sub_acc(X, accX, lane, wr, wc);
    __syncthreads();
    tile_s2g(X, A, ld, i0, k0);
}

// v[kk] = V[64 kb + kk][j] = W'[Qp-1-64kb-kk][Qp-1-j]: 64 contiguous doubles of column Qp-1-j of W' (descending), read
// straight from the inverse factor -- every lane its own 512-byte run, fully used, so no transposed copy is needed.
__device__ __forceinline__ void load_v_column(double (&v)[64], const double* __restrict__ Wp, int Qp, int kb, int j) {
    const double2* src = reinterpret_cast<const double2*>(Wp + (size_t)(Qp - 1 - j) * Qp + (Qp - 64 * (kb + 1)));
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const double2 w = src[u];
        v[63 - 2 * u] = w.x;
        v[62 - 2 * u] = w.y;
    }
}

// pass 2 of Uv, one wave per tile of the FULL tile grid: rows of tile (kb, jb) of Uv written into LR = Uv^T (zeros for
// kb > jb).  Runs as the extra workgroups of the Sigma = W'^T W' launch (k_gemm32 mode 0): neither needs the other.
struct UvArgs {
    const double* Wp;        // nullptr: no Uv role in this launch
    const double* p;
    const double* ck;
    const double* ak;
    const double* partial;
    double* LR;
    int64_t* stamps;
    // device-side join with the K_uu chain (side stream): the product workgroups read K_uu^-1 only in their epilogue and
    // wait there until *join >= join_need (set by k_join_set behind the chain's last kernel); nullptr = the caller joined
    // the streams with an event
    const long long* join;
    long long join_need;
    int spin_limit;          // polls before a late join gives up (its trace shares become NaN and SYNC_LATE_KINV is set)
    int* sync_status;
};
__global__ void k_join_wait(const long long* w, long long need, int spin_limit, int* sync_status, int bit, int trace_slot) {
    TraceScope trace(trace_slot);
    if (threadIdx.x == 0 && blockIdx.x == 0) spin_until(w, need, spin_limit, sync_status, bit);
}
__global__ void k_join_set(long long* w, long long v) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        __hip_atomic_store((__attribute__((address_space(1))) long long*)w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void uv_cols_role(const UvArgs& u, int Qp, int kb, int jb) {
    const int lane = threadIdx.x, j = 64 * jb + lane;
    double* out = u.LR + (size_t)(64 * kb) * Qp + j;
    if (kb > jb) {
#pragma unroll 16
        for (int kk = 0; kk < 64; ++kk) out[(size_t)kk * Qp] = 0.0;
        stamp_exit(u.stamps);
        return;
    }
    double v[64];
    load_v_column(v, u.Wp, Qp, kb, j);
    // rows below this tile, nearest first (fixed order).  Eight loads in flight per pass, unconditional (clamped index, the
    // surplus multiplied away): with a load per loop trip this one wave waited out up to Tn - 1 memory latencies in a row
    // and held the whole Sigma launch at 13 us (its product workgroups need 5).
    double T = 0.0;
    for (int b0 = jb; b0 > kb; b0 -= 8) {
        double pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int b = b0 - e;
            pv[e] = u.partial[(size_t)max(b, kb + 1) * Qp + j] * ((b > kb) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) T += pv[e];
    }
    // the tile's 64 coefficients: one coalesced load each (lane kk holds row 64 kb + kk's), handed out with v_readlane --
    // as uniform loads inside the loop they were 192 scalar loads whose latencies this single wave sat out one after another
    const double ckl = u.ck[64 * kb + lane], akl = u.ak[64 * kb + lane], pl = u.p[64 * kb + lane];
    auto lane_value = [](double x, int src) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src), __builtin_amdgcn_readlane(__double2loint(x), src));
    };
#pragma unroll
    for (int kk = 63; kk >= 0; --kk) {
        out[(size_t)kk * Qp] = fma(lane_value(ckl, kk), v[kk], lane_value(akl, kk) * T);
        T = fma(lane_value(pl, kk), v[kk], T);
    }
    stamp_exit(u.stamps);
}

// ------------------------------------------------------------------------------------------------
// Tile GEMM with 32 x 32 output per block (4 waves x one 16 x 16 MFMA tile), for M x M products where a 64 x 64 tiling
// leaves too few blocks for 256 CUs.
//   mode 0  C(I,J) = sum_{k >= I} W(k,I)^T W(k,J)   lower tiles, mirrored; optional index reversal   (Sigma = W^T W)
//   mode 3  C = A B                                 every tile                                     (theta gradient)
// (I, J, k are 64-tile indices; W lower triangular, so the k range skips the structural zeros.)
// ------------------------------------------------------------------------------------------------
// mode 0 extras (all nullable): with `mu` the kernel also writes R = C + mu mu^T (Sigma_v + mu mu^T, GPnode/UniSGPnode.jl:67)
// and, with `Psi2` (d_out = 1), the block's shares of the two traces of the :w rule / average energy (:196-238, :337-359):
// tr(R Psi2) into trace_part[blockIdx.x] and tr(Kuu^-1 Psi2) into trace_part[n + blockIdx.x], n = the launch's Tn (Tn + 1) / 2 * 4 product workgroups.
__global__ void __launch_bounds__(256) k_gemm32(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                int ld, int Tn, int mode, int s, int rev, const double* __restrict__ mu,
                                                double* __restrict__ R, const double* __restrict__ Psi2,
                                                const double* __restrict__ Kinv, double* __restrict__ trace_part, UvArgs uv,
                                                const double* __restrict__ Sacc) {
    __shared__ double As[64 * PS32];
    __shared__ double Bs[64 * PS32];
    __shared__ double tred[4];
    TraceScope trace(mode == 0 ? (mu ? 5 : 6) : 255);
    if (mode == 0 && uv.Wp) {                                            // extra workgroups: pass 2 of Uv (uv_cols_role)
        const int ngemm = Tn * (Tn + 1) / 2 * 4;
        if ((int)blockIdx.x >= ngemm) {
            if (threadIdx.x >= 64) return;
            const int e = blockIdx.x - ngemm;
            uv_cols_role(uv, ld, e % Tn, e / Tn);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int quad = blockIdx.x & 3, qi = quad >> 1, qj = quad & 1;
    int I, J, kbeg, kend;
    if (mode == 0) {
        tile_from_index(blockIdx.x >> 2, I, J);
        kbeg = Sacc ? max(I, Tn - 2) : I;                                // with Sacc only the last two block rows are left to add
        kend = Tn;
    } else {                                                             // mode 3: general C = A B, every tile
        const int t = blockIdx.x >> 2;
        I = t / Tn; J = t % Tn; kbeg = 0; kend = Tn;
    }
    const int r0 = I * TB + qi * 32, c0 = J * TB + qj * 32;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    const int li = lane & 15, lk = lane >> 4;
    if (mode == 0 && Sacc && I < Tn - 2) {                               // the block rows before the last two, collected by
#pragma unroll                                                           // sigma_row_tile during the factorisation
        for (int r = 0; r < 4; ++r) acc[r] = Sacc[(size_t)(c0 + wc * 16 + li) * ld + r0 + wr * 16 + lk + 4 * r];
    }
    // mode 0: the epilogue's operands (this thread's 4-row run of Psi2 / Kuu^-1 and its mu entries, see below) are fetched
    // now, so that their latency hides behind the product instead of following it
    const int oc = tid >> 3, o4 = (tid & 7) * 4;
    const int gcA = rev ? ld - 1 - (c0 + oc) : c0 + oc, grA = rev ? ld - 1 - (r0 + o4 + 3) : r0 + o4;
    const int gcB = rev ? ld - 1 - (r0 + oc) : r0 + oc, grB = rev ? ld - 1 - (c0 + o4 + 3) : c0 + o4;
    const size_t offA = (size_t)gcA * ld + grA, offB = (size_t)gcB * ld + grB;
    double2 p0 = make_double2(0.0, 0.0), p1 = p0, q0 = p0, q1 = p0;
    double muA[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, muB[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    // K_uu^-1 comes from the side stream's chain.  Normally that finished long ago: one poll, and the operands are fetched
    // now like the others.  If not, the product goes first and the workgroup waits in front of its epilogue.
    bool kinv_late = false;
    if (mode == 0 && mu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { muA[e] = mu[grA + e]; muB[e] = mu[grB + e]; }
        muA[4] = mu[gcA];
        muB[4] = mu[gcB];
        if (Psi2) {
            p0 = *reinterpret_cast<const double2*>(Psi2 + offA); p1 = *reinterpret_cast<const double2*>(Psi2 + offA + 2);
            kinv_late = uv.join && !join_ready(uv.join, uv.join_need);
            if (!kinv_late) { q0 = *reinterpret_cast<const double2*>(Kinv + offA); q1 = *reinterpret_cast<const double2*>(Kinv + offA + 2); }
        }
    }
    for (int k = kbeg; k < kend; ++k) {
        __syncthreads();
        if (mode == 0) load_panel32_t(As, A, ld, k * TB, r0, tid);      // As[kk][i] = W[k*64+kk, r0+i]
        else load_panel32_n(As, A, ld, r0, k * TB, tid);                  // As[kk][i] = A[r0+i, k*64+kk]
        load_panel32_t(Bs, B, ld, k * TB, c0, tid);                       // Bs[kk][j] = B[k*64+kk, c0+j]
        __syncthreads();
        const double* ap = As + lk * PS32 + wr * 16 + li;
        const double* bp = Bs + lk * PS32 + wc * 16 + li;
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k4 * 4 * PS32], bp[k4 * 4 * PS32], acc, 0, 0, 0);
    }
    double tsum = 0.0, tsumK = 0.0;
    if (mode != 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            C[(size_t)(c0 + wc * 16 + li) * ld + r0 + wr * 16 + lk + 4 * r] = acc[r];
        return;
    }
    // mode 0 epilogue through LDS: the accumulator layout scatters a wave's store over 16 columns (32-byte pieces of 16
    // different lines); staged as a 32 x 32 quadrant, every thread owns 4 consecutive rows of one column, 8 threads a
    // 256-byte run -- for the quadrant itself and, with rows and columns exchanged, for its mirror image.  (The direct form
    // held this launch at 15 us for 9 MB of traffic.)
    __syncthreads();
    double* Qd = As;                                                     // Qd[c][r], stride 33
#pragma unroll
    for (int r = 0; r < 4; ++r) Qd[(wc * 16 + li) * 33 + wr * 16 + lk + 4 * r] = acc[r];
    __syncthreads();
    {   // the quadrant: column c0 + oc, rows r0 + o4 .. + 3 (index-reversed: a descending, still contiguous run)
        double v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = Qd[oc * 33 + o4 + (rev ? 3 - e : e)];
        *reinterpret_cast<double2*>(C + offA) = make_double2(v[0], v[1]);
        *reinterpret_cast<double2*>(C + offA + 2) = make_double2(v[2], v[3]);
        if (mu) {
            double rv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rv[e] = fma(muA[e], muA[4], v[e]);
            *reinterpret_cast<double2*>(R + offA) = make_double2(rv[0], rv[1]);
            *reinterpret_cast<double2*>(R + offA + 2) = make_double2(rv[2], rv[3]);
            if (Psi2) {
                tsum = fma(rv[0], p0.x, fma(rv[1], p0.y, fma(rv[2], p1.x, rv[3] * p1.y)));
                if (kinv_late) {                                         // the K_uu chain was still running at entry
                    if (spin_until(uv.join, uv.join_need, uv.spin_limit, uv.sync_status, SYNC_LATE_KINV)) {
                        // (bypassing this XCD's L2: lines of the previous sweep's K_uu^-1 may sit there -- the kernel-start
                        // invalidate came before the chain's write-back)
                        const __attribute__((address_space(1))) double* kq = (const __attribute__((address_space(1))) double*)(Kinv + offA);
                        q0.x = __hip_atomic_load(kq + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        q0.y = __hip_atomic_load(kq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        q1.x = __hip_atomic_load(kq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        q1.y = __hip_atomic_load(kq + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else
                        q0.x = __builtin_nan("");                        // gave up: the trace (and the energy) come out NaN
                }
                tsumK = fma(q0.x, p0.x, fma(q0.y, p0.y, fma(q1.x, p1.x, q1.y * p1.y)));
            }
        }
    }
    {   // the mirror image: column r0 + oc, rows c0 + o4 .. + 3 (on diagonal tiles two quadrants write the same bits twice)
        double v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = Qd[(o4 + (rev ? 3 - e : e)) * 33 + oc];
        *reinterpret_cast<double2*>(C + offB) = make_double2(v[0], v[1]);
        *reinterpret_cast<double2*>(C + offB + 2) = make_double2(v[2], v[3]);
        if (mu) {
            double rv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rv[e] = fma(muB[e], muB[4], v[e]);
            *reinterpret_cast<double2*>(R + offB) = make_double2(rv[0], rv[1]);
            *reinterpret_cast<double2*>(R + offB + 2) = make_double2(rv[2], rv[3]);
        }
    }
    if (trace_part) {
        // off-diagonal tiles stand for both mirror images; on diagonal tiles all four quadrants are computed
        if (I != J) { tsum *= 2.0; tsumK *= 2.0; }
        tsum = block_sum(tsum, tred);
        tsumK = block_sum(tsumK, tred);
        if (tid == 0) {
            trace_part[blockIdx.x] = tsum;
            trace_part[Tn * (Tn + 1) / 2 * 4 + blockIdx.x] = tsumK;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mu = Sigma xi (one wave per row, Sigma symmetric so a row is read as a contiguous column)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_symv(const double* __restrict__ S, const double* __restrict__ x,
                                              double* __restrict__ y, int n, int ld) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s = fma(S[(size_t)row * ld + j], x[j], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = s;
}

// ------------------------------------------------------------------------------------------------
// Uv = chol(Sigma_v + mu mu^T).U  (GPnode/UniSGPnode.jl:67-69) WITHOUT a third factorisation and without a sequential
// rank-1 update.  Lambda was factored in index-reversed order, P Lambda P = L' L'^T, so with W' = L'^-1
//     V = P W' P  is upper triangular with V^T V = Sigma_v   (V = chol(Sigma_v).U),   and   V^-1 = P L' P.
// Write R = Sigma_v + mu mu^T = V^T (I + p p^T) V with p = V^-T mu.  The Cholesky factor of identity-plus-rank-one is
// known in closed form: with alpha_0 = 1, alpha_{k+1} = alpha_k + p_k^2,
//     C_kk = sqrt(alpha_{k+1} / alpha_k),   C_kj = p_k p_j / sqrt(alpha_k alpha_{k+1})  (j > k),   C^T C = I + p p^T,
// hence  Uv = C V,   Uv[k][j] = C_kk V[k][j] + (p_k / sqrt(alpha_k alpha_{k+1})) * sum_{m=k+1..j} p_m V[m][j].
// Everything is parallel over the columns j; the sum is a running suffix sum down each column (no cancellation:
// it is accumulated directly, not as x_j minus a prefix).
// p costs nothing extra: mu = Sigma xi = P W'^T W' P xi, so with t = W' (P xi) one has mu = P W'^T t and
// p = P L'^T P mu = P L'^T W'^T t = P t.
//   tvec_role      : t = W' (P xi), block by block during the factorisation
//   k_trmv_mu_scan : mu = P W'^T t, p = P t, alpha scan -> C_kk and p_k / sqrt(alpha_k alpha_{k+1})
//   k_uv_partial : per 64-row tile and column, sum_m p_m V[m][j]   (so that tiles can start their suffix sums independently)
//   k_uv_cols    : one wave per 64 x 64 tile, 64 rows in registers, writes the rows of Uv
// V[k][j] = W'[Qp-1-k][Qp-1-j] is read straight from the inverse factor (load_v_column).
// Output LR = Uv^T (lower, column-major: column k = row k of Uv), the layout potrf(R) would have produced.
// ------------------------------------------------------------------------------------------------
constexpr int CU_MAXQ = 4096;

// mu = P W'^T t (one wave per column of W'), p = P t, and -- in the extra last workgroup -- the alpha scan of p:
// ck[k] = C_kk, ak[k] = p_k / sqrt(alpha_k alpha_{k+1}).
__global__ void __launch_bounds__(256) k_trmv_mu_scan(const double* __restrict__ W, const double* __restrict__ tpart /* t */,
                                                      double* __restrict__ mu, double* __restrict__ p,
                                                      double* __restrict__ ck, double* __restrict__ ak,
                                                      double* __restrict__ uvpart, int Qp) {
    __shared__ double ts[CU_MAXQ];
    TraceScope trace(4);
    const int lane = threadIdx.x & 63;
    // mat-vec workgroups: the first eight entries of the wave's column of W' are requested before the LDS copy of t is
    // waited for -- one memory round trip for both instead of two (this launch sits alone on the critical path)
    const bool matvec = (int)blockIdx.x < Qp / 4;
    const int kcol = blockIdx.x * 4 + (threadIdx.x >> 6);
    double wv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (matvec) {
        const double* col = W + (size_t)kcol * Qp;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = kcol + lane + 64 * u;
            wv[u] = col[min(i, Qp - 1)] * ((i < Qp) ? 1.0 : 0.0);
        }
    }
    for (int i = threadIdx.x; i < Qp; i += 256) ts[i] = tpart[i];     // t = W' (P xi), advanced block by block by tvec_role
    __syncthreads();
    if ((int)blockIdx.x > Qp / 4) {
        // workgroups beyond the mat-vec and the scan: pass 1 of Uv, one wave per 64 x 64 tile (kb <= jb) of V:
        //   partial[kb][j] = sum_kk p_{64 kb + kk} V[64 kb + kk][j],  p = P t straight from the LDS copy of t
        if (threadIdx.x >= 64) return;
        int jb, kb;
        tile_from_index(blockIdx.x - Qp / 4 - 1, jb, kb);            // jb >= kb
        const int j = 64 * jb + lane;
        double v[64];
        load_v_column(v, W, Qp, kb, j);
        double sacc = 0.0;
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) sacc = fma(ts[Qp - 1 - (64 * kb + kk)], v[kk], sacc);
        uvpart[(size_t)kb * Qp + j] = sacc;
        return;
    }
    if ((int)blockIdx.x == Qp / 4) {
        if (threadIdx.x >= 64) return;
        const int per = (Qp + 63) / 64;
        const int e0 = lane * per, e1 = min((lane + 1) * per, Qp);
        double loc = 0.0;
        for (int e = e0; e < e1; ++e) { double v = ts[Qp - 1 - e]; p[e] = v; loc = fma(v, v, loc); }
        double inc = loc;
        for (int o = 1; o < 64; o <<= 1) {
            double u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        double alpha = 1.0 + (inc - loc);
        for (int e = e0; e < e1; ++e) {
            const double pe = ts[Qp - 1 - e], an = fma(pe, pe, alpha);
            const double ir = 1.0 / sqrt(alpha * an);
            ck[e] = an * ir;
            ak[e] = pe * ir;
            alpha = an;
        }
        return;
    }
    const int k = kcol;
    const double* col = W + (size_t)k * Qp;
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s = fma(wv[u], ts[min(k + lane + 64 * u, Qp - 1)], s);
    for (int i = k + lane + 512; i < Qp; i += 64) s = fma(col[i], ts[i], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) mu[Qp - 1 - k] = s;
}

// ------------------------------------------------------------------------------------------------
// Scalars of the sweep.  Two passes, fixed summation order (bitwise reproducible):
//   k_trace_partial: block g sums its columns of   slot 0: Kinv o Psi2  (tr(Kuu^-1 Psi2): sum I1 = s_kk - t1)
//                                                  slot 1 + a + b d_out: Rblk[a][b] o Psi2   (tr(R Psi2) / Psi4)
//   k_scalars: UniSGP  sum I2 = s_yy - 2 b^T mu + t2 ;  energy = 0.5 [ w (sum I1 + sum I2) - n E_logw + n log 2 pi ]
//                      (GPnode/UniSGPnode.jl:196-238, 337-359, 411-436 summed over the nodes)
//              MultiSGP inverse scale S = I1 + Ryy - EY - EY^T + Psi4 (GPnode/MultiSGPnode.jl:391-404),
//                      energy = 0.5 [ tr(W S) - n E_logdetW + n d_out log 2 pi ]   (GPnode/MultiSGPnode.jl:544-632)
// ------------------------------------------------------------------------------------------------
constexpr int TRACE_BLOCKS = 64;
constexpr int TRACE_SLOTS = 1 + MAXO * MAXO;


// partial[g] = block g's share of tr(Kuu^-1 Psi2)
__global__ void __launch_bounds__(256) k_trace_kinv(const double* __restrict__ stats, const double* __restrict__ Kinv,
                                                    double* __restrict__ partial, int M, int Mp) {
    __shared__ double red[4];
    const double* Psi2 = stats;
    const int tid = threadIdx.x;
    double t1 = 0.0;
    for (int j = blockIdx.x; j < M; j += gridDim.x)
        for (int i = tid; i < M; i += 256) t1 = fma(Kinv[(size_t)j * Mp + i], Psi2[(size_t)j * Mp + i], t1);
    t1 = block_sum(t1, red);
    if (tid == 0) partial[blockIdx.x] = t1;
}

// partial[g * d_out^2 + a + b d_out] = block g's share of tr(Rblk[a][b] Psi2)   (MultiSGP, and the theta objective at a
// new theta; the UniSGP sweep gets this trace from the epilogue of the Sigma = W^T W product instead)
__global__ void __launch_bounds__(256) k_trace_R(const double* __restrict__ stats, const double* __restrict__ R,
                                                 double* __restrict__ partial, int M, int Mp, int d_out, int Qp,
                                                 int64_t* stamps) {
    __shared__ double red[4];
    stamp_enter(stamps);
    const double* Psi2 = stats;
    const int tid = threadIdx.x;
    for (int a = 0; a < d_out; ++a)
        for (int b = 0; b < d_out; ++b) {
            double t2 = 0.0;
            for (int j = blockIdx.x; j < M; j += gridDim.x)
                for (int i = tid; i < M; i += 256)
                    t2 = fma(R[(size_t)(b * M + j) * Qp + a * M + i], Psi2[(size_t)j * Mp + i], t2);
            t2 = block_sum(t2, red);
            if (tid == 0) partial[(size_t)blockIdx.x * d_out * d_out + a + b * d_out] = t2;
        }
}

__global__ void __launch_bounds__(256) k_scalars(const double* __restrict__ stats, const double* __restrict__ partK, int nK,
                                                 const double* __restrict__ partR, int nR,
                                                 const double* __restrict__ mu, const double* __restrict__ Lkuu,
                                                 const double* __restrict__ Llam, const int* __restrict__ info,
                                                 const Params* __restrict__ P, double* __restrict__ out,
                                                 double* __restrict__ wishart, int M, int Mp, int d_out, int Q, int Qp,
                                                 int lam_off, int64_t* stamps, int64_t* all_stamps,
                                                 int64_t* totals, long long* done_word, long long done_value,
                                                 const double* __restrict__ ldK, int nldK, const double* __restrict__ ldL, int nldL,
                                                 double* __restrict__ mirror) {
    // mirror (may be nullptr): SGP_R_COUNT + 2 doubles of PINNED HOST memory that receive out[], the hand-off status word info[3]
    // and done_value -- what sgp_get_scalars / sgp_w_stats read after their wait instead of two blocking copies (~25 us apiece)
    // ldK / ldL (may be nullptr: then the diagonals of Lkuu / Llam are read): per-step sums of log L_cc left by the factorisation
    // launches (POTRF_LOGDET)
    // done_word (may be nullptr): set to done_value once this kernel -- the sweep's last reader of the K_uu chain's outputs --
    // has read them; the next sweep's chain waits for it in its first kernel (k_prep_xu) instead of on an event, which
    // between two kernels of this stream cost ~6 us of idle time
    __shared__ double red[4];
    __shared__ double redn[4 * 5];
    __shared__ double tr[TRACE_SLOTS];
    TraceScope trace(7);
    stamp_enter(stamps);
    const double* B = stats + (size_t)Mp * Mp;
    const double* sc = B + (size_t)Mp * d_out;
    const int tid = threadIdx.x;
    if (d_out == 1) {
        // UniSGP: every thread's share of all five sums first (independent loads, one memory round trip), then ONE
        // workgroup reduction -- this kernel is the last link of the critical path.  The last wave also fetches and folds
        // the sweep's phase stamps now, so that closing them at the end is stores only.
        StampFold fold = {0, 0, 0};
        if (all_stamps && tid >= 192) fold = stamp_fold_load(all_stamps, totals, tid - 192);
        // what the closing thread needs besides the five sums, fetched with everything else (not after the reduction)
        const double c_syy = sc[0], c_skk = sc[1], c_n = sc[2], c_w = P->W[0], c_sigma2 = P->sigma2, c_elogw = P->E_logw;
        const int c_i0 = info[0], c_i1 = info[1], c_i2 = info[2], c_i3 = mirror ? info[3] : 0;
        double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};         // tr(Kuu^-1 Psi2), tr(R Psi2), log|L_K|, log|L_Lambda|, b' mu
        for (int b = tid; b < nK; b += 256) v[0] += partK[b];
        for (int b = tid; b < nR; b += 256) v[1] += partR[b];
        for (int e = tid; e < M; e += 256) {
            if (!ldK) v[2] += log(Lkuu[(size_t)e * Mp + e]);
            v[4] = fma(B[e], mu[e], v[4]);
        }
        if (ldK) for (int e = tid; e < nldK; e += 256) v[2] += ldK[e];
        if (ldL) for (int e = tid; e < nldL; e += 256) v[3] += ldL[e];
        else for (int e = tid; e < Q; e += 256) v[3] += log(Llam[(size_t)(e + lam_off) * Qp + e + lam_off]);
#pragma unroll
        for (int i = 0; i < 5; ++i)
            for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
        if ((tid & 63) == 0)
#pragma unroll
            for (int i = 0; i < 5; ++i) redn[(tid >> 6) * 5 + i] = v[i];
        __syncthreads();
        if (tid == 0) {
            double t[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) t[i] = (redn[i] + redn[5 + i]) + (redn[10 + i] + redn[15 + i]);
            const double LOG2PI = 1.8378770664093454835606594728112;
            const double n = c_n, w = c_w;
            const double sum_I1 = c_sigma2 * c_skk - t[0];
            const double sum_I2 = c_syy - 2.0 * t[4] + t[1];
            out[0] = sum_I1;
            out[1] = sum_I2;
            out[2] = 0.5 * (w * (sum_I1 + sum_I2) - n * c_elogw + n * LOG2PI);
            out[3] = (double)c_i0;
            out[4] = (double)c_i1;
            out[5] = (double)c_i2;
            out[6] = 2.0 * t[2];
            out[7] = 2.0 * t[3];
            if (mirror) {
                mirror[0] = sum_I1; mirror[1] = sum_I2; mirror[2] = out[2]; mirror[3] = (double)c_i0; mirror[4] = (double)c_i1;
                mirror[5] = (double)c_i2; mirror[6] = 2.0 * t[2]; mirror[7] = 2.0 * t[3];
                mirror[SGP_R_COUNT] = (double)c_i3; mirror[SGP_R_COUNT + 1] = (double)done_value;
            }
            if (done_word)
                __hip_atomic_store((__attribute__((address_space(1))) long long*)done_word, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        stamp_exit(stamps);
        if (all_stamps && tid >= 192) stamp_fold_finish(fold, all_stamps, totals, tid - 192);
        return;
    }
    {   // wave w reduces slots w, w + 4, ... (slot 0: tr(Kuu^-1 Psi2), slot 1 + a + b d_out: tr(Rblk[a][b] Psi2)):
        // the lanes stride over the block partials in a fixed order, then a butterfly sum
        const int lane = tid & 63, wave = tid >> 6, nslotR = d_out * d_out;
        for (int slot = wave; slot < 1 + nslotR; slot += 4) {
            double v = 0.0;
            if (slot == 0) { for (int b = lane; b < nK; b += 64) v += partK[b]; }
            else           { for (int b = lane; b < nR; b += 64) v += partR[(size_t)b * nslotR + slot - 1]; }
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) tr[slot] = v;
        }
    }
    __syncthreads();
    const double s_kk = P->sigma2 * sc[1];
    const double n = sc[2];
    const double sum_I1 = s_kk - tr[0];
    double ld_k = 0.0, ld_l = 0.0;
    if (ldK) { for (int e = tid; e < nldK; e += 256) ld_k += ldK[e]; }
    else for (int e = tid; e < M; e += 256) ld_k += log(Lkuu[(size_t)e * Mp + e]);
    if (ldL) { for (int e = tid; e < nldL; e += 256) ld_l += ldL[e]; }
    else for (int e = tid; e < Q; e += 256) ld_l += log(Llam[(size_t)(e + lam_off) * Qp + e + lam_off]);
    ld_k = 2.0 * block_sum(ld_k, red);
    ld_l = 2.0 * block_sum(ld_l, red);
    const double LOG2PI = 1.8378770664093454835606594728112;
    if (d_out == 1) {
        double bmu = 0.0;
        for (int e = tid; e < M; e += 256) bmu = fma(B[e], mu[e], bmu);
        bmu = block_sum(bmu, red);
        if (tid == 0) {
            const double sum_I2 = sc[0] - 2.0 * bmu + tr[1];
            const double w = P->W[0];
            out[0] = sum_I1;
            out[1] = sum_I2;
            out[2] = 0.5 * (w * (sum_I1 + sum_I2) - n * P->E_logw + n * LOG2PI);
        }
    } else {
        const double* Ryy = sc + 8;
        double tr_WS = 0.0;
        for (int a = 0; a < d_out; ++a)
            for (int b = 0; b < d_out; ++b) {
                double ey_ab = 0.0, ey_ba = 0.0;             // EY[a][b] = mu^(b) . B[:, a]
                for (int e = tid; e < M; e += 256) {
                    ey_ab = fma(mu[b * M + e], B[(size_t)a * Mp + e], ey_ab);
                    ey_ba = fma(mu[a * M + e], B[(size_t)b * Mp + e], ey_ba);
                }
                ey_ab = block_sum(ey_ab, red);
                ey_ba = block_sum(ey_ba, red);
                double Sab = (a == b ? sum_I1 : 0.0) + Ryy[a + b * d_out] - ey_ab - ey_ba + tr[1 + a + b * d_out];
                if (tid == 0) wishart[a + b * d_out] = Sab;
                tr_WS += P->W[b + a * d_out] * Sab;
            }
        if (tid == 0) {
            out[0] = sum_I1;
            out[1] = 0.0;
            out[2] = 0.5 * (tr_WS - n * P->E_logw + n * d_out * LOG2PI);
        }
    }
    if (tid == 0) {
        out[3] = (double)info[0];
        out[4] = (double)info[1];
        out[5] = (double)info[2];
        out[6] = ld_k;
        out[7] = ld_l;
        if (mirror) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mirror[i] = out[i];
            mirror[SGP_R_COUNT] = (double)info[3]; mirror[SGP_R_COUNT + 1] = (double)done_value;
        }
    }
    stamp_exit(stamps);
    // last kernel of a sweep: close the sweep stamp and add this sweep's phase durations to the running totals
    __syncthreads();                                            // every wave's exit stamp of this kernel is in
    if (done_word && tid == 0)
        __hip_atomic_store((__attribute__((address_space(1))) long long*)done_word, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (all_stamps && tid < 64) stamp_accumulate(all_stamps, totals, tid);
}

// ------------------------------------------------------------------------------------------------
// Prediction: mean[s, o] = sum_m K(x*_s, u_m) mu[o*M + m]   (GPnode/UniSGPnode.jl:96-104 batched).
// One thread per test point, Xus tile and mu staged in LDS.
// ------------------------------------------------------------------------------------------------
template <int DT>   // DT > 0: input dimension known at compile time (x stays in registers); DT = 0: runtime D <= MAXD
__global__ void __launch_bounds__(256) k_predict(const double* __restrict__ Xus, const double* __restrict__ Xs,
                                                 const double* __restrict__ mu, double* __restrict__ mean,
                                                 const Params* __restrict__ P, int M, int Mp, int Drt, int64_t NS, int d_out) {
    __shared__ double us[MAXD * TB];
    __shared__ double ms[MAXO * TB];
    constexpr int DA = DT > 0 ? DT : MAXD;
    const int D = DT > 0 ? DT : Drt;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double x[DA];
#pragma unroll
    for (int d = 0; d < DA; ++d) x[d] = (s < NS && d < D) ? Xs[(size_t)s * D + d] * P->inv_ell[d] : 0.0;
    double acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = 0.0;
    for (int m0 = 0; m0 < M; m0 += TB) {
        __syncthreads();
        for (int t = threadIdx.x; t < D * TB; t += 256) us[t] = Xus[(size_t)(t / TB) * Mp + m0 + (t % TB)];
        for (int t = threadIdx.x; t < d_out * TB; t += 256) {
            int o = t / TB, m = m0 + (t % TB);
            ms[t] = (m < M) ? mu[o * M + m] : 0.0;
        }
        __syncthreads();
        const int mcount = (M - m0 < TB) ? (M - m0) : TB;
        for (int m = 0; m < mcount; ++m) {
            double d2 = 0.0;
#pragma unroll
            for (int d = 0; d < DA; ++d)
                if (d < D) { double t = x[d] - us[d * TB + m]; d2 = fma(t, t, d2); }
            double k = exp(-0.5 * d2);
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o < d_out) acc[o] = fma(k, ms[o * TB + m], acc[o]);
        }
    }
    if (s < NS)
        for (int o = 0; o < d_out; ++o) mean[(size_t)o * NS + s] = P->sigma2 * acc[o];
}

// generic K(A, B): na x nb column-major
__global__ void __launch_bounds__(256) k_kernelmatrix(const double* __restrict__ A, const double* __restrict__ B,
                                                      double* __restrict__ K, const Params* __restrict__ P,
                                                      int64_t na, int64_t nb, int D) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= na * nb) return;
    const int64_t i = e % na, j = e / na;
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) {
        double t = (A[(size_t)i * D + d] - B[(size_t)j * D + d]) * P->inv_ell[d];
        d2 = fma(t, t, d2);
    }
    K[e] = P->sigma2 * exp(-0.5 * d2);
}

// ------------------------------------------------------------------------------------------------
// Per-point :w quantities (GPnode/UniSGPnode.jl:196-238) from the resident K_uf:
//   out[n] = sum_m (F K_uf)[m, n]^2  with F lower-triangular (F = L^-1 -> |alpha_n|^2) or, with F = L_R^T
//   replaced by its transpose product, |Uv k_n|^2 = k_n^T R k_n = |L_R^T k_n|^2.
// Tile: 64 rows of F K x 64 points per block, K loop over block columns of F; squared column sums are
// reduced over the row-blocks by atomics-free two-pass (partial[rowblk][n]).
//   first factor used as stored (lower, rows i, sum over k <= i);  second factor transposed (upper), sum over k >= i.
// ------------------------------------------------------------------------------------------------
// ONE launch for both quadratic forms (round 4; round 3 launched the body once per factor and a third kernel re-read K_uf for
// k_n . mu): workgroup (point block nb, row tile I) walks T + 1 tile products --
//     steps 0 .. I   :  rows I of  W_K K_uf   (W_K = L_K^-1, lower: tile columns k <= I)        -> a = sum of squares down the column
//     steps I .. T-1 :  rows I of  Uv  K_uf   (Uv upper = LR^T: tile columns k >= I)            -> b
// (round 4a: both step ranges in ONE workgroup -- equal work, the K_uf tile of k = I staged once --; round 4b: one workgroup per
// factor's part, largest first, see the grid map below: equal items quantise badly on 512 slots.)  The column sums of squares of a
// part leave as two partial rows (one per wave row, summed in fixed order by k_w_point_finish -- no LDS beyond the panels, so
// that two workgroups share a CU, and no atomics).  The workgroup with I == nb mod T also forms k_n . mu from the K_uf tiles it
// stages anyway (four partial rows, one per wave).
// Software-pipelined as before: the next tile pair is fetched into registers while the matrix cores work on the current one;
// the triangular mask of the diagonal tile is applied in registers on the way into LDS; panels are stored as they arrive (see
// lstore_a / lstore_b).
//   pa, pb : [2 T][N] partial column sums (row 2 I + wr),  kmu : [4][N]
__global__ void __launch_bounds__(256, 2) k_quadform_fused(const double* __restrict__ Wk, const double* __restrict__ LR,
                                                        const double* __restrict__ Kuf, const double* __restrict__ mu,
                                                        double* __restrict__ pa, double* __restrict__ pb, double* __restrict__ kmu,
                                                        int ld, int T, int64_t N) {
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    constexpr int KS = KMS;                           // row stride of a [row][kk] panel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    // grid.y = 2 T work items per point block, LARGEST FIRST (workgroups are dispatched x fastest, then y: the long items of all point
    // blocks start first and the short ones fill the CUs as they drain -- list scheduling by the hardware's own dispatcher):
    //     y = 2 c     : the first factor's part of row tile I = T - 1 - c   (steps k = 0 .. I:      T - c products)
    //     y = 2 c + 1 : the second factor's part of row tile I = c           (steps k = I .. T - 1:  T - c products)
    // As ONE item per (point block, row tile) -- T + 1 products each, the K_uf tile of k = I staged once for both factors -- the
    // 1 256 equal workgroups of T needed three rounds on the 512 slots for 2.45 rounds' worth of work (PMC: 1.59 of 2 waves per SIMD
    // resident on average).
    const int cls = blockIdx.y >> 1, part_b = blockIdx.y & 1;
    const int I = part_b ? cls : T - 1 - cls;
    const int s_begin = part_b ? I + 1 : 0, s_end = part_b ? T : I;      // steps as before: s <= I first factor (k = s), s > I second (k = s - 1)
    const int64_t n0 = (int64_t)blockIdx.x * TB;
    double* As = lds;
    double* Bs = lds + TB * PS;
    Acc4 acc;
    acc_zero(acc);
    // staging maps, 4 passes of 256 threads each.  A of the first factor (W_K as stored: contiguous along i): thread -> (kk = t >> 4,
    // rows 4 (t & 15) ..); A of the second (LR^T: contiguous along kk) and B (K_uf columns: contiguous along kk): thread ->
    // (i or j = t >> 4, kk = 4 (t & 15) ..)
    double2 ra[4][2], rb[4][2];
    // (ONE copy of the loads, the LDS stores and the MFMA loop for both factors, the factor a run-time flag: with a copy per
    // factor the kernel needed 260 registers -- one workgroup per CU instead of two)
    auto gload_a = [&](int second, int k) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, hi = t >> 4, lo4 = (t & 15) * 4;
            const double* sa = second ? LR + (size_t)(I * TB + hi) * ld + k * TB + lo4      // LR[k*64 + lo4 .., I*64 + hi]
                                      : Wk + (size_t)(k * TB + hi) * ld + I * TB + lo4;     // W_K[I*64 + lo4 .., k*64 + hi]
            ra[u][0] = *reinterpret_cast<const double2*>(sa);
            ra[u][1] = *reinterpret_cast<const double2*>(sa + 2);
        }
    };
    auto gload_b = [&](int k) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, hi = t >> 4, lo4 = (t & 15) * 4;
            const int64_t n = n0 + hi;
            const double* sb = Kuf + (size_t)(n < N ? n : N - 1) * ld + k * TB + lo4;     // (clamped: the surplus is zeroed below)
            rb[u][0] = *reinterpret_cast<const double2*>(sb);
            rb[u][1] = *reinterpret_cast<const double2*>(sb + 2);
            if (n >= N) { rb[u][0] = make_double2(0.0, 0.0); rb[u][1] = make_double2(0.0, 0.0); }
        }
    };
    // Panels go into LDS the way they arrive (round 4b): an operand whose global reads run along the contraction index kk -- K_uf's
    // columns, LR^T -- is stored [row][kk] with a row stride of KS = 66 doubles (a thread's four kk = two 16-byte stores; the MFMA
    // operand read, 16 lanes = 16 rows at one kk, then hits 16 different bank pairs: 66 x 2 dwords = 4 mod 64), the first factor
    // (W_K as stored: contiguous along i) stays [kk][i] with stride PS.  Before, the kk-contiguous panels were TRANSPOSED on the
    // way in, four 8-byte stores per 32 bytes with an XOR swizzle: PMC counted 58 % of the LDS's active cycles as bank conflicts
    // (profiles/r04_ab_log.txt [21]).
    // first factor: As[kk = hi][i = lo4 + q] (lower: kk <= i); second: As[i = hi][kk = lo4 + q] (i <= kk) -- in both the entries to
    // drop from the DIAGONAL tile are those with lo4 + q < hi
    auto lstore_a = [&](int second, bool dg) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, hi = t >> 4, lo4 = (t & 15) * 4;
            double va[4] = {ra[u][0].x, ra[u][0].y, ra[u][1].x, ra[u][1].y};
#pragma unroll
            for (int q = 0; q < 4; ++q) va[q] = (dg && lo4 + q < hi) ? 0.0 : va[q];
            double* dst = As + hi * (second ? KS : PS) + lo4;
            *reinterpret_cast<double2*>(dst) = make_double2(va[0], va[1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(va[2], va[3]);
        }
    };
    auto lstore_b = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u, j = t >> 4, lo4 = (t & 15) * 4;
            double* dst = Bs + j * KS + lo4;                                                         // Bs[j][kk] = Kuf[k*64 + kk, n0 + j]
            *reinterpret_cast<double2*>(dst) = rb[u][0];
            *reinterpret_cast<double2*>(dst + 2) = rb[u][1];
        }
    };
    // k_n . mu from the staged K_uf tile: thread -> (point j = tid & 63, sixteen rows of the tile per wave)
    const bool kmu_duty = part_b && I == 0;          // (the second factor's part of row tile 0 walks over ALL tile columns of K_uf)
    double kacc = 0.0;
    // column sums of squares over the wave's 32 rows -> partial row 2 I + wr
    auto colsums_out = [&](double* __restrict__ part) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            double cs = 0.0;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs = fma(acc.t[ti][tj][r], acc.t[ti][tj][r], cs);
            cs += __shfl_xor(cs, 16);
            cs += __shfl_xor(cs, 32);
            const int64_t n = n0 + wc * 32 + tj * 16 + lane;
            if (lane < 16 && n < N) part[(size_t)(2 * I + wr) * N + n] = cs;
        }
    };
    gload_a(part_b, s_begin - part_b);
    gload_b(s_begin - part_b);
    const int li = lane & 15, lk = lane >> 4;
    const int r0 = wr * 32 + li, r1 = r0 + 16, c0 = wc * 32 + li, c1 = c0 + 16;
#pragma unroll 1
    for (int s = s_begin; s <= s_end; ++s) {          // steps 0 .. I: first factor, tile column k = s; I + 1 .. T: second, k = s - 1
        const int second = s > I ? 1 : 0, k = s - second;
        __syncthreads();                              // the previous tile pair has been consumed
        lstore_a(second, k == I);
        lstore_b();
        __syncthreads();
        if (s < s_end) {                              // the next step's tiles: in flight while the matrix cores run
            const int sn = s + 1, secn = sn > I ? 1 : 0, kn = sn - secn;
            gload_a(secn, kn);
            gload_b(kn);
        }
        if (kmu_duty) {
            const int j = tid & 63;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int kk = 16 * wave + q;
                kacc = fma(Bs[j * KS + kk], mu[k * TB + kk], kacc);
            }
        }
        // operand A of MFMA k-step kk: As[kk][row] (first factor) or As[row][kk] (second); operand B: Bs[column][kk]
        const int ak = second ? 1 : PS, ar = second ? KS : 1;
        const double* ap = As + lk * ak;
        const double* bp = Bs + lk;
#pragma unroll 4
        for (int kk = 0; kk < TB; kk += 4) {
            const double a0 = ap[r0 * ar], a1 = ap[r1 * ar];
            const double b0 = bp[c0 * KS], b1 = bp[c1 * KS];
            acc.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.t[0][0], 0, 0, 0);
            acc.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.t[0][1], 0, 0, 0);
            acc.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.t[1][0], 0, 0, 0);
            acc.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.t[1][1], 0, 0, 0);
            ap += 4 * ak;
            bp += 4;
        }
    }
    colsums_out(part_b ? pb : pa);
    if (kmu_duty) {
        const int64_t n = n0 + (tid & 63);
        if (n < N) kmu[(size_t)wave * N + n] = kacc;
    }
}

// I1_n = sigma2 - sum_r pa[r][n] ;  I2_n = y^2 + v - 2 y (k_n . mu) + sum_r pb[r][n]      (fixed summation order)
__global__ void __launch_bounds__(256) k_w_point_finish(const double* __restrict__ pa, const double* __restrict__ pb,
                                                        const double* __restrict__ kmu, const double* __restrict__ y,
                                                        const double* __restrict__ yv, double* __restrict__ I1,
                                                        double* __restrict__ I2, const Params* __restrict__ P, int T, int64_t N) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < 2 * T; ++r) { a += pa[(size_t)r * N + n]; b += pb[(size_t)r * N + n]; }
    const double km = (kmu[n] + kmu[(size_t)N + n]) + (kmu[(size_t)2 * N + n] + kmu[(size_t)3 * N + n]);
    const double yy = y[n], v = yv ? yv[n] : 0.0;
    I1[n] = P->sigma2 - a;
    I2[n] = yy * yy + v - 2.0 * yy * km + b;
}

// transpose-copy of a square column-major matrix (Uv = L_R^T on the way out)
__global__ void __launch_bounds__(256) k_transpose(const double* __restrict__ A, double* __restrict__ At, int ld) {
    __shared__ double tile[TB * (TB + 1)];
    const int I = blockIdx.x * TB, J = blockIdx.y * TB;
    for (int e = threadIdx.x; e < TB * TB; e += 256) {
        int c = e >> 6, r = e & 63;
        tile[r * (TB + 1) + c] = A[(size_t)(J + c) * ld + I + r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TB * TB; e += 256) {
        int c = e >> 6, r = e & 63;
        At[(size_t)(I + c) * ld + J + r] = tile[c * (TB + 1) + r];
    }
}

// ------------------------------------------------------------------------------------------------
// Analytic gradient of the hyper-parameter objective (helper_functions/derivative_helper.jl:23-39; the reference
// differentiates it with ForwardDiff, :55-63).  With q(v) fixed (mu, R = Sigma_v + mu mu^T), G = R - K_uu^-1 and
// H = K_uu^-1 Psi2 K_uu^-1:
//   f      = w/2 [ s_w sigma2 + tr(G Psi2) - 2 b^T mu ]
//   df     = w/2 [ s_w dsigma2 + sum_mn Z_mn dKuf_mn / Kuf_mn + sum_mm' H_mm' dKuu_mm' ],
//   Z_mn   = 2 (omega_n (G k_n)_m - omega_n y_n mu_m) Kuf_mn,
//   dk/dsigma2 = k / sigma2 ,  dk/dell_d = k (x_d - u_d)^2 / ell_d^3  (jitter does not depend on theta).
// k_theta_grad_uf: one 64 x 64 tile of G K_uf per block on the matrix cores, the contraction with the kernel
//   derivatives in the epilogue, block partial sums [slot 0: sum Z ; slot 1 + d: sum Z ((x_d - u_d) / ell_d)^2].
// k_theta_grad_uu: the same contraction of H with the K_uu derivatives.
// k_theta_grad_finish: fixed-order sum of the partials (bitwise reproducible) and the chain-rule factors.
// ------------------------------------------------------------------------------------------------
constexpr int GRAD_SLOTS = MAXD + 1;

__global__ void __launch_bounds__(256) k_form_G(const double* __restrict__ R, const double* __restrict__ Kinv,
                                                double* __restrict__ G, size_t count) {
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < count) G[e] = R[e] - Kinv[e];
}

__global__ void __launch_bounds__(256) k_theta_grad_uf(const double* __restrict__ G, const double* __restrict__ Kuf,
                                                       const double* __restrict__ X, const double* __restrict__ Xus,
                                                       const double* __restrict__ Yw, const double* __restrict__ omega,
                                                       const double* __restrict__ mu, const Params* __restrict__ P,
                                                       double* __restrict__ partial, int Mp, int T, int D, int64_t N) {
    // grid (nblk, T, KS): with few points (minibatches) the K loop is split over blockIdx.z so that the launch fills the
    // chip; the contraction is linear in G K_uf, so every split contributes its own partial sums (the y mu term rides
    // with split 0)
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    __shared__ double ys[TB], om[TB], mus[TB];
    __shared__ double wsum[4][GRAD_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int I = blockIdx.y;
    const int64_t n0 = (int64_t)blockIdx.x * TB;
    double* As = lds;
    double* Bs = lds + TB * PS;
    Acc4 acc;
    acc_zero(acc);
    const int ks = blockIdx.z, KS = gridDim.z;
    const int kbeg = (int)((int64_t)T * ks / KS), kend = (int)((int64_t)T * (ks + 1) / KS);
    for (int k = kbeg; k < kend; ++k) {
        __syncthreads();
        load_panel_n(As, G, Mp, I * TB, k * TB, TB, tid);          // As[kk][i] = G[I*64 + i, k*64 + kk]  (G symmetric)
        for (int t = tid; t < TB * 16; t += 256) {                 // Bs[j][kk] = Kuf[k*64 + kk, n0 + j]  (stored as it arrives: tile_mma_bk)
            int j = t >> 4, g = t & 15;
            int64_t n = n0 + j;
            double2 v0 = make_double2(0.0, 0.0), v1 = v0;
            if (n < N) {
                const double* src = Kuf + (size_t)n * Mp + k * TB + g * 4;
                v0 = *reinterpret_cast<const double2*>(src);
                v1 = *reinterpret_cast<const double2*>(src + 2);
            }
            double* dst = Bs + j * KMS + g * 4;
            *reinterpret_cast<double2*>(dst) = v0;
            *reinterpret_cast<double2*>(dst + 2) = v1;
        }
        __syncthreads();
        tile_mma_bk(acc, As, Bs, TB, lane, wr, wc);
    }
    __syncthreads();
    // the panels are done: their LDS now holds the scaled coordinates of this block's 64 inducing rows / 64 points
    double* us = lds;
    double* xs = lds + MAXD * TB;
    for (int t = tid; t < D * TB; t += 256) {
        int d = t / TB, r = t % TB;
        us[t] = Xus[(size_t)d * Mp + I * TB + r];
    }
    for (int t = tid; t < D * TB; t += 256) {
        int p = t / D, d = t % D;
        int64_t n = n0 + p;
        xs[d * TB + p] = (n < N) ? X[(size_t)n * D + d] * P->inv_ell[d] : 0.0;
    }
    if (tid < TB) {
        int64_t n = n0 + tid;
        ys[tid] = (n < N) ? Yw[n] : 0.0;
        om[tid] = (n < N) ? (omega ? omega[n] : 1.0) : 0.0;
        mus[tid] = mu[I * TB + tid];
    }
    __syncthreads();
    double z[2][2][4];
    double e0 = 0.0;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int col = acc_col(lane, wc, tj);
            const int64_t n = n0 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = acc_row(lane, wr, ti, r);
                const double kv = (n < N) ? Kuf[(size_t)n * Mp + I * TB + row] : 0.0;
                const double v = 2.0 * (om[col] * acc.t[ti][tj][r] - (ks == 0 ? ys[col] * mus[row] : 0.0)) * kv;
                z[ti][tj][r] = v;
                e0 += v;
            }
        }
    for (int o = 32; o > 0; o >>= 1) e0 += __shfl_xor(e0, o);
    if (lane == 0) wsum[wave][0] = e0;
    for (int d = 0; d < D; ++d) {
        double e = 0.0;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const double xv = xs[d * TB + acc_col(lane, wc, tj)];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double t = xv - us[d * TB + acc_row(lane, wr, ti, r)];
                    e = fma(z[ti][tj][r], t * t, e);
                }
            }
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        if (lane == 0) wsum[wave][1 + d] = e;
    }
    __syncthreads();
    if (tid <= D)
        partial[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * GRAD_SLOTS + tid] =
            (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
}

// one 64 x 64 tile of H o dK_uu per workgroup (grid T x T): the kernel value is computed once per entry, 16 entries per
// thread stay in registers across the D + 1 contractions.  (A first version re-evaluated the kernel in every pass with
// 64 workgroups striding over the matrix: 280 us at M = 600 -- a quarter of a training step.)
__global__ void __launch_bounds__(256) k_theta_grad_uu(const double* __restrict__ H, const double* __restrict__ Xus,
                                                       const Params* __restrict__ P, double* __restrict__ partial,
                                                       int M, int Mp, int D) {
    __shared__ double ui[MAXD * TB];
    __shared__ double uj[MAXD * TB];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int I = blockIdx.x * TB, J = blockIdx.y * TB;
    for (int t = tid; t < D * TB; t += 256) {
        const int d = t / TB, r = t % TB;
        ui[t] = Xus[(size_t)d * Mp + I + r];
        uj[t] = Xus[(size_t)d * Mp + J + r];
    }
    __syncthreads();
    const int r = tid & 63, c0 = (tid >> 6) * 16;
    const double s2 = P->sigma2;
    double hk[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double t = ui[d * TB + r] - uj[d * TB + c0 + e];
            d2 = fma(t, t, d2);
        }
        const bool in = (I + r < M) && (J + c0 + e < M);
        const double h = H[(size_t)(J + c0 + e) * Mp + I + r];
        hk[e] = in ? h * (s2 * exp(-0.5 * d2)) : 0.0;
    }
    double* out = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * GRAD_SLOTS;
    {
        double v = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) v += hk[e];
        v = block_sum(v, red);
        if (tid == 0) out[0] = v;
    }
    for (int d = 0; d < D; ++d) {
        const double a = ui[d * TB + r];
        double v = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double t = a - uj[d * TB + c0 + e];
            v = fma(hk[e], t * t, v);
        }
        v = block_sum(v, red);
        if (tid == 0) out[1 + d] = v;
    }
}

// tot[slot] = fixed-order sum of the data half's block partials: in a data-sharded run this small vector (not the partials) is
// what the ranks sum-all-reduce before k_theta_grad_finish -- the K_uu half and the s_w term come from the REDUCED statistics
// and are the same on every rank already (helper_functions/derivative_helper.jl:29-38 is a sum over points)
__global__ void __launch_bounds__(256) k_theta_grad_fold(const double* __restrict__ part_uf, int n_uf, double* __restrict__ tot, int D) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double acc[GRAD_SLOTS];
#pragma unroll
    for (int sl = 0; sl < GRAD_SLOTS; ++sl) acc[sl] = 0.0;
    for (int b = tid; b < n_uf; b += 256) {
#pragma unroll
        for (int sl = 0; sl < GRAD_SLOTS; ++sl)
            if (sl <= D) acc[sl] += part_uf[(size_t)b * GRAD_SLOTS + sl];
    }
#pragma unroll
    for (int sl = 0; sl < GRAD_SLOTS; ++sl) {
        double v = 0.0;
        if (sl <= D) v = block_sum(acc[sl], red);        // (uniform)
        if (tid == 0) tot[sl] = v;
    }
}

// grad[0] = df/dsigma2, grad[1 ..] = df/dell (n_ell = 1: one shared lengthscale, else one per dimension)
__global__ void __launch_bounds__(256) k_theta_grad_finish(const double* __restrict__ part_uf, int n_uf,
                                                           const double* __restrict__ part_uu, int n_uu,
                                                           const double* __restrict__ stats_scal, const Params* __restrict__ P,
                                                           double* __restrict__ grad, int D, int n_ell,
                                                           const long long* wait_word, long long wait_need, int spin_limit,
                                                           int* sync_status) {
    // wait_word (may be nullptr): the K_uu half of the gradient (part_uu) was formed on the other stream; it is complete when
    // the word reaches wait_need (bounded wait, SYNC_LATE_GRAD_JOIN if it gives up; the partials are then read past this XCD's L2)
    if (wait_word) {
        if (threadIdx.x == 0) spin_until(wait_word, wait_need, spin_limit, sync_status, SYNC_LATE_GRAD_JOIN);
        __syncthreads();
    }
    // thread t sums the partial blocks t, t + 256, ... for ALL slots at once (independent loads; a first version walked the
    // blocks slot by slot with one wave: 31 us of load latency), then one workgroup reduction per slot -- fixed order
    __shared__ double red[4];
    __shared__ double tot[GRAD_SLOTS];
    const int tid = threadIdx.x;
    double acc[GRAD_SLOTS];
#pragma unroll
    for (int sl = 0; sl < GRAD_SLOTS; ++sl) acc[sl] = 0.0;
    for (int b = tid; b < n_uf; b += 256) {
#pragma unroll
        for (int sl = 0; sl < GRAD_SLOTS; ++sl)
            if (sl <= D) acc[sl] += part_uf[(size_t)b * GRAD_SLOTS + sl];
    }
    for (int b = tid; b < n_uu; b += 256) {
#pragma unroll
        for (int sl = 0; sl < GRAD_SLOTS; ++sl)
            if (sl <= D)
                acc[sl] += wait_word ? __hip_atomic_load((const __attribute__((address_space(1))) double*)(part_uu + (size_t)b * GRAD_SLOTS + sl),
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : part_uu[(size_t)b * GRAD_SLOTS + sl];
    }
#pragma unroll
    for (int sl = 0; sl < GRAD_SLOTS; ++sl) {
        if (sl <= D) {                                   // uniform
            const double v = block_sum(acc[sl], red);
            if (tid == 0) tot[sl] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double hw = 0.5 * P->W[0];
        grad[0] = hw * (stats_scal[1] + tot[0] / P->sigma2);
        if (n_ell == 1) {
            double g = 0.0;
            for (int d = 0; d < D; ++d) g += tot[1 + d];
            grad[1] = hw * g * P->inv_ell[0];
        } else {
            for (int d = 0; d < D; ++d) grad[1 + d] = hw * tot[1 + d] * P->inv_ell[d];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Device-paced minibatch training (sgp_train_* in sgp_api.hip): the host only enqueues; the data window's scalars, the
// softplus map theta -> kernel parameters and the optimiser step are tiny kernels between the sweep's own.
// ------------------------------------------------------------------------------------------------
// data scalars of the window y[0 .. n) (what sgp_set_data computes on the host): sum y^2 in S_YY and in the Ryy slot, n in
// S_W and S_N.  One workgroup, fixed summation order.
__global__ void __launch_bounds__(256) k_train_window(const double* __restrict__ y, int64_t n, double* __restrict__ scal,
                                                      int count) {
    __shared__ double red[4];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) a += y[i] * y[i];
    const double syy = block_sum(a, red);
    for (int e = threadIdx.x; e < count; e += 256) {
        double v = 0.0;
        if (e == 0 /* SGP_S_YY */ || e == 8 /* Ryy[0] of d_out = 1 */) v = syy;
        if (e == 1 /* SGP_S_W */ || e == 2 /* SGP_S_N */) v = (double)n;
        scal[e] = v;
    }
}

struct TrainState {
    double theta[MAXD + 1];     // raw (pre-softplus) parameters: sigma2 first, then the lengthscale(s)
    double m[MAXD + 1];         // AdaMax first moment
    double u[MAXD + 1];         // AdaMax infinity-norm accumulator
    double bp[2];               // running powers of beta
    double eta, beta1, beta2, eps;
    double steps, rejected;     // optimiser steps taken / minibatches whose factorisations failed (theta left alone)
    // classification (SGP_LIKELIHOOD_PROBIT): q(w) = Gamma(ga, gb), carried over the minibatches and never reset
    // (experiments/classification_banana.ipynb cell 9: `shape, rate = params(qw)`)
    double ga, gb;
    double kind;                // 0 Gaussian likelihood with a fixed w (regression), 1 Probit with a Gamma q(w)
};

__device__ __forceinline__ double softplus_dev(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }

// Classification minibatch (experiments/classification_banana.ipynb cell 7: `f[i] ~ UniSGP(x[i], v, w, theta); y[i] ~ Probit(f[i])`):
// q(f_i) = the moment-matched product of the UniSGP :out message N(mz_i, 1 / mean(q_w)) (GPnode/UniSGPnode.jl:96-104; mz = k_i' mu_v
// from k_predict on the window) with the Probit likelihood of the label y_i in {0, 1}:
//     s = 2 y - 1,  vz = 1 / w,  g = s mz / sqrt(1 + vz),  r = phi(g) / Phi(g),
//     mean = mz + s vz r / sqrt(1 + vz),   var = vz - vz^2 / (1 + vz) r (g + r).
// r through erfcx for g < 0 (phi / Phi = sqrt(2 / pi) / erfcx(-g / sqrt 2): no cancellation, no underflow in the tail).
// Writes the window's data as the sweep reads it -- mean (twice: y and omega y), variance -- and its data scalars
// (sum mean^2 + var, n, n: what sgp_set_data computes on the host).  One workgroup, fixed summation order.
__global__ void __launch_bounds__(256) k_probit_window(const double* __restrict__ label, const double* __restrict__ mz, int64_t n,
                                                       const TrainState* __restrict__ st, double* __restrict__ ymean,
                                                       double* __restrict__ yw, double* __restrict__ yvar, double* __restrict__ scal,
                                                       int count) {
    __shared__ double red[4];
    const double vz = st->gb / st->ga, q = 1.0 / sqrt(1.0 + vz);
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double sgn = 2.0 * label[i] - 1.0, m = mz[i];
        const double g = sgn * m * q;
        double r;
        if (g < 0.0) r = 0.79788456080286535588 / erfcx(-g * 0.70710678118654752440);
        else r = exp(-0.5 * g * g) * 0.39894228040143267794 / (0.5 * erfc(-g * 0.70710678118654752440));
        const double mf = m + sgn * vz * r * q;
        const double vf = vz - vz * vz / (1.0 + vz) * r * (g + r);
        ymean[i] = mf;
        yw[i] = mf;
        yvar[i] = vf;
        acc += mf * mf + vf;
    }
    const double syy = block_sum(acc, red);
    for (int e = threadIdx.x; e < count; e += 256) {
        double v = 0.0;
        if (e == 0 /* SGP_S_YY */ || e == 8 /* Ryy[0] of d_out = 1 */) v = syy;
        if (e == 1 /* SGP_S_W */ || e == 2 /* SGP_S_N */) v = (double)n;
        scal[e] = v;
    }
}

// One thread: Flux's AdaMax (m = b1 m + (1 - b1) g; u = max(b2 u, |g|); theta -= eta / (1 - b1^t) m / (u + eps)) on the raw
// parameters with the chain rule through softplus (d softplus = sigmoid), then the kernel parameters of the NEXT sweep
// written where k_prep_xu reads them.  `update` = 0 only writes the parameters (first step of a run).
// `update`: 1 = optimiser step (skipped and counted if a factorisation of this minibatch failed or a device-word wait gave up),
// 2 = status only (a minibatch without a learning step: a failure is still counted), 0 = only write the parameters (first step
// of a run).
// Probit runs (st->kind = 1) first update q(w) = Gamma(a + n / 2, b + (sum I1 + sum I2) / 2) from the sweep's scalars and n = the
// statistics' node count (`n_window` points at it; nullptr when update = 0)
// (GPnode/UniSGPnode.jl:219-238 summed over the window); the gradient was formed at the sweep's mean(q_w) and the objective is
// linear in it, so it is rescaled to the NEW mean (`grad_llh_new!(...; w = mean(qw))`, classification_banana.ipynb cell 9), and
// the next sweep's noise precision is written with the kernel parameters.
__global__ void k_train_adamax(TrainState* __restrict__ st, const double* __restrict__ grad, const double* __restrict__ out,
                               Params* __restrict__ src, int D, int n_ell, int update, const int* __restrict__ sync_status,
                               const Params* __restrict__ swept, const double* __restrict__ n_window) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool probit = st->kind == 1.0;
    if (update) {
        const bool ok = out[3] == 0.0 && out[4] == 0.0 && (!sync_status || *sync_status == 0);   // SGP_R_INFO_KUU, SGP_R_INFO_LAMBDA
        double gscale = 1.0;
        if (probit && ok) {
            // (the window's size as the sweep's statistics carry it, SGP_S_N: in a data-sharded run that is the REDUCED scalar --
            // the whole minibatch, like out[0] + out[1] beside it -- not this rank's slice)
            st->ga += 0.5 * *n_window;
            st->gb += 0.5 * (out[0] + out[1]);
            gscale = (st->ga / st->gb) / swept->W[0];
        }
        if (update == 2) {
            if (!ok) st->rejected += 1.0;
            if (probit) { src->W[0] = st->ga / st->gb; src->E_logw = log(st->ga / st->gb); }
            return;
        }
        if (ok) {
            for (int i = 0; i <= n_ell; ++i) {
                const double th = st->theta[i];
                const double g = gscale * grad[i] / (1.0 + exp(-th));
                const double m = st->beta1 * st->m[i] + (1.0 - st->beta1) * g;
                const double u = fmax(st->beta2 * st->u[i], fabs(g));
                st->m[i] = m;
                st->u[i] = u;
                st->theta[i] = th - (st->eta / (1.0 - st->bp[0])) * m / (u + st->eps);
            }
            st->bp[0] *= st->beta1;
            st->bp[1] *= st->beta2;
            st->steps += 1.0;
        } else
            st->rejected += 1.0;
    }
    src->sigma2 = softplus_dev(st->theta[0]);
    for (int d = 0; d < D; ++d) src->inv_ell[d] = 1.0 / softplus_dev(st->theta[1 + (n_ell == 1 ? 0 : d)]);
    if (probit) { src->W[0] = st->ga / st->gb; src->E_logw = log(st->ga / st->gb); }
}

}  // namespace sgp
