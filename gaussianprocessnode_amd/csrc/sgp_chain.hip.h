// sgp_chain.hip.h -- the blocked Cholesky of the sweep's two M x M chains (K_uu and Lambda) as ONE persistent launch.
//
// Why.  With one launch per 64-column step (k_potrf_step) a step costs ~19.5 us at M = 512: tiles in (3.3) + factor the
// diagonal tile (9.1) + solve the tile below it (5.4) + tiles out (0.65) + launch gap (1.5), of which only the 64-pivot
// chain (~3.4 us) is inherently sequential.  A cross-workgroup hand-off costs microseconds too, so replacing kernel
// boundaries by hand-offs one for one gains nothing: the hand-offs have to leave the critical path.
//
// How.  One CRITICAL workgroup (8 waves) owns the diagonal tile (j, j) AND the tile below it (j+1, j) of every step and
// keeps them in LDS from step to step: it factors the 128 x 64 panel (the solve of the lower tile rides in waves of its
// own), forms the next diagonal tile's update L_{j+1,j} L_{j+1,j}^T on the matrix cores while it goes, and rolls over to
// step j+1 without touching memory.  Its eight waves are independent agents that synchronise through progress counters
// in LDS -- there is no s_barrier in the loop, so a lower tile that arrives late delays nothing but its own rows.
// Everything else is dataflow around it, with slack:
//   solver (a, b), a >= b+2 : accumulates A_ab -= L_aj L_bj^T as the columns j < b appear, then solves against L_bb as
//                             ITS 16-column blocks appear (right-looking, so only one 16-pivot solve follows the last
//                             block), publishes L_ab;
//   feeder = solver (a, a-2): additionally forms the two products of its fresh L_{a,a-2} that the critical workgroup
//                             needs one step later -- tile (a, a-1) and the far part of (a, a) -- and ships them;
//   near owner (a, a-1|a)   : accumulates those two tiles' updates from the columns <= a-3 for the feeder;
//   inverter                : L_jj^-1 for the inverse factor.
//
// Hand-offs.  A flag behind drained write-through stores was measured at ~4.5 us per hop here (s_waitcnt vmcnt(0) on sc1
// stores alone: ~2.2 us), several times the guide's price for DATA-TAGGED granules (MI355X_MICROARCH.md, handoff-1to1:
// ~1 us), so the data is its own tag: every handed-off double is written exactly once per launch, by an agent-scope relaxed
// atomic store (global_store_dwordx2 sc1), into memory that holds a SENTINEL -- a signalling NaN, which no arithmetic can
// produce or propagate unchanged -- and consumers poll the data itself with sc1 loads until no sentinel is left.  No flag,
// no drain, no fence.  The buffers are double-buffered by launch parity; every launch puts the sentinels back into the
// OTHER parity's buffers for the launch after it (the kernel boundary in between makes them visible).
// Every spin is bounded: after CH_SPIN_LIMIT polls a waiter raises the abort word, everybody unwinds, and the host
// reports an error instead of a hung GPU.  All arithmetic is in a fixed order: results are bitwise reproducible.
#pragma once
#include "sgp_kernels.hip.h"

namespace sgp {

constexpr int CH_TMAX = 12;                 // tile rows the persistent path supports
constexpr int CH_THREADS = 512;             // the critical workgroup and the feeders use 8 waves, the other helpers the first 4
constexpr int CH_SPIN_LIMIT = 1 << 20;      // polls (each >= ~0.5 us) before a waiter gives up: ~1 s

constexpr int CH_F_ABORT = 0;               // words 0, 1 of `flags`: raised by a waiter of a launch of parity 0 / 1 that gave up
constexpr int CH_F_GATE = 2;                // set by the sweep's streaming SYRK once its workgroups are resident (k_chain_gate)
constexpr int CH_F_COUNT = 3;

constexpr unsigned long long CH_SENT_BITS = 0x7FF4DEADBEEF0001ull;    // signalling NaN (quiet bit 51 clear)

// The arguments live in device memory (one constant struct per chain and launch parity) and the kernel takes a pointer:
// passed by value, the struct was copied to SCRATCH in the prologue and its fields re-read from there in the hot loops --
// every such scratch_load is followed by s_waitcnt vmcnt(0), i.e. by a wait for all the write-through stores in flight.
struct ChainArgs {
    double* A;               // out: L (lower tiles), sentinel-filled on entry: consumers poll it
    double* Far;             // mailbox matrix: far parts of the near-diagonal tiles (a, a-1), (a, a), at their home positions
    double* Ship;            // mailbox matrix: what the feeders ship to the critical workgroup, at the same home positions
    double* rinv_all;        // mailbox (ld): 1 / diag(L), published with the diagonal blocks
    double* A_next;          // the other parity's buffers, which this launch refills with sentinels (may be nullptr)
    double* Far_next;
    double* Ship_next;
    double* rinv_next;
    const double* Ain;       // source 0: the input matrix in a buffer of its own
    int ld, Tn;
    int* info;
    int n_valid;
    double* Winv;            // out (may be nullptr): diagonal tiles of L^-1
    long long* abortw;       // this launch parity's abort word (nonzero: a waiter gave up, everybody unwinds)
    long long* abortw_next;  // the other parity's, which this launch clears for the launch after it (may be nullptr)
    LamForm form;            // source 1 (form.stats != nullptr): Lambda = Lambda0 + W (x) Psi2 evaluated on the fly, index-reversed
    long long* trace;        // diagnostics (may be nullptr): [step][16] 100 MHz ticks of the critical workgroup, see sgp_get_chain_trace
    const double* Xus;       // source 2 (Xus != nullptr): K_uu + jitter I from the scaled inducing inputs (D x ld SoA), pad = identity
    const Params* P;
    int M, D;
};

// (the casts to the global address space matter: pointers taken out of the by-value argument struct are generic to the
// compiler, generic accesses become FLAT instructions, and those count on lgkmcnt as well as vmcnt -- every wait for an LDS
// read then also waited ~1.5 us for the global loads in flight)
#define CH_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ double ldc(const double* p) {
    return __hip_atomic_load((const CH_GLOBAL double*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc(double* p, double v) {
    __hip_atomic_store((CH_GLOBAL double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Stores of hand-off data are 8-byte agent-scope atomics the compiler can see, laid out so that 16 lanes cover one 128-byte
// column segment.  (Two things were tried and dropped.  Scattered 8-byte sc1 stores -- adjacent lanes in different columns --
// cost their wave ~0.5 us per instruction.  16-byte sc1 stores through inline asm were faster per byte but corrupted data:
// the compiler reuses the data registers right after an asm statement it cannot know to be a wide store, and under
// write-through back-pressure the store reads them late -- lanes 12..15 of every 16, first element, were seen to carry the
// next block's values.)
// 16-byte write-through store through the raw-buffer intrinsic (buffer_store_dwordx4 ... sc1): an instruction the compiler's
// hazard and wait-count passes know, unlike the inline-asm form.  `base` must be wave-uniform, `off` is in doubles.
#ifndef CH_STORE16
#define CH_STORE16 1
#endif
typedef int ch_v4i __attribute__((ext_vector_type(4)));
typedef double ch_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stc16(double* base, size_t off, double a, double b) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const ch_d2 v = {a, b};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_v4i, v), r, (int)(off * 8), 0, 16 /* sc1 */);
}
__device__ __forceinline__ long long fl_load(const long long* p) {
    return __hip_atomic_load((const CH_GLOBAL long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fl_store(long long* p, long long v) {
    __hip_atomic_store((CH_GLOBAL long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ bool is_sent(double v) { return (unsigned long long)__double_as_longlong(v) == CH_SENT_BITS; }
__device__ __forceinline__ double sentinel() { return __longlong_as_double((long long)CH_SENT_BITS); }

// ---- bounded spinning -----------------------------------------------------------------------------------------------
// one more round of a wave-level poll loop; false = give up (somebody raised the abort word, or this waiter does now)
__device__ __forceinline__ bool spin_more(int& it, const ChainArgs& g) {
    __builtin_amdgcn_s_sleep(1);
    if ((++it & 31) == 0) {
        if (fl_load(g.abortw) != 0) return false;
        if (it > CH_SPIN_LIMIT) { fl_store(g.abortw, 1); return false; }
    }
    return true;
}
// Wave-level wait on a few PROBE words (one address per lane, nullptr = none): waiters poll these -- the word each producing
// wave stores last -- instead of the payload, whose full load (and verification: stores of one wave need not land in order)
// follows only then.  Polling whole payloads from ~200 waiting waves was ~1 TB/s of sc1 traffic and slowed every hand-off.
__device__ __forceinline__ bool wave_probe(const double* addr, const ChainArgs& g) {
    int it = 0;
    for (;;) {
        const bool bad = addr ? is_sent(ldc(addr)) : false;
        if (!__any(bad)) { compiler_fence(); return true; }
        if (!spin_more(it, g)) return false;
    }
}
// the word that the wave owning row block w of a 64 x 64 tile stores last (see the publishers): (16 w + 15, 63)
__device__ __forceinline__ const double* tile_probe_addr(const double* M, size_t ld, int row0, int col0, int w) {
    return M + (size_t)(col0 + 63) * ld + row0 + 16 * w + 15;
}
// LDS progress counters of a workgroup whose waves run without barriers (wave-level)
// Relaxed atomics plus an explicit wait for the wave's LDS operations: acquire / release at workgroup scope make the compiler
// emit s_waitcnt vmcnt(0) as well, i.e. every counter update would wait ~2 us for the wave's write-through stores in flight
// (seen in the ISA and in the trace).  The data these counters guard is in LDS, whose operations a wave issues in order.
__device__ __forceinline__ int lds_get(int* p) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_set(int* p, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_inc(int* p) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool lwait(int* p, int need, int* abortl) {
    int it = 0;
    while (lds_get(p) < need) {
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 63) == 0) {
            if (lds_get(abortl) != 0) return false;
            if (it > (CH_SPIN_LIMIT << 2)) { lds_set(abortl, 1); return false; }
        }
    }
    return true;
}
// a waiter gave up or saw somebody else give up: raise the abort word for everybody, mark the chain's status word
__device__ __forceinline__ void chain_abort(const ChainArgs& g, int* abortl) {
    if ((threadIdx.x & 63) == 0) {
        fl_store(g.abortw, 1);
        atomicExch(g.info, -1);
        if (abortl) __hip_atomic_store(abortl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ---- mailbox tiles: t = thread index within the 256 threads that move the tile ------------------------------------------
// put the sentinels back into tile (ti, tj) of a mailbox matrix
__device__ __forceinline__ void tile_reset(double* M, size_t ld, int ti, int tj, int t) {
    if (!M) return;
    const double sv = sentinel();
#if CH_STORE16
    const int p = t & 31, c0 = t >> 5;
#pragma unroll
    for (int u = 0; u < 8; ++u) stc16(M, (size_t)(tj * TB + c0 + 8 * u) * ld + ti * TB + 2 * p, sv, sv);
#else
    const int r = t & 63, c0 = t >> 6;
#pragma unroll
    for (int u = 0; u < 16; ++u) stc(M + (size_t)(tj * TB + c0 + 4 * u) * ld + ti * TB + r, sv);
#endif
}
// registers <- mailbox tile; returns true while a sentinel is still in this thread's share
__device__ __forceinline__ bool tile_poll(double (&v)[16], const double* M, size_t ld, int row0, int col0, int t) {
    const int r = t & 63, c0 = t >> 6;
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        v[u] = ldc(M + (size_t)(col0 + c0 + 4 * u) * ld + row0 + r);
        bad |= is_sent(v[u]);
    }
    return bad;
}
__device__ __forceinline__ void tile_s2g_c(const double* S, double* A, size_t ld, int row0, int col0, int t) {
    // wave w = t >> 6 publishes row block w (so that its last word is the tile's probe word (16 w + 15, 63))
    const int w = t >> 6, lane = t & 63;
#if CH_STORE16
    const int c = lane >> 3, p = lane & 7, r = 16 * w + 2 * p;
#pragma unroll
    for (int u = 0; u < 8; ++u) stc16(A, (size_t)(col0 + c + 8 * u) * ld + row0 + r, S[r * LT + c + 8 * u], S[(r + 1) * LT + c + 8 * u]);
#else
    const int r = 16 * w + (lane & 15), q = lane >> 4;
#pragma unroll
    for (int u = 0; u < 16; ++u) stc(A + (size_t)(col0 + q + 4 * u) * ld + row0 + r, S[r * LT + q + 4 * u]);
#endif
}
// Workgroup-level wait for whole mailbox tiles (threads < 256 poll, every thread of the workgroup takes part in the votes).
// `n` tiles (1 or 2) land in registers; false = abort.
__device__ __forceinline__ bool wg_poll_tiles(double (&v0)[16], const double* M0, int r0, int c0, double (&v1)[16], const double* M1,
                                              int r1, int c1, int n, size_t ld, const ChainArgs& g) {
    const int tid = threadIdx.x;
    int it = 0;
    // probe phase: threads 0..3 (and 4..7) watch the last word of each row block of the tile(s)
    const double* pa = nullptr;
    if (tid < 4) pa = tile_probe_addr(M0, ld, r0, c0, tid);
    else if (tid < 8 && n > 1) pa = tile_probe_addr(M1, ld, r1, c1, tid - 4);
    for (;;) {
        const bool bad = pa ? is_sent(ldc(pa)) : false;
        if (!__syncthreads_or(bad ? 1 : 0)) break;
        __builtin_amdgcn_s_sleep(4);
        if ((++it & 15) == 0) {
            int ab = 0;
            if (tid == 0) {
                ab = (fl_load(g.abortw) != 0) || it > CH_SPIN_LIMIT;
                if (ab) chain_abort(g, nullptr);
            }
            if (__syncthreads_or(ab)) return false;
        }
    }
    for (;;) {
        bool bad = false;
        if (tid < 256) {
            bad = tile_poll(v0, M0, ld, r0, c0, tid);
            if (n > 1) bad |= tile_poll(v1, M1, ld, r1, c1, tid);
        }
        if (!__syncthreads_or(bad ? 1 : 0)) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((++it & 15) == 0) {
            int ab = 0;
            if (tid == 0) {
                ab = (fl_load(g.abortw) != 0) || it > CH_SPIN_LIMIT;
                if (ab) chain_abort(g, nullptr);
            }
            if (__syncthreads_or(ab)) return false;
        }
    }
}

// raw tile (row0.., col0..) of the chain's input matrix into the LDS tile S: formed on the fly (Lambda chain) or read
template <bool DENSE, bool MULTI>
__device__ __forceinline__ void tile_form_s_impl(double* S, const LamForm& f, int Qp, int row0, int col0, int t) {
    const double prior_iso = f.P->prior_iso, w00 = f.P->W[0];
    const int r = t & 63, c0 = t >> 6;
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
        v[u] = lambda_entry<DENSE, MULTI>(f, Qp - 1 - (row0 + r), Qp - 1 - (col0 + c0 + 4 * u), Qp, prior_iso, w00);
#pragma unroll
    for (int u = 0; u < 16; ++u) S[r * LT + c0 + 4 * u] = v[u];
}
// K_uu tile from the scaled inducing inputs (what k_gram_uu computes): thread t keeps its row's coordinates in registers
__device__ __forceinline__ void tile_gram_s(double* S, const ChainArgs& g, int ti, int tj, int t) {
    const int r = t & 63, c0 = t >> 6;
    const int gi = ti * TB + r;
    const double s2 = g.P->sigma2, jit = g.P->jitter;
    double d2[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) d2[u] = 0.0;
    for (int d = 0; d < g.D; ++d) {
        const double ui = g.Xus[(size_t)d * g.ld + gi];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double tt = ui - g.Xus[(size_t)d * g.ld + tj * TB + c0 + 4 * u];
            d2[u] = fma(tt, tt, d2[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int gj = tj * TB + c0 + 4 * u;
        double v;
        if (gi < g.M && gj < g.M) v = s2 * exp(-0.5 * d2[u]) + (gi == gj ? jit : 0.0);
        else v = (gi == gj) ? 1.0 : 0.0;
        S[r * LT + c0 + 4 * u] = v;
    }
}
__device__ __forceinline__ void tile_raw_s(double* S, const ChainArgs& g, int ti, int tj, int t) {
    if (g.Xus) { tile_gram_s(S, g, ti, tj, t); return; }
    if (g.form.stats) {
        const bool dense = g.form.prior_form == 1, multi = g.form.d_out > 1;
        if (dense) { if (multi) tile_form_s_impl<true, true>(S, g.form, g.ld, ti * TB, tj * TB, t); else tile_form_s_impl<true, false>(S, g.form, g.ld, ti * TB, tj * TB, t); }
        else       { if (multi) tile_form_s_impl<false, true>(S, g.form, g.ld, ti * TB, tj * TB, t); else tile_form_s_impl<false, false>(S, g.form, g.ld, ti * TB, tj * TB, t); }
    } else {
        const int r = t & 63, c0 = t >> 6;
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = g.Ain[(size_t)(tj * TB + c0 + 4 * u) * g.ld + ti * TB + r];
#pragma unroll
        for (int u = 0; u < 16; ++u) S[r * LT + c0 + 4 * u] = v[u];
    }
}

// one entry (gi, gj) of the chain's input matrix
__device__ __forceinline__ double raw_entry(const ChainArgs& g, int gi, int gj) {
    if (g.Xus) {
        double d2 = 0.0;
        for (int d = 0; d < g.D; ++d) {
            const double tt = g.Xus[(size_t)d * g.ld + gi] - g.Xus[(size_t)d * g.ld + gj];
            d2 = fma(tt, tt, d2);
        }
        if (gi < g.M && gj < g.M) return g.P->sigma2 * exp(-0.5 * d2) + (gi == gj ? g.P->jitter : 0.0);
        return (gi == gj) ? 1.0 : 0.0;
    }
    if (g.form.stats) {
        const double prior_iso = g.form.P->prior_iso, w00 = g.form.P->W[0];
        const int ri = g.ld - 1 - gi, rj = g.ld - 1 - gj;
        const bool dense = g.form.prior_form == 1, multi = g.form.d_out > 1;
        if (dense) return multi ? lambda_entry<true, true>(g.form, ri, rj, g.ld, prior_iso, w00) : lambda_entry<true, false>(g.form, ri, rj, g.ld, prior_iso, w00);
        return multi ? lambda_entry<false, true>(g.form, ri, rj, g.ld, prior_iso, w00) : lambda_entry<false, false>(g.form, ri, rj, g.ld, prior_iso, w00);
    }
    return g.Ain[(size_t)gj * g.ld + gi];
}

// ---- 16 x 16 building blocks shared by the critical workgroup's waves and the solvers' waves ------------------------------
// 16 pivots of the diagonal block at columns 16 cb of the wave's own 16 rows (Rw: row pr at Rw + pr * LT).  The pivot loop
// is potf2_tile's (DPP row broadcast over the full symmetric block, one ds_bpermute per pivot, deferred scaling, the
// update formed as (x y) / d so that the block stays bitwise symmetric).  Writes L into Rw, the prepared block into Dp,
// 1 / diag into rinv, and returns the wave's L entries: lane (pr, pq) holds L[pr][4 i + pq] in lout[i] (zero above the diagonal).
__device__ __forceinline__ void pivot_block16(double* Rw, int cb, double* Dp, double* rinv, int* info, int col_base, int n_valid,
                                              double (&lout)[4], double& rinv_lane) {
    const int lane = threadIdx.x & 63;
    const int pr = lane & 15, pq = lane >> 4;
    double a[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * i + pq;
        a[i] = (c <= pr) ? Rw[pr * LT + 16 * cb + c] : Rw[c * LT + 16 * cb + pr];
        lo[i] = 0.0;
    }
    double dsave = 1.0;
    static_for<16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int kq = k & 3, ki = k >> 2;
        double y[4];
#pragma unroll
        for (int i = ki; i < 4; ++i) y[i] = row_bcast<k>(a[i]);
        const int src = 4 * (16 * kq + pr);
        const double x = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(a[ki])),
                                          __builtin_amdgcn_ds_bpermute(src, __double2loint(a[ki])));
        const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[ki]), 16 * kq + k),
                                          __builtin_amdgcn_readlane(__double2loint(a[ki]), 16 * kq + k));
        dsave = (lane == k) ? d : dsave;
        const double r = __builtin_amdgcn_rcp(d);
        const double e = fma(-d, r, 1.0);
        const double w = fma(e, e, e);
        const double dinv = fma(r, w, r);
        lo[ki] = (pq == kq && pr >= k) ? x : lo[ki];
#pragma unroll
        for (int i = ki; i < 4; ++i) a[i] = fma(-dinv, x * y[i], a[i]);
    });
    const unsigned long long failed = __ballot(lane < 16 && !(dsave > 0.0));
    if (failed != 0ull && lane == 0) {
        const int bad = __builtin_ctzll(failed);
        if (col_base + 16 * cb + bad < n_valid) atomicCAS(info, 0, col_base + 16 * cb + bad + 1);
    }
    const double ri = rsqrt_nr(dsave);
    rinv_lane = ri;                                       // lanes 0..15: 1 / L_kk of pivot k = lane
    if (lane < 16) rinv[16 * cb + lane] = ri;
    __builtin_amdgcn_wave_barrier();
    const double rrow = rinv[16 * cb + pr];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * i + pq;
        const double l = lo[i] * rinv[16 * cb + c];
        Rw[pr * LT + 16 * cb + c] = l;
        Dp[cb * DPB + pr * DPS + c] = (c < pr) ? l * rrow : 0.0;
        lout[i] = l;
    }
}

// solve the wave's 16 x 16 block at columns 16 cb against the prepared diagonal block; x (solve coordinates: lane = 4 rr + q
// holds row rr, columns 4 i + q) stays in registers for the caller to publish
__device__ __forceinline__ void solve_block16(double* Rw, int cb, const double* Dp, const double* rinv, double (&x)[4]) {
    const int lane = threadIdx.x & 63, rr = lane >> 2, q = lane & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = Rw[rr * LT + 16 * cb + 4 * i + q];
    solve16(x, Dp + cb * DPB, rinv + 16 * cb, q, [](auto) {});
#pragma unroll
    for (int i = 0; i < 4; ++i) Rw[rr * LT + 16 * cb + 4 * i + q] = x[i];
}

// right-looking rank-16 update of the wave's block at columns 16 c:  Rw[:, 16 c ..] -= Rw[:, 16 cb ..] * Rc[:, 16 cb ..]^T
// (Rc = the 16 rows of L that belong to block row c)
__device__ __forceinline__ void update_block16(double* Rw, const double* Rc, int c, int cb) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double* ap = Rw + li * LT + 16 * cb + lk;
    const double* bp = Rc + li * LT + 16 * cb + lk;
    const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
    const double b0 = bp[0], b1 = bp[4], b2 = bp[8], b3 = bp[12];
    // four independent accumulators: a dependent v_mfma_f64 chain costs ~200 cycles per link
    const d4 Z = (d4){0.0, 0.0, 0.0, 0.0};
    const d4 p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, Z, 0, 0, 0);
    const d4 p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, Z, 0, 0, 0);
    const d4 p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, Z, 0, 0, 0);
    const d4 p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, Z, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Rw[(lk + 4 * r) * LT + 16 * c + li] -= (p0[r] + p1[r]) + (p2[r] + p3[r]);
}

// publish a 16 x 16 block that sits in LDS (B[r * stride + c]) to global rows grow0.., columns gcol0.. : lane = 16 q + r
// stores row r of columns q, q + 4, q + 8, q + 12 -- sixteen lanes per 128-byte column segment, four instructions
__device__ __forceinline__ void publish16_lds(const double* B, int stride, double* A, size_t ld, int grow0, int gcol0) {
#if CH_STORE16
    // lane = 8 c' + p stores rows 2 p, 2 p + 1 of column c' (first instruction) and c' + 8 (second): eight lanes per 128-byte segment
    const int lane = threadIdx.x & 63, c = lane >> 3, p = lane & 7;
    const double v00 = B[(2 * p) * stride + c], v01 = B[(2 * p + 1) * stride + c];
    const double v10 = B[(2 * p) * stride + c + 8], v11 = B[(2 * p + 1) * stride + c + 8];
    stc16(A, (size_t)(gcol0 + c) * ld + grow0 + 2 * p, v00, v01);
    stc16(A, (size_t)(gcol0 + c + 8) * ld + grow0 + 2 * p, v10, v11);
#else
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = B[r * stride + 4 * i + q];
#pragma unroll
    for (int i = 0; i < 4; ++i) stc(A + (size_t)(gcol0 + 4 * i + q) * ld + grow0 + r, v[i]);
#endif
}
__device__ __forceinline__ void publish_zero16(double* A, size_t ld, int grow0, int gcol0) {
#if CH_STORE16
    const int lane = threadIdx.x & 63, c = lane >> 3, p = lane & 7;
    stc16(A, (size_t)(gcol0 + c) * ld + grow0 + 2 * p, 0.0, 0.0);
    stc16(A, (size_t)(gcol0 + c + 8) * ld + grow0 + 2 * p, 0.0, 0.0);
#else
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) stc(A + (size_t)(gcol0 + 4 * i + q) * ld + grow0 + r, 0.0);
#endif
}

// ---- the critical workgroup ---------------------------------------------------------------------------------------------
// LDS of the launch (shared with the helper roles): `lds` (2 x 64 x PS doubles) and `tiles` (2 x 64 x LT).
// Buffers: B0 = tiles, B1 = tiles + 64 LT (always the lower tile X), B2 = lds.  S alternates between B0 and B2; the other
// one of the pair holds the next diagonal tile while its update is being collected.
// Duties of the S waves once their pivot run is over (the lower tile usually arrives late, so its waves' phases are what the
// next step waits for: they carry nothing but solves and their right-looking updates):
//   * D blocks -- the 16 x 16 blocks (R, C), R >= C, of X X^T, the next diagonal tile's update: wave 0: 1, waves 1..3: 3 each;
//   * publishing the lower tile's column blocks 0..2 (a 1 KB write-through store costs its wave ~0.4 us, 0.8 us per block):
//     wave 0 for X waves 4 and 7, wave 1 for 5, wave 2 for 6 (block 3 the X waves publish themselves: they are done then);
//   * wave 0: receiving the far part of the next diagonal tile.
constexpr int D_MAXB = 3;
__device__ __forceinline__ int d_block_count(int wave) { return wave >= 4 ? 0 : (wave == 0 ? 1 : 3); }
__device__ __forceinline__ void d_block(int wave, int e, int& R, int& C) {
    //  wave 0: (0,0)   wave 1: (1,0) (1,1) (2,0)   wave 2: (2,1) (2,2) (3,0)   wave 3: (3,1) (3,2) (3,3)
    switch (wave * 4 + e) {
        case 0: R = 0; C = 0; break;
        case 4: R = 1; C = 0; break;   case 5: R = 1; C = 1; break;   case 6: R = 2; C = 0; break;
        case 8: R = 2; C = 1; break;   case 9: R = 2; C = 2; break;   case 10: R = 3; C = 0; break;
        case 12: R = 3; C = 1; break;  case 13: R = 3; C = 2; break;  case 14: R = 3; C = 3; break;
        default: R = 0; C = 0; break;
    }
}

__device__ void chain_critical(const ChainArgs& g, double* lds, double* tiles, double* dprep, double* rinv, int* sy) {
    // sy: [0] runDone, [1..8] rowDone[wave], [9] dDone, [10] dtReady, [11] abort
    int* runDone = sy + 0;
    int* rowDone = sy + 1;
    int* dDone = sy + 9;
    int* dtReady = sy + 10;
    int* abortl = sy + 11;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int Tn = g.Tn;
    const size_t ld = g.ld;
    double* B0 = tiles;
    double* B1 = tiles + TB * LT;
    double* B2 = lds;
    if (g.form.stats) stamp_enter(g.form.stamps);
    // step 0: (0,0) -> B0, (1,0) -> B1, (1,1) -> B2
    if (tid < 256) {
        tile_raw_s(B0, g, 0, 0, tid);
        if (Tn > 1) tile_raw_s(B2, g, 1, 1, tid);
    } else if (Tn > 1) {
        tile_raw_s(B1, g, 1, 0, tid - 256);
    }
    if (tid < 16) sy[tid] = (tid == 10) ? 1 : 0;           // (Dt of step 0 is in place)
    __syncthreads();
    const bool swave = wave < 4;
    const int w = swave ? wave : wave - 4;                 // row block within the tile
    const int nd = d_block_count(wave);
    if (g.trace && tid == 0) {                             // shader clock vs the 100 MHz constant clock: the clock the chip holds
        ((CH_GLOBAL long long*)g.trace)[11 * 32 + 28] = (long long)__builtin_amdgcn_s_memtime();
        ((CH_GLOBAL long long*)g.trace)[11 * 32 + 29] = realtime_ticks();
    }
#define CH_FAIL() do { chain_abort(g, abortl); return; } while (0)
#define CH_TRACE(slot) do { if (g.trace && lane == 0) ((CH_GLOBAL long long*)g.trace)[j * 32 + (slot)] = realtime_ticks(); } while (0)
    for (int j = 0; j < Tn; ++j) {
        double* S = (j & 1) ? B2 : B0;
        double* Dt = (j & 1) ? B0 : B2;
        double* X = B1;
        const bool has_x = j + 1 < Tn;
        const int j0 = j * TB;
        d4 dacc[D_MAXB];
#pragma unroll
        for (int e = 0; e < D_MAXB; ++e) dacc[e] = (d4){0.0, 0.0, 0.0, 0.0};
        // slice s of the wave's D blocks: needs column block s of both row blocks of X
        auto d_slice_block = [&](int s, int e, d4& acc) -> bool {
            int R, C;
            d_block(wave, e, R, C);
            if (!lwait(rowDone + 4 + R, 4 * j + s + 1, abortl)) return false;
            if (!lwait(rowDone + 4 + C, 4 * j + s + 1, abortl)) return false;
            const double* ap = X + (16 * R + li) * LT + 16 * s + lk;
            const double* bp = X + (16 * C + li) * LT + 16 * s + lk;
            const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
            const double b0 = bp[0], b1 = bp[4], b2 = bp[8], b3 = bp[12];
            const d4 Z = (d4){0.0, 0.0, 0.0, 0.0};
            const d4 p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, Z, 0, 0, 0);
            const d4 p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, Z, 0, 0, 0);
            const d4 p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, Z, 0, 0, 0);
            const d4 p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, Z, 0, 0, 0);
            acc += (p0 + p1) + (p2 + p3);
            return true;
        };
        auto d_slice = [&](int s) -> bool {                 // (explicit per-block calls: a loop with an early exit put dacc into scratch)
            if (nd > 0 && !d_slice_block(s, 0, dacc[0])) return false;
            if (nd > 1 && !d_slice_block(s, 1, dacc[1])) return false;
            if (nd > 2 && !d_slice_block(s, 2, dacc[2])) return false;
            return true;
        };
        auto d_tail_block = [&](int e, const d4& acc) {
            int R, C;
            d_block(wave, e, R, C);
#pragma unroll
            for (int r = 0; r < 4; ++r) Dt[(16 * R + lk + 4 * r) * LT + 16 * C + li] -= acc[r];
        };
        auto d_tail = [&]() -> bool {
            if (nd > 0) {
                if (!lwait(dtReady, j + 1, abortl)) return false;
                d_tail_block(0, dacc[0]);
                if (nd > 1) d_tail_block(1, dacc[1]);
                if (nd > 2) d_tail_block(2, dacc[2]);
            }
            lds_inc(dDone);
            return true;
        };
        if (swave) {
            double* Rw = S + 16 * w * LT;
            if (j > 0 && !lwait(dDone, 8 * j, abortl)) CH_FAIL();          // S of this step is final
            if (wave == 0) CH_TRACE(0);
            for (int cb = 0; cb < w; ++cb) {
                if (!lwait(runDone, 4 * j + cb + 1, abortl)) CH_FAIL();
                if (cb + 1 == w) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);   // (the wave that pivots next is the critical path)
                double x[4];
                solve_block16(Rw, cb, dprep, rinv, x);
                lds_set(rowDone + wave, 4 * j + cb + 1);
                update_block16(Rw, Rw, w, cb);                              // own diagonal block first: the next pivot run waits for it
                if (cb + 1 < w) publish16_lds(Rw + 16 * cb, LT, g.A, ld, j0 + 16 * w, j0 + 16 * cb);
                for (int c = cb + 1; c < w; ++c) {
                    if (!lwait(rowDone + c, 4 * j + cb + 1, abortl)) CH_FAIL();
                    update_block16(Rw, S + 16 * c * LT, c, cb);
                }
            }
            double lreg[4], rl;
            __builtin_amdgcn_s_setprio(3);
            pivot_block16(Rw, w, dprep, rinv, g.info, j0, g.n_valid, lreg, rl);
            lds_set(runDone, 4 * j + w + 1);
            __builtin_amdgcn_s_setprio(0);
            if (wave == 3) CH_TRACE(1);
            if (w > 0) publish16_lds(Rw + 16 * (w - 1), LT, g.A, ld, j0 + 16 * w, j0 + 16 * (w - 1));   // (solved just before the run)
            publish16_lds(Rw + 16 * w, LT, g.A, ld, j0 + 16 * w, j0 + 16 * w);
            if (lane < 16) stc(g.rinv_all + j0 + 16 * w + lane, rl);
            for (int c = w + 1; c < 4; ++c) publish_zero16(g.A, ld, j0 + 16 * w, j0 + 16 * c);
            if (has_x) {
                // wave 0 also receives the far part of the next diagonal tile (j+1, j+1) -- its lower blocks -- shipped by the
                // feeder of row j+1 one step ahead of its use; its buffer is last step's S (every wave has left that step:
                // dDone).  The shipment is probed between the D slices and waited for only at the end.
                bool need_dt = (wave == 0 && j > 0);
                const double* src = g.Ship + (size_t)((j + 1) * TB) * ld + (j + 1) * TB;     // element (r, c) at src[c * ld + r]
                auto dt_probe = [&]() -> bool {             // one word of each of the ten lower blocks
                    int R = 0;
                    while ((R + 1) * (R + 2) / 2 <= lane) ++R;
                    const int C = lane - R * (R + 1) / 2;
                    const double v = (lane < 10) ? ldc(src + (size_t)(16 * C) * ld + 16 * R) : 0.0;
                    return !__any(is_sent(v));
                };
                auto dt_receive = [&]() -> bool {           // lanes along the rows; column c needs rows >= 16 (c / 16)
                    int it = 0;
#pragma unroll 1
                    for (int ch = 0; ch < 4; ++ch) {
                        for (;;) {
                            double v[16];
                            bool bad = false;
#pragma unroll
                            for (int u = 0; u < 16; ++u) {                 // (clamped rows instead of a select: see issue_block)
                                const int c = 16 * ch + u;
                                v[u] = ldc(src + (size_t)c * ld + max(lane, 16 * (c >> 4)));
                                bad |= is_sent(v[u]);
                            }
                            if (!__any(bad)) {
#pragma unroll
                                for (int u = 0; u < 16; ++u) Dt[lane * LT + 16 * ch + u] = v[u];
                                break;
                            }
                            if (!spin_more(it, g)) return false;
                        }
                    }
                    lds_set(dtReady, j + 1);
                    need_dt = false;
                    CH_TRACE(8);
                    return true;
                };
                for (int s = 0; s < 4; ++s) {
                    if (need_dt && dt_probe() && !dt_receive()) CH_FAIL();
                    if (s < 3 && wave < 3) {               // the lower tile's column block s, for the X wave(s) this wave publishes for
                        const int xw = 4 + wave;
                        if (!lwait(rowDone + xw, 4 * j + s + 1, abortl)) CH_FAIL();
                        publish16_lds(X + 16 * wave * LT + 16 * s, LT, g.A, ld, (j + 1) * TB + 16 * wave, j0 + 16 * s);
                        if (wave == 0) {
                            if (!lwait(rowDone + 7, 4 * j + s + 1, abortl)) CH_FAIL();
                            publish16_lds(X + 48 * LT + 16 * s, LT, g.A, ld, (j + 1) * TB + 48, j0 + 16 * s);
                        }
                    }
                    if (!d_slice(s)) CH_FAIL();
                }
                if (need_dt && !dt_receive()) CH_FAIL();
                if (!d_tail()) CH_FAIL();
                if (wave == 0) CH_TRACE(6);
                if (wave == 1) CH_TRACE(7);
            }
        } else if (has_x) {
            double* Rw = X + 16 * w * LT;
            if (j > 0) {
                // the wave's 16 rows of tile (j+1, j), fully updated, from the feeder of row j+1 (everybody has finished
                // reading last step's X: dDone)
                if (!lwait(dDone, 8 * j, abortl)) CH_FAIL();
                if (wave == 4) CH_TRACE(5);
                const double* src = g.Ship + (size_t)j0 * ld + (j + 1) * TB + 16 * w;
                if (!wave_probe(lane == 0 ? tile_probe_addr(g.Ship, ld, (j + 1) * TB, j0, w) : nullptr, g)) CH_FAIL();
                int it = 0;
                for (;;) {
                    double v[16];
                    bool bad = false;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        v[u] = ldc(src + (size_t)(lk + 4 * u) * ld + li);
                        bad |= is_sent(v[u]);
                    }
                    if (!__any(bad)) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) Rw[li * LT + lk + 4 * u] = v[u];
                        break;
                    }
                    if (!spin_more(it, g)) CH_FAIL();
                }
                if (wave == 4) CH_TRACE(2);
            }
            __builtin_amdgcn_s_setprio(2);
            for (int cb = 0; cb < 4; ++cb) {
                if (!lwait(runDone, 4 * j + cb + 1, abortl)) CH_FAIL();
                double x[4];
                solve_block16(Rw, cb, dprep, rinv, x);
                lds_set(rowDone + wave, 4 * j + cb + 1);
                if (cb == 3) publish16_lds(Rw + 48, LT, g.A, ld, (j + 1) * TB + 16 * w, j0 + 48);      // (blocks 0..2: the S waves)
                // right-looking updates of the blocks to the right, their MFMAs issued together
                {
                    const double* ap = Rw + li * LT + 16 * cb + lk;
                    const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
                    const d4 Z = (d4){0.0, 0.0, 0.0, 0.0};
                    d4 u[3];
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const int c = cb + 1 + e;
                        u[e] = Z;
                        if (c < 4) {
                            if (!lwait(rowDone + c, 4 * j + cb + 1, abortl)) CH_FAIL();
                            const double* bp = S + (16 * c + li) * LT + 16 * cb + lk;
                            const d4 p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bp[0], Z, 0, 0, 0);
                            const d4 p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bp[4], Z, 0, 0, 0);
                            const d4 p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bp[8], Z, 0, 0, 0);
                            const d4 p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, bp[12], Z, 0, 0, 0);
                            u[e] = (p0 + p1) + (p2 + p3);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const int c = cb + 1 + e;
                        if (c < 4) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) Rw[(lk + 4 * r) * LT + 16 * c + li] -= u[e][r];
                        }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (wave == 4) CH_TRACE(3);
            if (!d_tail()) CH_FAIL();
            if (wave == 4) CH_TRACE(4);
        }
    }
    if (g.trace && tid == 0) {
        ((CH_GLOBAL long long*)g.trace)[11 * 32 + 30] = (long long)__builtin_amdgcn_s_memtime();
        ((CH_GLOBAL long long*)g.trace)[11 * 32 + 31] = realtime_ticks();
    }
#undef CH_TRACE
#undef CH_FAIL
}

// ---- helper roles ---------------------------------------------------------------------------------------------------------
// acc += L_aj L_bj^T for j in [0, jend): the two tiles are polled into registers, staged as MFMA panels in `lds` (threads
// >= 256 -- the feeder's product waves -- only take part in the votes and barriers)
__device__ __forceinline__ bool accumulate_updates(Acc4& acc, int a, int b, int jend, const ChainArgs& g, double* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = (wave >> 1) & 1, wc = wave & 1;
    const bool worker = tid < 256;
    double* P0 = lds;
    double* P1 = lds + TB * PS;
    for (int j = 0; j < jend; ++j) {
        double v0[16], v1[16];
        if (!wg_poll_tiles(v0, g.A, a * TB, j * TB, v1, g.A, b * TB, j * TB, (a != b) ? 2 : 1, g.ld, g)) return false;
        if (worker) {
            const int i = tid & 63, k0 = tid >> 6;            // panel[k * PS + i] = L[(row0 + i), (col0 + k)]: tile_poll's layout
#pragma unroll
            for (int u = 0; u < 16; ++u) P0[(k0 + 4 * u) * PS + i] = v0[u];
            if (a != b) {
#pragma unroll
                for (int u = 0; u < 16; ++u) P1[(k0 + 4 * u) * PS + i] = v1[u];
            }
        }
        __syncthreads();
        if (worker) tile_mma(acc, P0, (a != b) ? P1 : P0, TB, lane, wr, wc);
        __syncthreads();
    }
    return true;
}

// near owner (a, b), b = a - 1 or a, a >= 3: the tile's updates from the columns <= a - 3, published into Far.  It also
// refills the next launch's mailboxes at its tile's position.
__device__ void chain_near_owner(const ChainArgs& g, int a, int b, double* lds, double* tiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    double* Xt = tiles + TB * LT;
    tile_reset(g.A_next, g.ld, a, b, tid);
    tile_reset(g.Far_next, g.ld, a, b, tid);
    tile_reset(g.Ship_next, g.ld, a, b, tid);
    tile_raw_s(Xt, g, a, b, tid);
    Acc4 acc;
    acc_zero(acc);
    if (!accumulate_updates(acc, a, b, a - 2, g, lds)) return;
    __syncthreads();
    tile_sub_acc(Xt, acc, lane, wr, wc);
    __syncthreads();
    tile_s2g_c(Xt, g.Far, g.ld, a * TB, b * TB, tid);
}

// solver (a, b), a >= b + 2.  After the accumulation phase its waves are independent agents (no workgroup barrier): wave w
// owns rows 16 w .. of the tile, keeps its own copy of the incoming 16-column block of L_bb in LDS, solves, publishes from
// registers, and applies the block's rank-16 update to the blocks to its right.
// The FEEDER (a == b + 2) runs with all eight waves: the critical workgroup adopts row a at step b + 1 and needs tile
// (a, b+1) updated through column b (urgent: its lower-tile waves wait for it) and the far part of (a, a) updated through
// column b (one step later).  Both column-b terms, P = L_ab L_{b+1,b}^T (16 blocks) and P2 = L_ab L_ab^T (10 lower blocks),
// are collected slice by slice as the feeder's own column blocks and those of L_{b+1,b} appear, in registers, spread over the
// eight waves; the far parts come from the near owners (or, for row 2, from the input), and the results leave from registers.
constexpr int SBS = 17;                                   // row stride of a wave's private block column
constexpr int SOLVER_PRIV = 64 * SBS + DPB + 16;          // block column (<= 64 rows), prepared diagonal block, 1 / diag

__device__ void chain_solver(const ChainArgs& g, int a, int b, double* lds, double* tiles, int* sy) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = (wave >> 1) & 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const bool feeder = (a == b + 2), worker = tid < 256;
    const size_t ld = g.ld;
    double* Xt = tiles + TB * LT;
    if (worker) {
        tile_reset(g.A_next, ld, a, b, tid);
        if (feeder && a == 2) {                            // row 2 has no near owners: their share of the refill
            tile_reset(g.A_next, ld, 2, 1, tid); tile_reset(g.A_next, ld, 2, 2, tid);
            tile_reset(g.Ship_next, ld, 2, 1, tid); tile_reset(g.Ship_next, ld, 2, 2, tid);
        }
        tile_raw_s(Xt, g, a, b, tid);
    }
    {
        Acc4 acc;
        acc_zero(acc);
        if (!accumulate_updates(acc, a, b, b, g, lds)) return;
        __syncthreads();
        if (worker) tile_sub_acc(Xt, acc, lane, wr, wc);
    }
    if (tid < 16) sy[tid] = 0;
    __syncthreads();
#define FD_TRACE(slot) do { if (g.trace && feeder && lane == 0) ((CH_GLOBAL long long*)g.trace)[(a - 1) * 32 + (slot)] = realtime_ticks(); } while (0)
    if (wave == 0) FD_TRACE(13);
    // ---- no workgroup barrier below ----
    int* rowDoneF = sy + 1;                                // [4] column blocks solved by solver wave w
    int* abortl = sy + 11;
#define CH_FAIL() do { chain_abort(g, abortl); return; } while (0)
    if (wave < 4) {
        // ---- solver wave: the poll of column block cb + 1 is in flight while block cb is solved
        double* priv = lds + wave * SOLVER_PRIV;
        double* Sw = priv;
        double* Dpw = priv + 64 * SBS;
        double* rvw = Dpw + DPB;
        double* Rw = Xt + 16 * wave * LT;
        double v[16], rv;
        // column block cb of L_bb is complete when the words its writers store last have appeared: the pivot block's and the
        // solved blocks' last words (rows 16 (cb + l) + 15, column 16 cb + 15) and the last 1 / diag of the block
        auto probe_block = [&](int cb) -> bool {
            const double* pa = nullptr;
            if (lane < 4 && cb + lane < 4) pa = g.A + (size_t)(b * TB + 16 * cb + 15) * ld + b * TB + 16 * (cb + lane) + 15;
            else if (lane == 4) pa = g.rinv_all + b * TB + 16 * cb + 15;
            return wave_probe(pa, g);
        };
        auto issue_block = [&](int cb) {                   // registers <- rows 16 cb .. 63 of column block cb of L_bb, 1 / diag
            // (clamped addresses, no select on the loaded values: `cond ? load : 0` makes the compiler wait for every load before
            // it issues the next one -- 17 round trips in a row, measured at 4 us; the surplus lanes fill unused rows of Sw)
            const int rowi = min(16 * cb + lane, TB - 1);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = ldc(g.A + (size_t)(b * TB + 16 * cb + c) * ld + b * TB + rowi);
            rv = ldc(g.rinv_all + b * TB + 16 * cb + (lane & 15));
        };
        auto incomplete = [&]() -> bool {
            bool bad = is_sent(rv);
#pragma unroll
            for (int c = 0; c < 16; ++c) bad |= is_sent(v[c]);
            return __any(bad);
        };
        if (!probe_block(0)) CH_FAIL();
        issue_block(0);
        for (int cb = 0; cb < 4; ++cb) {
            int it = 0;
            while (incomplete()) {                         // (the load issued a block ahead came too early, or a straggling store)
                if (!spin_more(it, g) || !probe_block(cb)) CH_FAIL();
                issue_block(cb);
            }
            if (wave == 0) FD_TRACE(16 + 4 * cb);
#pragma unroll
            for (int c = 0; c < 16; ++c) Sw[lane * SBS + c] = v[c];
            if (lane < 16) rvw[lane] = rv;
            if (cb < 3) issue_block(cb + 1);                              // in flight while this block is worked on; checked next round
            {
                const int pr = lane & 15;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * i + (lane >> 4);
                    Dpw[pr * DPS + c] = (c < pr) ? Sw[pr * SBS + c] * rvw[pr] : 0.0;
                }
            }
            {
                double x[4];
                const int rr = lane >> 2, q = lane & 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = Rw[rr * LT + 16 * cb + 4 * i + q];
                solve16(x, Dpw, rvw, q, [](auto) {});
#pragma unroll
                for (int i = 0; i < 4; ++i) Rw[rr * LT + 16 * cb + 4 * i + q] = x[i];
            }
            lds_set(rowDoneF + wave, cb + 1);                            // (wave 4 + w publishes the block)
            if (wave == 0 && cb == 3) FD_TRACE(9);
            if (wave == 0) FD_TRACE(17 + 4 * cb);
            {
                const double* ap = Rw + li * LT + 16 * cb + lk;
                const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
                const d4 Z = (d4){0.0, 0.0, 0.0, 0.0};
                d4 u[3];
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const int c = cb + 1 + e;
                    u[e] = Z;
                    if (c < 4) {
                        const double* bp = Sw + (16 * (c - cb) + li) * SBS + lk;
                        const d4 p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bp[0], Z, 0, 0, 0);
                        const d4 p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bp[4], Z, 0, 0, 0);
                        const d4 p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bp[8], Z, 0, 0, 0);
                        const d4 p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, bp[12], Z, 0, 0, 0);
                        u[e] = (p0 + p1) + (p2 + p3);
                    }
                }
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const int c = cb + 1 + e;
                    if (c < 4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Rw[(lk + 4 * r) * LT + 16 * c + li] -= u[e][r];
                    }
                }
            }
            if (wave == 0) FD_TRACE(19 + 4 * cb);
        }
        if (!feeder) return;
        // The far part of tile (a, a) needs nothing from the critical workgroup: P2 = L_ab L_ab^T, the wave's share of its ten
        // lower blocks (3, 3, 2, 2), all four slices at once, then far - P2 leaves through the wave's staging block.
        //  wave 0: (0,0) (3,0) (3,3)   wave 1: (1,0) (1,1) (3,1)   wave 2: (2,0) (2,1)   wave 3: (2,2) (3,2)
        // (computed, not tabulated: a table indexed by the wave number lands in scratch memory)
        const int nq = (wave < 2) ? 3 : 2;
        auto q_block = [&](int e, int& R, int& C) {
            if (e == 0) { R = (wave == 3) ? 2 : wave; C = (wave == 3) ? 2 : 0; }
            else if (e == 1) { R = (wave == 0 || wave == 3) ? 3 : wave; C = (wave == 0) ? 0 : ((wave == 3) ? 2 : 1); }
            else { R = 3; C = (wave == 0) ? 3 : 1; }
        };
        if (b >= 1 && !wave_probe(lane < 4 ? tile_probe_addr(g.Far, ld, a * TB, a * TB, lane) : nullptr, g)) CH_FAIL();
        double* stage = tiles + wave * (16 * 17);          // (the S half of `tiles` is free in a solver)
#pragma unroll 1
        for (int e = 0; e < nq; ++e) {
            int R, C;
            q_block(e, R, C);
            d4 f;                                          // the far part of the block, accumulator layout
            if (b >= 1) {
                int it = 0;
                for (;;) {
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f[r] = ldc(g.Far + (size_t)(a * TB + 16 * C + li) * ld + a * TB + 16 * R + lk + 4 * r);
                        bad |= is_sent(f[r]);
                    }
                    if (!__any(bad)) break;
                    if (!spin_more(it, g)) CH_FAIL();
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] = raw_entry(g, a * TB + 16 * R + lk + 4 * r, a * TB + 16 * C + li);
            }
            if (!lwait(rowDoneF + R, 4, abortl) || !lwait(rowDoneF + C, 4, abortl)) CH_FAIL();
            const double* ap = Xt + (16 * R + li) * LT + lk;
            const double* bp = Xt + (16 * C + li) * LT + lk;
            const d4 Z = (d4){0.0, 0.0, 0.0, 0.0};
            d4 q0 = Z, q1 = Z, q2 = Z, q3 = Z;
#pragma unroll
            for (int k = 0; k < 64; k += 16) {
                q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k], bp[k], q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k + 4], bp[k + 4], q1, 0, 0, 0);
                q2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k + 8], bp[k + 8], q2, 0, 0, 0);
                q3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k + 12], bp[k + 12], q3, 0, 0, 0);
            }
            f -= (q0 + q1) + (q2 + q3);
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(lk + 4 * r) * 17 + li] = f[r];
            publish16_lds(stage, 17, g.Ship, ld, a * TB + 16 * R, a * TB + 16 * C);
        }
        if (wave == 0) FD_TRACE(12);
        return;
    }
    // ---- waves 4..7.  In a plain solver wave 4 + w publishes the blocks solver wave w has solved (a 1 KB write-through store
    // costs its wave ~0.4 us: off the solving waves).  In a feeder it is also product wave x = w: block row x of
    // P = L_ab L_{b+1,b}^T (4 blocks), collected slice by slice as both factors appear; X' = far - P leaves at the end.
    const int x = wave - 4;
    double* Rx = Xt + 16 * x * LT;
    if (!feeder) {
        for (int cb = 0; cb < 4; ++cb) {
            if (!lwait(rowDoneF + x, cb + 1, abortl)) CH_FAIL();
            publish16_lds(Rx + 16 * cb, LT, g.A, ld, a * TB + 16 * x, b * TB + 16 * cb);
        }
        return;
    }
    // the accumulators start out as the far part of tile (a, b+1) (accumulator layout: element (16 x + lk + 4 r, 16 e + li)) and
    // the slices of P are subtracted from them: no second set of registers (this kernel must not spill)
    d4 accP[4];
    if (b >= 1) {
        if (!wave_probe(lane < 4 ? tile_probe_addr(g.Far, ld, a * TB, (b + 1) * TB, lane) : nullptr, g)) CH_FAIL();
        int it = 0;
        for (;;) {
            bool bad = false;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accP[e][r] = ldc(g.Far + (size_t)((b + 1) * TB + 16 * e + li) * ld + a * TB + 16 * x + lk + 4 * r);
                    bad |= is_sent(accP[e][r]);
                }
            if (!__any(bad)) break;
            if (!spin_more(it, g)) CH_FAIL();
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) accP[e][r] = raw_entry(g, a * TB + 16 * x + lk + 4 * r, (b + 1) * TB + 16 * e + li);
    }
    for (int s = 0; s < 4; ++s) {
        if (!lwait(rowDoneF + x, s + 1, abortl)) CH_FAIL();
        publish16_lds(Rx + 16 * s, LT, g.A, ld, a * TB + 16 * x, b * TB + 16 * s);       // the partner's block, for the other rows' updates
        if (s == 3 && x == 0) FD_TRACE(14);
        // slice s of L_{b+1,b} (the critical workgroup's lower tile), straight into MFMA operand layout: 4 blocks x 4 k-steps
        double bv[4][4];
        if (!wave_probe(lane < 4 ? g.A + (size_t)(b * TB + 16 * s + 15) * ld + (b + 1) * TB + 16 * lane + 15 : nullptr, g)) CH_FAIL();
        int it = 0;
        for (;;) {
            bool bad = false;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    bv[e][k4] = ldc(g.A + (size_t)(b * TB + 16 * s + 4 * k4 + lk) * ld + (b + 1) * TB + 16 * e + li);
                    bad |= is_sent(bv[e][k4]);
                }
            if (!__any(bad)) break;
            if (!spin_more(it, g)) CH_FAIL();
        }
        const d4 Z = (d4){0.0, 0.0, 0.0, 0.0};
        const double* ap = Rx + li * LT + 16 * s + lk;
        const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const d4 p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv[e][0], Z, 0, 0, 0);
            const d4 p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv[e][1], Z, 0, 0, 0);
            const d4 p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bv[e][2], Z, 0, 0, 0);
            const d4 p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, bv[e][3], Z, 0, 0, 0);
            accP[e] -= (p0 + p1) + (p2 + p3);
        }
    }
    // X' = far - P, rows 16 x .. of tile (a, b+1), through the wave's staging block
    if (x == 0) FD_TRACE(10);
    {
        double* stage = tiles + wave * (16 * 17);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(lk + 4 * r) * 17 + li] = accP[e][r];
            publish16_lds(stage, 17, g.Ship, ld, a * TB + 16 * x, (b + 1) * TB + 16 * e);
        }
    }
    if (x == 0) FD_TRACE(11);
#undef FD_TRACE
#undef CH_FAIL
}

// inverter: W_jj = L_jj^-1 as the diagonal tiles appear (plain stores: the consumers are later launches).  It also refills
// the part of the next launch's mailboxes that no other helper covers: the tiles the critical workgroup owns from the start.
__device__ void chain_inverter(const ChainArgs& g, double* lds, double* tiles, double* rinv) {
    const int tid = threadIdx.x;
    double* S = tiles;
    double* Wt = tiles + TB * LT;
    tile_reset(g.A_next, g.ld, 0, 0, tid);
    if (g.Tn > 1) { tile_reset(g.A_next, g.ld, 1, 0, tid); tile_reset(g.A_next, g.ld, 1, 1, tid); }
    if (g.Tn == 3) {                                       // (with Tn >= 4 the feeder of row 2 exists and does these)
        tile_reset(g.A_next, g.ld, 2, 1, tid); tile_reset(g.A_next, g.ld, 2, 2, tid);
    }
    if (g.rinv_next) for (int e = tid; e < g.ld; e += 256) stc(g.rinv_next + e, sentinel());
    for (int j = 0; j < g.Tn; ++j) {
        double v[16], dummy[16];
        if (!wg_poll_tiles(v, g.A, j * TB, j * TB, dummy, g.A, 0, 0, 1, g.ld, g)) return;
        {
            const int r = tid & 63, c0 = tid >> 6;
#pragma unroll
            for (int u = 0; u < 16; ++u) S[r * LT + c0 + 4 * u] = v[u];
        }
        int it = 0;
        for (;;) {                                         // 1 / diag of this tile
            const double rv = (tid < TB) ? ldc(g.rinv_all + j * TB + tid) : 0.0;
            if (!__syncthreads_or(is_sent(rv) ? 1 : 0)) { if (tid < TB) rinv[tid] = rv; break; }
            if (++it > CH_SPIN_LIMIT) { if (tid == 0) chain_abort(g, nullptr); return; }
        }
        __syncthreads();
        trtri_tile(S, rinv, Wt, lds);
        tile_s2g(Wt, g.Winv, g.ld, j * TB, j * TB);
        __syncthreads();
    }
}

// roles by block index: 0 critical | solvers (a, b), a = 2.., b = 0..a-2 | near owners a = 3.., b = a-1, a | inverter
__host__ __device__ inline int chain_n_solvers(int Tn) { return Tn >= 2 ? (Tn - 1) * (Tn - 2) / 2 : 0; }
__host__ __device__ inline int chain_n_near(int Tn) { return Tn >= 4 ? 2 * (Tn - 3) : 0; }
__host__ __device__ inline int chain_blocks(int Tn, bool with_inverse) {
    return 1 + chain_n_solvers(Tn) + chain_n_near(Tn) + (with_inverse ? 1 : 0);
}

__global__ void __launch_bounds__(CH_THREADS) k_chol_chain(const ChainArgs* __restrict__ gp) {
    const ChainArgs& g = *gp;
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    __shared__ __attribute__((aligned(16))) double tiles[2 * TB * LT];
    __shared__ double dprep[4 * DPB];
    __shared__ double rinv[TB];
    __shared__ int sy[16];
    const int bid = blockIdx.x;
    if (g.form.stats && bid == (int)gridDim.x - 1) {
        // xi = xi0 + vec(B W), for the forward solve that follows the factorisation (later launches read it)
        const double* B = g.form.stats + (size_t)g.form.Mp * g.form.Mp;
        for (int gi = threadIdx.x; gi < g.ld; gi += CH_THREADS) {
            double v = 0.0;
            if (gi < g.form.Q) {
                const int aa = gi / g.form.M, i = gi % g.form.M;
                v = (g.form.prior_form == 1) ? g.form.xi0[gi] : 0.0;
                for (int e = 0; e < g.form.d_out; ++e) v = fma(B[(size_t)e * g.form.Mp + i], g.form.P->W[e + aa * g.form.d_out], v);
            }
            g.form.xi[gi] = v;
        }
    }
    if (bid == 0) {
        if (g.abortw_next && threadIdx.x == 0) fl_store(g.abortw_next, 0);
        chain_critical(g, lds, tiles, dprep, rinv, sy);
        return;
    }
    int e = bid - 1;
    const int ns = chain_n_solvers(g.Tn), nn = chain_n_near(g.Tn);
    if (e < ns) {
        int a = 2, b = e;
        while (b > a - 2) { b -= a - 1; ++a; }
        chain_solver(g, a, b, lds, tiles, sy);               // (eight waves: four solve, four publish / form the feeder's products)
        return;
    }
    if (threadIdx.x >= 256) return;                        // the other helpers are four waves
    if (e < ns + nn) {
        e -= ns;
        const int a = 3 + (e >> 1);
        chain_near_owner(g, a, a - 1 + (e & 1), lds, tiles);
    } else {
        chain_inverter(g, lds, tiles, rinv);
    }
}

// fill a mailbox buffer with sentinels (handle creation; the launches keep them up afterwards)
__global__ void k_chain_fill(double* p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = sentinel();
}

// gate in front of a chain launch on a side stream: returns when *flag >= need (set by the sweep's streaming SYRK once its
// single resident round of workgroups has been dispatched), so that the chain's workgroups -- each takes a whole CU's LDS --
// do not occupy CUs the SYRK's round was sized for.  Bounded: after ~1 s it lets the chain go anyway.
__global__ void k_chain_gate(const long long* flag, long long need) {
    int it = 0;
    while (fl_load(flag) < need && ++it < CH_SPIN_LIMIT) __builtin_amdgcn_s_sleep(8);
}

// the inverse-factor / Sigma / forward-solve roles of k_potrf_step's launch j, without the factorisation (which the
// persistent launch has finished): block row j - 1 of W finished, block row j pre-accumulated, Sigma's row j - 2, block j - 1 of t
__global__ void __launch_bounds__(256) k_chain_extras(double* __restrict__ A, int ld, int j, int Tn, double* __restrict__ Winv,
                                                      double* __restrict__ Sacc, const double* __restrict__ tv_xi,
                                                      double* __restrict__ tv_t) {
    __shared__ __attribute__((aligned(16))) double lds[2 * TB * PS];
    __shared__ __attribute__((aligned(16))) double tiles[TB * LT];
    int e = blockIdx.x;
    const int nfin = (j >= 2) ? 2 * (j - 1) : 0;
    const int npre = (j < Tn) ? nfin : 0;
    const int nsig = (Sacc && j >= 2 && j < Tn) ? (j - 1) * j / 2 : 0;     // (the last launch leaves row Tn - 2 to the product launch, see k_potrf_step)
    if (e < nfin) winv_row_tile(A, Winv, ld, j - 1, e >> 1, e & 1, lds, 2);
    else if (e < nfin + npre) { e -= nfin; winv_row_tile(A, Winv, ld, j, e >> 1, e & 1, lds, 1); }
    else if (e < nfin + npre + nsig) {
        int I, J;
        tile_from_index(e - nfin - npre, I, J);
        sigma_row_tile(Winv, Sacc, ld, j - 2, I, J, lds, tiles);
    } else {
        tvec_role<4>(A, Winv, tv_xi, tv_t, ld, j - 1, lds);
    }
}

}  // namespace sgp
