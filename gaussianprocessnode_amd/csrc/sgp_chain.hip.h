// sgp_chain.hip.h -- the blocked Cholesky of the sweep's two M x M chains (K_uu and Lambda) as ONE persistent launch.
//
// Why.  With one launch per 64-column step (k_potrf_step) a step costs ~19.5 us at M = 512: tiles in (3.3) + factor the
// diagonal tile (9.1) + solve the tile below it (5.4) + tiles out (0.65) + launch gap (1.5), of which only the 64-pivot
// chain (~3.4 us) is inherently sequential.  Every cross-workgroup hand-off on gfx950 (write-through stores, flag, poll,
// L2-served loads) costs 2-3 us, about what a kernel boundary costs, so replacing boundaries by flags one for one gains
// nothing: the hand-offs have to leave the critical path.
//
// How.  One CRITICAL workgroup (8 waves) owns the diagonal tile (j, j) AND the tile below it (j+1, j) of every step and
// keeps them in LDS from step to step: it factors the 128 x 64 panel (the solve of the lower tile rides in the waves that
// idle during the pivot runs), forms the next diagonal tile's update L_{j+1,j} L_{j+1,j}^T on the matrix cores while it
// goes, and rolls over to step j+1 without touching memory.  Its eight waves are independent agents that synchronise
// through progress counters in LDS -- there is no s_barrier in the loop, so a lower tile that arrives late delays nothing
// but its own rows.  Everything else is dataflow around it, with slack:
//   solver (a, b), a >= b+2 : accumulates A_ab -= L_aj L_bj^T as the columns j < b are published, then solves against
//                             L_bb as ITS 16-column blocks arrive (right-looking, so only one 16-pivot solve follows the
//                             last block), publishes L_ab;
//   feeder = solver (a, a-2): additionally forms the two products of its fresh L_{a,a-2} that the critical workgroup
//                             needs one step later -- tile (a, a-1) and the far part of (a, a) -- and ships them;
//   near owner (a, a-1|a)   : accumulates those two tiles' updates from the columns <= a-3 for the feeder;
//   inverter                : L_jj^-1 for the inverse factor.
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", table row 1): every handed-off byte is stored and loaded with
// agent-scope relaxed atomics (global_store/load ... sc1), each storing wave drains its stores (s_waitcnt vmcnt(0)) before
// the flag it signals with (per-wave flags in the critical workgroup; barrier + one flag elsewhere), consumers poll with
// sc1 loads.  Flags carry epoch * 8 + count, so they are never reset.  Every spin is bounded: after CH_SPIN_LIMIT polls a
// waiter raises the abort word, everybody unwinds, and the host reports an error instead of a hung GPU.
// All arithmetic is in a fixed order: results are bitwise reproducible run to run.
#pragma once
#include "sgp_kernels.hip.h"

namespace sgp {

constexpr int CH_TMAX = 12;                 // tile rows the persistent path supports (flag table size)
constexpr int CH_THREADS = 512;             // the critical workgroup uses all 8 waves, the helpers the first 4
constexpr int CH_SPIN_LIMIT = 1 << 21;      // polls (each >= ~0.3 us) before a waiter gives up: ~1 s

// flag table (long long each)
constexpr int CH_F_CB = 0;                              // [j][wave 0..7]: blocks published by that wave of the critical WG
constexpr int CH_F_FL = CH_F_CB + CH_TMAX * 8;          // [a][b]: solver tile L_ab published
constexpr int CH_F_SHIP = CH_F_FL + CH_TMAX * CH_TMAX;  // [a]: feeder of row a shipped tiles (a, a-1) and (a, a)
constexpr int CH_F_SHIP2 = CH_F_SHIP + CH_TMAX;         // [a]: ... and the far part of (a, a), one step less urgent
constexpr int CH_F_FAR = CH_F_SHIP2 + CH_TMAX;          // [a][2]: near owner's far tile ready
constexpr int CH_F_W = CH_F_FAR + CH_TMAX * 2;          // [i][c]: tile W_ic of the inverse factor published
constexpr int CH_F_ABORT = CH_F_W + CH_TMAX * CH_TMAX;
constexpr int CH_F_GATE = CH_F_ABORT + 1;               // set by the sweep's streaming SYRK once its workgroups are resident
constexpr int CH_F_COUNT = CH_F_GATE + 1;

struct ChainArgs {
    double* A;               // out: L (lower tiles); inside the launch it also carries the tiles shipped to the critical workgroup
    const double* Ain;       // in (source 0): the matrix, in a buffer of its own -- NEVER the same memory as A: every byte of A is
                             // handed between workgroups with sc1 accesses, and a plain load of the same line by anybody on the
                             // reader's XCD would let its L2 serve a stale copy
    int ld, Tn;
    int* info;
    int n_valid;
    double* Winv;            // out (may be nullptr): diagonal tiles of L^-1
    double* Far;             // scratch matrix (ld x ld): far parts of the near-diagonal tiles, at their home positions
    double* rinv_all;        // scratch (ld): 1 / diag(L), published with the diagonal blocks
    long long* flags;        // CH_F_COUNT words, zero at creation
    long long epoch;         // > 0, grows with every launch that uses `flags`
    LamForm form;            // source 1 (form.stats != nullptr): Lambda = Lambda0 + W (x) Psi2 evaluated on the fly, index-reversed
    long long* trace;        // diagnostics (may be nullptr): [step][8] 100 MHz ticks of the critical workgroup, see sgp_get_chain_trace
    const double* Xus;       // source 2 (Xus != nullptr): K_uu + jitter I from the scaled inducing inputs (D x ld SoA), pad = identity
    const Params* P;
    int M, D;
};

__device__ __forceinline__ double ldc(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long fl_load(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fl_store(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }

// ---- bounded waits ------------------------------------------------------------------------------------------------
// wave-level wait on up to 8 flags at base[lane], lanes [lo, hi]: every lane of the wave returns the same answer
__device__ __forceinline__ bool gwait_lanes(const long long* base, int lo, int hi, long long need, long long* flags, long long epoch) {
    const int lane = threadIdx.x & 63;
    const bool mine = lane >= lo && lane <= hi;
    int it = 0;
    for (;;) {
        const long long v = mine ? fl_load(base + lane) : need;
        if (__all(v >= need)) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 31) == 0) {
            if (fl_load(flags + CH_F_ABORT) == epoch) return false;
            if (it > CH_SPIN_LIMIT) { fl_store(flags + CH_F_ABORT, epoch); return false; }
        }
    }
    compiler_fence();
    return true;
}
__device__ __forceinline__ bool gwait(const long long* f, long long need, long long* flags, long long epoch) {
    return gwait_lanes(f - 0, 0, 0, need, flags, epoch);
}
// LDS progress counter of the critical workgroup (wave-level)
__device__ __forceinline__ int lds_get(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_set(int* p, int v) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_inc(int* p) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// ---- tile movement with coherent (sc1) accesses; t = thread index within the 256 threads that move the tile -----------
// LDS tile S[r * LT + c]  <-  global column-major tile at (row0, col0)
__device__ __forceinline__ void tile_g2s_c(double* S, const double* A, size_t ld, int row0, int col0, int t) {
    const int r = t & 63, c0 = t >> 6;
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = ldc(A + (size_t)(col0 + c0 + 4 * u) * ld + row0 + r);
#pragma unroll
    for (int u = 0; u < 16; ++u) S[r * LT + c0 + 4 * u] = v[u];
}
__device__ __forceinline__ void tile_s2g_c(const double* S, double* A, size_t ld, int row0, int col0, int t) {
    const int r = t & 63, c0 = t >> 6;
#pragma unroll
    for (int u = 0; u < 16; ++u) stc(A + (size_t)(col0 + c0 + 4 * u) * ld + row0 + r, S[r * LT + c0 + 4 * u]);
}
// MFMA operand panel P[k * PS + i] = G[(row0 + i) + (col0 + k) * ld], 64 x 64
__device__ __forceinline__ void panel_g2s_c(double* P, const double* G, size_t ld, int row0, int col0, int t) {
    const int i = t & 63, k0 = t >> 6;
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = ldc(G + (size_t)(col0 + k0 + 4 * u) * ld + row0 + i);
#pragma unroll
    for (int u = 0; u < 16; ++u) P[(k0 + 4 * u) * PS + i] = v[u];
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
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Rw[(lk + 4 * r) * LT + 16 * c + li] -= acc[r];
}

// publish the wave's 16 x 16 block (global rows grow0.., columns gcol0..) from registers
__device__ __forceinline__ void publish_solved16(double* A, size_t ld, int grow0, int gcol0, const double (&x)[4]) {
    const int lane = threadIdx.x & 63, rr = lane >> 2, q = lane & 3;           // solve coordinates
#pragma unroll
    for (int i = 0; i < 4; ++i) stc(A + (size_t)(gcol0 + 4 * i + q) * ld + grow0 + rr, x[i]);
}
__device__ __forceinline__ void publish_pivot16(double* A, size_t ld, int grow0, int gcol0, const double (&l)[4]) {
    const int lane = threadIdx.x & 63, pr = lane & 15, pq = lane >> 4;         // pivot coordinates
#pragma unroll
    for (int i = 0; i < 4; ++i) stc(A + (size_t)(gcol0 + 4 * i + pq) * ld + grow0 + pr, l[i]);
}
__device__ __forceinline__ void publish_zero16(double* A, size_t ld, int grow0, int gcol0) {
    const int lane = threadIdx.x & 63, pr = lane & 15, pq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) stc(A + (size_t)(gcol0 + 4 * i + pq) * ld + grow0 + pr, 0.0);
}

// ---- the critical workgroup ---------------------------------------------------------------------------------------------
// LDS of the launch (shared with the helper roles): `lds` (2 x 64 x PS doubles) and `tiles` (2 x 64 x LT).
// Buffers: B0 = tiles, B1 = tiles + 64 LT (always the lower tile X), B2 = lds.  S alternates between B0 and B2; the other
// one of the pair holds the next diagonal tile while its update is being collected.
// D blocks (the 16 x 16 blocks (R, C), R >= C, of X X^T) are owned by the waves with slack: 1, 2 and the X waves.
// (the lower tile usually arrives late, so its waves' phases are what the next step waits for: they carry one block each,
// the S waves -- idle once their pivot run is over -- two; wave 3 pivots last and carries none)
__device__ __forceinline__ int d_block_count(int wave) { return wave == 3 ? 0 : (wave < 3 ? 2 : 1); }
__device__ __forceinline__ void d_block(int wave, int e, int& R, int& C) {
    //  wave 0: (0,0) (1,0)   wave 1: (1,1) (2,0)   wave 2: (2,1) (2,2)   waves 4..7: (3,0) (3,1) (3,2) (3,3)
    switch (wave * 2 + e) {
        case 0: R = 0; C = 0; break;   case 1: R = 1; C = 0; break;
        case 2: R = 1; C = 1; break;   case 3: R = 2; C = 0; break;
        case 4: R = 2; C = 1; break;   case 5: R = 2; C = 2; break;
        case 8: R = 3; C = 0; break;   case 10: R = 3; C = 1; break;
        case 12: R = 3; C = 2; break;  case 14: R = 3; C = 3; break;
        default: R = 0; C = 0; break;
    }
}

// a waiter gave up (deadlock guard) or saw somebody else give up: raise the abort word for everybody, mark the chain's status
__device__ __forceinline__ void chain_abort(const ChainArgs& g, int* abortl) {
    if ((threadIdx.x & 63) == 0) {
        fl_store(g.flags + CH_F_ABORT, g.epoch);
        atomicExch(g.info, -1);
        if (abortl) __hip_atomic_store(abortl, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
    long long* F = g.flags;
    const long long ep8 = g.epoch * 8;
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
#define CH_FAIL() do { chain_abort(g, abortl); return; } while (0)
#define CH_TRACE(slot) do { if (g.trace && lane == 0) g.trace[j * 8 + (slot)] = realtime_ticks(); } while (0)
    for (int j = 0; j < Tn; ++j) {
        double* S = (j & 1) ? B2 : B0;
        double* Dt = (j & 1) ? B0 : B2;
        double* X = B1;
        const bool has_x = j + 1 < Tn;
        const int j0 = j * TB;
        long long* myflag = F + CH_F_CB + j * 8 + wave;
        int published = 0;                                 // blocks of this wave's rows handed to the memory system
        d4 dacc[2];
        dacc[0] = (d4){0.0, 0.0, 0.0, 0.0};
        dacc[1] = dacc[0];
        // slice s of the wave's D blocks: needs column block s of both row blocks of X
        auto d_slice = [&](int s) -> bool {
            for (int e = 0; e < nd; ++e) {
                int R, C;
                d_block(wave, e, R, C);
                if (!lwait(rowDone + 4 + R, 4 * j + s + 1, abortl)) return false;
                if (!lwait(rowDone + 4 + C, 4 * j + s + 1, abortl)) return false;
                const double* ap = X + (16 * R + li) * LT + 16 * s + lk;
                const double* bp = X + (16 * C + li) * LT + 16 * s + lk;
                const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
                const double b0 = bp[0], b1 = bp[4], b2 = bp[8], b3 = bp[12];
                d4 acc = dacc[e];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
                dacc[e] = acc;
            }
            return true;
        };
        auto d_tail = [&]() -> bool {
            if (nd > 0) {
                if (!lwait(dtReady, j + 1, abortl)) return false;
                for (int e = 0; e < nd; ++e) {
                    int R, C;
                    d_block(wave, e, R, C);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Dt[(16 * R + lk + 4 * r) * LT + 16 * C + li] -= dacc[e][r];
                }
            }
            lds_inc(dDone);
            return true;
        };
        auto flag_published = [&]() {                       // after the wave's stores have drained
            drain_stores();
            if (lane == 0) fl_store(myflag, ep8 + published);
        };
        if (swave) {
            double* Rw = S + 16 * w * LT;
            if (j > 0 && !lwait(dDone, 8 * j, abortl)) CH_FAIL();          // S of this step is final
            if (wave == 0) CH_TRACE(0);
            for (int cb = 0; cb < w; ++cb) {
                if (!lwait(runDone, 4 * j + cb + 1, abortl)) CH_FAIL();
                double x[4];
                solve_block16(Rw, cb, dprep, rinv, x);
                lds_set(rowDone + wave, 4 * j + cb + 1);
                publish_solved16(g.A, ld, j0 + 16 * w, j0 + 16 * cb, x);
                ++published;
                update_block16(Rw, Rw, w, cb);                              // own diagonal block first: the next pivot run waits for it
                for (int c = cb + 1; c < w; ++c) {
                    if (!lwait(rowDone + c, 4 * j + cb + 1, abortl)) CH_FAIL();
                    update_block16(Rw, S + 16 * c * LT, c, cb);
                }
                if (cb + 1 < w) flag_published();                           // (the wave that pivots next flags after its run)
            }
            double lreg[4], rl;
            pivot_block16(Rw, w, dprep, rinv, g.info, j0, g.n_valid, lreg, rl);
            lds_set(runDone, 4 * j + w + 1);
            if (wave == 3) CH_TRACE(1);
            publish_pivot16(g.A, ld, j0 + 16 * w, j0 + 16 * w, lreg);
            if (lane < 16) stc(g.rinv_all + j0 + 16 * w + lane, rl);
            for (int c = w + 1; c < 4; ++c) publish_zero16(g.A, ld, j0 + 16 * w, j0 + 16 * c);
            published = w + 1;
            flag_published();
            if (has_x) {
                // wave 0 also receives the far part of the next diagonal tile (j+1, j+1), shipped by the feeder of row j+1
                // one step ahead of its use; its buffer is last step's S (every wave has left that step: dDone).  The
                // shipment is polled between the D slices and waited for only at the end.
                bool need_dt = (wave == 0 && j > 0);
                auto receive_dt = [&]() {
#pragma unroll 1
                    for (int ch = 0; ch < 2; ++ch) {
                        double v[32];
#pragma unroll
                        for (int u = 0; u < 32; ++u) v[u] = ldc(g.A + (size_t)((j + 1) * TB + 32 * ch + u) * ld + (j + 1) * TB + lane);
#pragma unroll
                        for (int u = 0; u < 32; ++u) Dt[lane * LT + 32 * ch + u] = v[u];
                    }
                    lds_set(dtReady, j + 1);
                    need_dt = false;
                };
                for (int s = 0; s < 4; ++s) {
                    if (need_dt && fl_load(F + CH_F_SHIP2 + (j + 1)) >= ep8 + 4) { compiler_fence(); receive_dt(); }
                    if (!d_slice(s)) CH_FAIL();
                }
                if (need_dt) {
                    if (!gwait(F + CH_F_SHIP2 + (j + 1), ep8 + 4, F, g.epoch)) CH_FAIL();
                    receive_dt();
                }
                if (!d_tail()) CH_FAIL();
            }
        } else if (has_x) {
            double* Rw = X + 16 * w * LT;
            if (j > 0) {
                // the wave's 16 rows of tile (j+1, j), fully updated, from the feeder of row j+1 (everybody has finished
                // reading last step's X: dDone)
                if (!lwait(dDone, 8 * j, abortl)) CH_FAIL();
                if (wave == 4) CH_TRACE(5);
                if (!gwait(F + CH_F_SHIP + (j + 1), ep8 + 4, F, g.epoch)) CH_FAIL();
                if (wave == 4) CH_TRACE(2);
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = ldc(g.A + (size_t)(j0 + lk + 4 * u) * ld + (j + 1) * TB + 16 * w + li);
#pragma unroll
                for (int u = 0; u < 16; ++u) Rw[li * LT + lk + 4 * u] = v[u];
            }
            for (int cb = 0; cb < 4; ++cb) {
                if (!lwait(runDone, 4 * j + cb + 1, abortl)) CH_FAIL();
                double x[4];
                solve_block16(Rw, cb, dprep, rinv, x);
                lds_set(rowDone + wave, 4 * j + cb + 1);
                publish_solved16(g.A, ld, (j + 1) * TB + 16 * w, j0 + 16 * cb, x);
                ++published;
                for (int c = cb + 1; c < 4; ++c) {
                    if (!lwait(rowDone + c, 4 * j + cb + 1, abortl)) CH_FAIL();
                    update_block16(Rw, S + 16 * c * LT, c, cb);
                }
                if (!d_slice(cb)) CH_FAIL();
                flag_published();                                           // ~1 us after the stores: they have drained
            }
            if (wave == 4) CH_TRACE(3);
            if (!d_tail()) CH_FAIL();
            if (wave == 4) CH_TRACE(4);
        }
    }
#undef CH_TRACE
#undef CH_FAIL
}

// ---- helper roles (256 threads, workgroup barriers) -----------------------------------------------------------------------
// workgroup-level wait: wave 0 polls, everybody learns the outcome
__device__ __forceinline__ bool wg_wait_lanes(const long long* base, int lo, int hi, long long need, const ChainArgs& g, int* okw) {
    if (threadIdx.x < 64) {
        const bool ok = gwait_lanes(base, lo, hi, need, g.flags, g.epoch);
        if (!ok) chain_abort(g, nullptr);
        if (threadIdx.x == 0) *okw = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *okw != 0;
    __syncthreads();
    return ok;
}
// L_aj is published by the critical workgroup when a <= j + 1 (rows j: waves 0-3, rows j+1: waves 4-7), else by solver (a, j)
__device__ __forceinline__ bool wg_wait_L(int a, int j, const ChainArgs& g, int* okw) {
    const long long ep8 = g.epoch * 8;
    if (a == j) {                                          // wave w of the critical workgroup publishes w + 1 blocks of its rows
        if (threadIdx.x < 64) {
            bool ok = true;
            for (int w = 0; w < 4 && ok; ++w) ok = gwait(g.flags + CH_F_CB + j * 8 + w, ep8 + w + 1, g.flags, g.epoch);
            if (!ok) chain_abort(g, nullptr);
            if (threadIdx.x == 0) *okw = ok ? 1 : 0;
        }
        __syncthreads();
        const bool ok = *okw != 0;
        __syncthreads();
        return ok;
    }
    if (a == j + 1) return wg_wait_lanes(g.flags + CH_F_CB + j * 8, 4, 7, ep8 + 4, g, okw);
    return wg_wait_lanes(g.flags + CH_F_FL + a * CH_TMAX + j, 0, 0, ep8 + 4, g, okw);
}

// acc += L_aj L_bj^T for j in [0, jend)  (panels through `lds`; threads >= 256 -- the feeder's product waves -- only take
// part in the barriers)
__device__ __forceinline__ bool accumulate_updates(Acc4& acc, int a, int b, int jend, const ChainArgs& g, double* lds, int* okw) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = (wave >> 1) & 1, wc = wave & 1;
    const bool worker = tid < 256;
    double* P0 = lds;
    double* P1 = lds + TB * PS;
    for (int j = 0; j < jend; ++j) {
        if (!wg_wait_L(a, j, g, okw)) return false;
        if (a != b && !wg_wait_L(b, j, g, okw)) return false;
        if (worker) {
            panel_g2s_c(P0, g.A, g.ld, a * TB, j * TB, tid);
            if (a != b) panel_g2s_c(P1, g.A, g.ld, b * TB, j * TB, tid);
        }
        __syncthreads();
        if (worker) tile_mma(acc, P0, (a != b) ? P1 : P0, TB, lane, wr, wc);
        __syncthreads();
    }
    return true;
}

// near owner (a, b), b = a - 1 or a, a >= 3: the tile's updates from the columns <= a - 3, published into Far
__device__ void chain_near_owner(const ChainArgs& g, int a, int b, double* lds, double* tiles, int* okw) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    double* Xt = tiles + TB * LT;
    tile_raw_s(Xt, g, a, b, tid);
    Acc4 acc;
    acc_zero(acc);
    if (!accumulate_updates(acc, a, b, a - 2, g, lds, okw)) return;
    __syncthreads();
    tile_sub_acc(Xt, acc, lane, wr, wc);
    __syncthreads();
    tile_s2g_c(Xt, g.Far, g.ld, a * TB, b * TB, tid);
    drain_stores();
    __syncthreads();
    if (tid == 0) fl_store(g.flags + CH_F_FAR + a * 2 + (b - (a - 1)), g.epoch * 8 + 4);
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
    const long long ep8 = g.epoch * 8;
    double* Xt = tiles + TB * LT;
    if (worker) tile_raw_s(Xt, g, a, b, tid);
    {
        Acc4 acc;
        acc_zero(acc);
        if (!accumulate_updates(acc, a, b, b, g, lds, sy)) return;
        __syncthreads();
        if (worker) tile_sub_acc(Xt, acc, lane, wr, wc);
    }
    if (tid < 16) sy[tid] = 0;
    __syncthreads();
    // ---- no workgroup barrier below ----
    int* rowDoneF = sy + 1;                                // [4] column blocks solved by solver wave w
    int* cntX = sy + 5;                                    // waves whose stores of X' have drained
    int* cntD = sy + 6;                                    // ... of the far part of (a, a)
    int* cntL = sy + 7;                                    // solver waves whose stores of L_ab have drained
    int* abortl = sy + 11;
#define CH_FAIL() do { chain_abort(g, abortl); return; } while (0)
    // ---- the feeder's product blocks of this wave
    int pR[2], pC[2], qR[2], qC[2], nq = 0;
    d4 accP[2], accQ[2];
    double farP[2][4], farQ[2][4];
    if (feeder) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = 2 * wave + e;
            pR[e] = p >> 2; pC[e] = p & 3;
            accP[e] = (d4){0.0, 0.0, 0.0, 0.0};
            accQ[e] = accP[e];
            qR[e] = 0; qC[e] = 0;
        }
        auto lower = [](int q, int& R, int& C) { R = 0; while ((R + 1) * (R + 2) / 2 <= q) ++R; C = q - R * (R + 1) / 2; };
        lower(wave, qR[0], qC[0]);
        nq = 1;
        if (wave == 4 || wave == 5) { lower(4 + wave, qR[1], qC[1]); nq = 2; }      // blocks 8 and 9
        // far parts in accumulator layout: element (16 R + lk + 4 r, 16 C + li)
        if (b >= 1) {
            if (!gwait_lanes(g.flags + CH_F_FAR + a * 2, 0, 1, ep8 + 4, g.flags, g.epoch)) CH_FAIL();
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    farP[e][r] = ldc(g.Far + (size_t)((b + 1) * TB + 16 * pC[e] + li) * ld + a * TB + 16 * pR[e] + lk + 4 * r);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    farQ[e][r] = (e >= nq) ? 0.0 : ldc(g.Far + (size_t)(a * TB + 16 * qC[e] + li) * ld + a * TB + 16 * qR[e] + lk + 4 * r);
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) farP[e][r] = raw_entry(g, a * TB + 16 * pR[e] + lk + 4 * r, (b + 1) * TB + 16 * pC[e] + li);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) farQ[e][r] = (e >= nq) ? 0.0 : raw_entry(g, a * TB + 16 * qR[e] + lk + 4 * r, a * TB + 16 * qC[e] + li);
        }
    }
    int ns = 0;                                            // product slices this wave has done
    // slice s of the wave's product blocks; `blocking` = false: only if every input is there already
    auto do_slice = [&](int s, bool blocking) -> int {    // 1 done, 0 not ready, -1 abort
        const long long needC = ep8 + s + 1;
        if (!blocking) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (fl_load(g.flags + CH_F_CB + b * 8 + 4 + pC[e]) < needC) return 0;
                if (lds_get(rowDoneF + pR[e]) < s + 1) return 0;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (e < nq && (lds_get(rowDoneF + qR[e]) < s + 1 || lds_get(rowDoneF + qC[e]) < s + 1)) return 0;
            compiler_fence();
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (!gwait(g.flags + CH_F_CB + b * 8 + 4 + pC[e], needC, g.flags, g.epoch)) return -1;
                if (!lwait(rowDoneF + pR[e], s + 1, abortl)) return -1;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (e < nq && (!lwait(rowDoneF + qR[e], s + 1, abortl) || !lwait(rowDoneF + qC[e], s + 1, abortl))) return -1;
        }
        double bv[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                bv[e][k4] = ldc(g.A + (size_t)(b * TB + 16 * s + 4 * k4 + lk) * ld + (b + 1) * TB + 16 * pC[e] + li);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const double* ap = Xt + (16 * pR[e] + li) * LT + 16 * s + lk;
            d4 acc = accP[e];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * k4], bv[e][k4], acc, 0, 0, 0);
            accP[e] = acc;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e < nq) {
                const double* ap = Xt + (16 * qR[e] + li) * LT + 16 * s + lk;
                const double* bp = Xt + (16 * qC[e] + li) * LT + 16 * s + lk;
                d4 acc = accQ[e];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * k4], bp[4 * k4], acc, 0, 0, 0);
                accQ[e] = acc;
            }
        }
        return 1;
    };
    if (wave < 4) {
        double* priv = lds + wave * SOLVER_PRIV;
        double* Sw = priv;
        double* Dpw = priv + 64 * SBS;
        double* rvw = Dpw + DPB;
        double* Rw = Xt + 16 * wave * LT;
        for (int cb = 0; cb < 4; ++cb) {
            // column block cb of L_bb: row blocks cb..3, published by waves cb..3 of the critical workgroup (count >= cb + 1)
            if (!gwait_lanes(g.flags + CH_F_CB + b * 8, cb, 3, ep8 + cb + 1, g.flags, g.epoch)) CH_FAIL();
            {
                const int rowi = 16 * cb + lane;                          // lanes along the rows 16 cb .. 63
                double v[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = (rowi < TB) ? ldc(g.A + (size_t)(b * TB + 16 * cb + c) * ld + b * TB + rowi) : 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) Sw[lane * SBS + c] = v[c];
                if (lane < 16) rvw[lane] = ldc(g.rinv_all + b * TB + 16 * cb + lane);
            }
            {
                const int pr = lane & 15;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * i + (lane >> 4);
                    Dpw[pr * DPS + c] = (c < pr) ? Sw[pr * SBS + c] * rvw[pr] : 0.0;
                }
            }
            double x[4];
            {
                const int rr = lane >> 2, q = lane & 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = Rw[rr * LT + 16 * cb + 4 * i + q];
                solve16(x, Dpw, rvw, q, [](auto) {});
#pragma unroll
                for (int i = 0; i < 4; ++i) Rw[rr * LT + 16 * cb + 4 * i + q] = x[i];
            }
            lds_set(rowDoneF + wave, cb + 1);
            publish_solved16(g.A, ld, a * TB + 16 * wave, b * TB + 16 * cb, x);
            for (int c = cb + 1; c < 4; ++c) {
                const double* ap = Rw + li * LT + 16 * cb + lk;
                const double* bp = Sw + (16 * (c - cb) + li) * SBS + lk;
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * k4], bp[4 * k4], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Rw[(lk + 4 * r) * LT + 16 * c + li] -= acc[r];
            }
            if (feeder)
                while (ns < cb) {                                         // earlier slices, if their inputs have arrived
                    const int rc = do_slice(ns, false);
                    if (rc < 0) CH_FAIL();
                    if (rc == 0) break;
                    ++ns;
                }
        }
        drain_stores();
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cntL, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && old == 3) fl_store(g.flags + CH_F_FL + a * CH_TMAX + b, ep8 + 4);
    }
    if (!feeder) return;
    for (; ns < 4; ++ns)
        if (do_slice(ns, true) < 0) CH_FAIL();
    // X' = far - P: tile (a, b+1), urgent
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            stc(g.A + (size_t)((b + 1) * TB + 16 * pC[e] + li) * ld + a * TB + 16 * pR[e] + lk + 4 * r, farP[e][r] - accP[e][r]);
    drain_stores();
    {
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cntX, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && old == 7) fl_store(g.flags + CH_F_SHIP + a, ep8 + 4);
    }
    // far part of (a, a), lower blocks
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (e < nq) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                stc(g.A + (size_t)(a * TB + 16 * qC[e] + li) * ld + a * TB + 16 * qR[e] + lk + 4 * r, farQ[e][r] - accQ[e][r]);
        }
    drain_stores();
    {
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cntD, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && old == 7) fl_store(g.flags + CH_F_SHIP2 + a, ep8 + 4);
    }
#undef CH_FAIL
}

// inverter: W_jj = L_jj^-1 as the diagonal tiles are published (plain stores: the consumers are later launches)
__device__ void chain_inverter(const ChainArgs& g, double* lds, double* tiles, double* rinv, int* okw) {
    const int tid = threadIdx.x;
    double* S = tiles;
    double* Wt = tiles + TB * LT;
    for (int j = 0; j < g.Tn; ++j) {
        if (!wg_wait_L(j, j, g, okw)) return;
        tile_g2s_c(S, g.A, g.ld, j * TB, j * TB, tid);
        if (tid < TB) rinv[tid] = ldc(g.rinv_all + j * TB + tid);
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

__global__ void __launch_bounds__(CH_THREADS) k_chol_chain(ChainArgs g) {
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
        chain_critical(g, lds, tiles, dprep, rinv, sy);
        return;
    }
    int e = bid - 1;
    const int ns = chain_n_solvers(g.Tn), nn = chain_n_near(g.Tn);
    if (e < ns) {
        int a = 2, b = e;
        while (b > a - 2) { b -= a - 1; ++a; }
        if (threadIdx.x >= 256 && a != b + 2) return;      // four waves, except the feeders
        chain_solver(g, a, b, lds, tiles, sy);
        return;
    }
    if (threadIdx.x >= 256) return;                        // the other helpers are four waves
    if (e < ns + nn) {
        e -= ns;
        const int a = 3 + (e >> 1);
        chain_near_owner(g, a, a - 1 + (e & 1), lds, tiles, sy);
    } else {
        chain_inverter(g, lds, tiles, rinv, sy);
    }
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
    const int nsig = (Sacc && j >= 2) ? (j - 1) * j / 2 : 0;
    if (e < nfin) winv_row_tile(A, Winv, ld, j - 1, e >> 1, e & 1, lds, 2);
    else if (e < nfin + npre) { e -= nfin; winv_row_tile(A, Winv, ld, j, e >> 1, e & 1, lds, 1); }
    else if (e < nfin + npre + nsig) {
        int I, J;
        tile_from_index(e - nfin - npre, I, J);
        sigma_row_tile(Winv, Sacc, ld, j - 2, I, J, lds, tiles);
    } else {
        tvec_role(A, Winv, tv_xi, tv_t, ld, j - 1, lds);
    }
}

}  // namespace sgp
