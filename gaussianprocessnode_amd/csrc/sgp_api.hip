// sgp_api.hip -- C ABI (include/sgp_hip.h) over the gfx950 kernels of sgp_kernels.hip.h.
//
// The handle owns every device buffer of one rank's shard; a sweep is two launch sequences
// (sgp_sweep_local, sgp_sweep_finish) with the packed statistics buffer as the only hand-off, so that a
// multi-GPU caller can sum-all-reduce that buffer in between (RCCL through torch.distributed).  Launches
// are eager by default (~45 kernels per sweep, enqueued well ahead of the GPU); SGP_FLAG_GRAPH captures each
// sequence once into a hipGraph and replays it -- bitwise the same results, measured ~20 us per sweep slower.
// Per-sweep scalars travel through a pinned Params block that each sequence's first kernel mirrors on the device.
#include "../../include/sgp_hip.h"
#include "sgp_kernels.hip.h"
#ifdef SGP_WITH_PERSISTENT_CHAIN          // the round-2 experiment (one persistent launch per factorisation): correct, slower,
#include "sgp_chain.hip.h"                // and therefore only in the variant library (`_build.build(variant="chain")`)
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

using namespace sgp;

namespace {

thread_local std::string g_create_error;
std::atomic<int> g_live_handles{0};       // handles alive in this process (two sweeping at once cannot promise each other CUs)

// Every live handle, so that none outlives the process' orderly exit: a handle that is still alive when exit() runs the static
// destructors -- streams (one of them CU-masked), events, pinned blocks, possibly an all-reduce hook whose trampoline the host
// language has already freed -- was seen to end a rocprofv3-profiled run in SIGSEGV inside __cxa_finalize (round 3:
// tools/hooked_train.py).  The first sgp_create registers an atexit handler; it is registered AFTER the HIP runtime initialised
// (sgp_create has just talked to it), so it runs BEFORE the runtime's own teardown and destroys what the caller left behind.
// sgp_destroy on a handle that is no longer in the registry is a no-op (a finaliser that runs later still).
std::mutex g_registry_mutex;
std::vector<sgp_handle*>& registry() { static std::vector<sgp_handle*>* r = new std::vector<sgp_handle*>(); return *r; }   // (never destructed)
bool g_atexit_registered = false;
void destroy_live_handles_at_exit() {
    std::vector<sgp_handle*> live;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        live = registry();
    }
    for (sgp_handle* h : live) sgp_destroy(h);
}
void register_handle(sgp_handle* h) {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    registry().push_back(h);
    if (!g_atexit_registered) {
        g_atexit_registered = true;
        atexit(destroy_live_handles_at_exit);
    }
}
bool unregister_handle(sgp_handle* h) {          // false: not (or no longer) a live handle
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    auto& r = registry();
    auto it = std::find(r.begin(), r.end(), h);
    if (it == r.end()) return false;
    r.erase(it);
    return true;
}

#ifdef SGP_WITH_PERSISTENT_CHAIN
// The persistent factorisation launches need ALL their workgroups resident at once (they wait for each other).  Two handles
// sweeping at the same time could each hold part of the chip: a handle that starts a sweep while another one's may still
// be running waits for the device first.
sgp_handle* g_chain_owner = nullptr;
#endif

struct Graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int64_t key_n = -1;
    int key_prior = -1;
    int key_omega = -1;
    void* key_stats = nullptr;
    bool valid = false;
    void reset() {
        if (exec) hipGraphExecDestroy(exec);
        if (graph) hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr; valid = false;
    }
};

}  // namespace

// words of sgp_handle::dJoin
enum { WORD_JOIN = 0,      // the K_uu chain of the sweep has finished (k_join_set)          -> Sigma launch (k_gemm32)
       WORD_DONE = 1,      // the sweep's last reader of the K_uu chain's outputs is through   -> next sweep's first kernels
       WORD_GATE = 2,      // the streaming SYRK's resident round is on the CUs               -> K_uu chain's first kernel
       WORD_GRAD = 3,      // the K_uu half of the theta gradient is complete                -> k_theta_grad_finish
       WORD_ASM0 = 4,      // group 0's assembly of overlapped sweep number (value) has started     -> the masked statistics stream
       WORD_GROUP0 = 8,    // + g: statistics group g of overlapped sweep number (value) is assembled -> Lambda chain, statM
       WORD_COUNT = 8 + LAM_MAX_GROUPS };
constexpr int RESERVED_CUS_PER_SE = 2;      // of 8: the masked statistics stream runs on 6 CUs per shader engine (192 of 256)

// one group of tile rows of Psi2 = tile columns [c0, c1) of P Lambda P: its SYRK launch, slab area and assemble launch
struct StatGroup {
    int c0, c1;            // tile columns of P Lambda P (index-reversed), formed by the Lambda chain's step `form_step`
    int row_lo, nrows;     // the same as tile rows of Psi2: [T - c1, T - c0)
    int ntiles;            // their lower tiles
    SyrkGeom geom;         // split of the point axis (one resident round of workgroups on the group's CUs)
    int form_step;
    bool masked;           // runs on statM
    size_t slab_off;       // doubles into dSlabs
};

struct sgp_handle {
    sgp_config cfg{};
    int M = 0, Mp = 0, D = 0, dout = 1, Q = 0, Qp = 0, T = 0, TQ = 0;
    int64_t n = 0, n_max = 0;
    double n_nodes = 0;
    bool has_omega = false, has_yv = false, have_data = false, have_kernel = false, have_inducing = false;
    int prior_form = 2;            // 1 dense precision, 2 isotropic
    bool swept_local = false, swept = false, stats_dirty = false;
    bool in_flight = false;        // a sweep may still be executing (its streams are non-blocking)
    bool sync_reported = false;    // a getter has reported dInfo[3] (check_sync_status): the next sweep clears it
    hipStream_t last_stream = nullptr;   // the stream the last sweep's tail was enqueued on (sgp_sweep_finish): what stream order covers
    uint64_t data_gen = 0, swept_data_gen = ~0ull;   // bumped by set_data / set_inducing; recorded by the sweep
    Params swept_params{};                            // kernel / noise parameters the last sweep ran with
    int n_ell = 1;
    // device buffers
    double *dXu = nullptr, *dXus = nullptr, *dX = nullptr, *dYw = nullptr, *dY = nullptr, *dYv = nullptr, *dOmega = nullptr;
    double *dKuf = nullptr, *dBpart = nullptr, *dSlabs = nullptr, *dStatsOwn = nullptr, *dStats = nullptr, *dDataScal = nullptr;
    double *dKuu = nullptr, *dWk = nullptr, *dKinv = nullptr;
    double *dLam = nullptr, *dWl = nullptr, *dSigma = nullptr, *dR = nullptr, *dXi = nullptr, *dMu = nullptr;
    double *dLambda0 = nullptr, *dXi0 = nullptr, *dOut = nullptr, *dWishart = nullptr, *dTrace = nullptr, *dTmp = nullptr;
    double *dPa = nullptr, *dPb = nullptr, *dKmu = nullptr, *dUvT = nullptr, *dScratch = nullptr, *dOut2 = nullptr, *dUvWork = nullptr;
    double *dGradM = nullptr, *dGradPart = nullptr, *dGrad = nullptr;   // theta-gradient scratch (allocated on first use)
    double* dSaccK = nullptr;      // K_uu chain: Sigma-style accumulator of K_uu^-1 = W_K^T W_K (see sigma_row_tile)
    bool use_chain = false;        // (always false without SGP_WITH_PERSISTENT_CHAIN)
    long long gate_epoch = 0;      // value the sweep's SYRK stores into the gate word
#ifdef SGP_WITH_PERSISTENT_CHAIN
    // persistent factorisation launch (sgp_chain.hip.h), one set per chain: [0] K_uu, [1] Lambda
    long long* dChainFlags[2] = {nullptr, nullptr};
    long long chain_epoch[2] = {0, 0};
    bool gate_kuu = false;         // the K_uu chain of the sweep being enqueued waits behind the gate (see k_chain_gate)
    double* dKuuAlt = nullptr;     // the other parity of dKuu / dLam: the launches alternate (see sgp_chain.hip.h, hand-offs)
    double* dLamAlt = nullptr;
    double* dChainFar[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};     // [chain][parity] mailbox matrices
    double* dChainShip[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    double* dChainRinv[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    long long* dChainTrace[2] = {nullptr, nullptr};   // diagnostics, allocated when SGP_CHAIN_TRACE is set
    ChainArgs* dChainArgs[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [chain][parity]: the launches' argument structs
    ChainArgs chain_args_shadow[2][2];                // what the device copies hold (rewritten only when something changes)
    bool chain_args_valid[2][2] = {{false, false}, {false, false}};
#endif
    // environment switches (diagnostics / A-B), read once in sgp_create
    bool env_no_gate = false, env_join_event = false, env_grad_one_stream = false;
    int spin_limit = JOIN_SPIN_LIMIT;   // polls before a bounded device-word wait gives up (SGP_SPIN_LIMIT: tests shorten it)
    double* dCall = nullptr;       // scratch of the per-call outputs (sgp_predict, sgp_w_stats): grows, never shrinks
    size_t call_capacity = 0;
    int* dInfo = nullptr;
    int64_t* dStamps = nullptr;
    int64_t* dStampTotals = nullptr;
    Params* hParams = nullptr;     // pinned
    double* hMirror = nullptr;     // pinned: the last sweep's scalars + hand-off status word + epoch, written by k_scalars itself
    long long mirror_epoch = -1;   // done_epoch of the sweep whose k_scalars was given hMirror; -1: something enqueued since may have changed the status word
    double* hStage = nullptr;      // pinned staging of sgp_set_data (minibatches: one synchronisation instead of six blocking copies)
    size_t stage_doubles = 0;
    uint64_t params_gen = 1;       // bumped by every setter that changes hParams or Xu
    uint64_t main_prep_gen = 0;    // the generation k_prep_xu last mirrored onto the main stream's copies (dXus, dParams)
    Params* dParams = nullptr;
    Params* dParamsK = nullptr;    // the K_uu chain's own copy (it runs on the side stream)
    long long* dJoin = nullptr;    // device-side join word of the two streams (see UvArgs::join)
    long long join_epoch = 0;
    bool dev_words = false;        // this sweep's streams meet through device words (eager launches)
    bool use_events = false;       // ... and (captured graphs, sweeps with an all-reduce hook) through events
    bool overlap_now = false;      // the sweep being enqueued is an overlapped one (sweep_overlapped)
    bool join_by_flag = false;     // this sweep's F2 waits on dJoin inside k_gemm32 instead of on evSide
    bool gate_side = false;        // the K_uu chain waits for the SYRK's resident round (dJoin[2]); the SYRK grid then uses all CUs
    long long grad_epoch = 0;      // dJoin[3]: the K_uu half of the theta gradient is complete (enqueue_theta_grad)
    long long done_epoch = 0;      // dJoin[1]: the last value a sweep's final kernel was told to write (see k_scalars)
    const Params* params_src = nullptr;   // what k_prep_xu mirrors: hParams, or dTrainParams while a device-paced run is open
    // device-paced training (sgp_train_*): the resident training set, the optimiser state and the parameter source
    double *dTrainX = nullptr, *dTrainY = nullptr;
    int64_t train_N = 0;
    TrainState* dTrain = nullptr;
    Params* dTrainParams = nullptr;
    bool training = false;
    bool train_probit = false;     // the open run is a classification run (sgp_train_likelihood)
    double* dXusK = nullptr;
    hipStream_t own = nullptr, side = nullptr;
    hipEvent_t evSide = nullptr, evDone = nullptr;
    hipEvent_t evGroup[LAM_MAX_GROUPS] = {nullptr};   // data-sharded overlapped sweep: group g's reduced statistics are in place (statM -> own)
    // Overlapped sweep (plan_overlap): the statistics are produced in groups of tile rows -- the first on the sweep's own stream,
    // the others on statM, a CU-masked queue that leaves RESERVED_CUS_PER_SE compute units per shader engine to the two
    // factorisation chains -- while the Lambda chain already factors the tile columns it has.
    hipStream_t statM = nullptr;
    int stat_cus_masked = 0;       // CUs statM may use (0: no masked stream -- the overlapped sweep is off)
    bool overlap = false;          // the resident data / sizes qualify (set_point_count)
    int ngroups = 0;
    StatGroup grp[LAM_MAX_GROUPS];
    long long stat_epoch = 0;      // number of the last overlapped sweep: what its groups' words dJoin[WORD_GROUP0 + g] receive
    int env_overlap = -1;          // SGP_OVERLAP: 0 off, 1 on wherever it is possible; default: where the planner's model says it pays
    int env_g1_mode = 2;           // SGP_G1_AFTER (see enqueue_stats_overlapped)
    int env_syrk_wt = 0;           // SGP_SYRK_WT (see plan_overlap)
    bool env_syrk_wide = true;     // SGP_SYRK_WIDE=0: the 256-thread SYRK everywhere (A/B switch)
    bool defer_request = false, kuu_deferred = false;   // the K_uu chain's steps enqueued alternately with the Lambda chain's (sgp_sweep, see enqueue_finish1)
    bool env_interleave = true;     // the K_uu chain's and the Lambda chain's launches enqueued alternately (SGP_INTERLEAVE=0: chain after chain).  A sweep
                                    // that starts on an idle device -- the first of a block, every sweep of a caller that fetches something in between --
                                    // otherwise has its Lambda chain wait for the host to get through the other chain's 14 launches; once the host is a
                                    // sweep ahead the order makes no difference (profiles/r04_ab_log.txt [30], [37], [38])
    bool env_no_zero_copy = false; // SGP_NO_ZERO_COPY=1: sgp_w_stats copies its results back instead of writing them to pinned memory (A/B switch)
    int64_t gate_min = 10000;      // points x lower tiles from which the SYRK is taken to fill the chip (SGP_GATE_MIN: A/B switch; see set_point_count)
    bool syrk_wide = false;        // the resident problem's SYRK launches are k_syrk_direct (set_point_count)
    std::vector<int> env_overlap_cols;   // SGP_OVERLAP_COLS: group boundaries (tile columns of P Lambda P), e.g. "3" or "2,4"
    int nblk = 0, ntiles = 0, num_cus = 256;
    SyrkGeom geom{};               // the plain sweep's single SYRK launch over all tile rows (set_point_count)
    int64_t stats_count = 0;
    size_t slab_capacity = 0;
    Graph gLocal, gFinish, gFinish2, gKuu;
    double* dBred = nullptr;
    double* dPack = nullptr;       // exchange buffer of data-sharded sweeps: [lower tiles | B | scalars] (allocated with the hook)
    int64_t pack_count = 0;
    bool pack_now = false;         // the statistics being enqueued go to dPack (exchange_stats follows)
    sgp_allreduce_fn allreduce = nullptr;   // the multi-GPU exchange step of sgp_sweep (see include/sgp_hip.h)
    void* allreduce_ctx = nullptr;
    void* rccl_comm = nullptr;
    double ryy_data[MAXO * MAXO] = {0};   // sum omega y y' of the current data (without the output-covariance term)
    std::string err;
};

#define HIPCHK(h, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            char buf_[512];                                                                               \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (h) (h)->err = buf_; else g_create_error = buf_;                                           \
            return SGP_ERR_HIP;                                                                           \
        }                                                                                                 \
    } while (0)

static int fail(sgp_handle* h, int code, const char* msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// the side stream (K_uu chain) yields to whatever stream carries the critical path
static hipError_t create_low_priority_stream(hipStream_t* s) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
}

// Setters change state that an enqueued sweep reads WHEN IT EXECUTES (the pinned parameter block, the data buffers, the
// prior): they first wait for any sweep still in flight.  The library's streams are non-blocking, so the implicit
// synchronisation of hipMemcpy with the legacy default stream does not cover them.
// per-call device scratch: a training loop calls sgp_predict / sgp_w_stats every minibatch, hipMalloc + hipFree per call
// was ~0.1 ms each
static int call_scratch(sgp_handle* h, size_t count, double** out) {
    if (count > h->call_capacity) {
        if (h->dCall) hipFree(h->dCall);
        h->dCall = nullptr;
        h->call_capacity = 0;
        const size_t want = count + count / 2;
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dCall), sizeof(double) * want));
        h->call_capacity = want;
    }
    *out = h->dCall;
    return 0;
}

// Host waits of the hot loop's getters.  hipStreamSynchronize / hipDeviceSynchronize may put the calling thread to sleep until an
// interrupt arrives (the runtime's choice of wait mode): tens of microseconds of wake-up latency behind a 230 us sweep, paid by
// every "sweep, fetch something, next sweep" iteration of a host-paced caller (measured with tools/wstats_time.py: a sweep +
// sgp_get_scalars took 300-350 us of wall time for 223 us of device time).  The waits below first POLL hipStreamQuery for up to ~2 ms
// -- several sweeps' worth -- and only then fall back to the blocking call.  SGP_SPIN_WAIT=0 restores the plain blocking waits.
static bool g_spin_wait = [] { const char* e = getenv("SGP_SPIN_WAIT"); return !(e && atoi(e) == 0); }();
static hipError_t wait_stream(hipStream_t s) {
    if (g_spin_wait) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; ++it) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) return e;
            if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    return hipStreamSynchronize(s);
}
// the library's own streams first, polled; the device-wide call behind them returns at once unless foreign work is still queued
static hipError_t drain_device(sgp_handle* h) {
    for (hipStream_t s : {h->own, h->side, h->statM})
        if (s) { const hipError_t e = wait_stream(s); if (e != hipSuccess) return e; }
    return hipDeviceSynchronize();
}

static int quiesce(sgp_handle* h) {
    if (h->training) return fail(h, SGP_ERR_ARG, "a device-paced training run is open on this handle: call sgp_train_end first");
    if (h->in_flight) {
        HIPCHK(h, hipSetDevice(h->cfg.device));
        HIPCHK(h, drain_device(h));
        h->in_flight = false;
    }
    return 0;
}

// streaming-SYRK grid: tiles x point-chunks, sized to ONE resident round: the kernel's 40 KB of LDS let 4 workgroups
// share a CU, i.e. 1024 slots on 256 CUs.  A grid just above that (the first version: 1152) leaves a straggler round in
// which 128 workgroups run one to a CU at a fraction of the matrix-core rate -- PMC: the CUs were busy 71 % of the launch;
// the largest chunk count with tiles x chunks <= slots (1008 at M = 512) keeps every CU at 4 workgroups from start to end.
// Chunk counts are rounded down so that tiles x chunks is a multiple of the 8 XCDs (k_syrk_stream's block map) when that
// costs at most one eighth of the chunks.
// In a plain sweep the K_uu chain runs beside this kernel and each workgroup of its Cholesky steps takes the LDS of a whole
// CU: SYRK_RESERVED_CUS are left out of the slot count for it unless the chain is gated behind this launch (set_point_count).
// Measured at T with the round-1 grid (sweeps/s, SYRK us): 40 reserved 3312 / 66.5, 24 the same, 8: 3355 / 62.6, 0: 3202 /
// 76.7 (a second round).
#ifndef SYRK_BLOCKS_PER_CU
#define SYRK_BLOCKS_PER_CU 4
#endif
#ifndef SYRK_RESERVED_CUS
#define SYRK_RESERVED_CUS 8
#endif
// wide (round 4): the launch is k_syrk_direct -- ONE 512-thread workgroup per CU, whose eight waves take the item's k-steps (4 points)
// round-robin, each with the whole tile in its accumulators, and add their partial tiles up in LDS: one slab per CU.  Any item count
// works there (its block map pads to a multiple of 8); a chunk is a multiple of 4 SYRK_WAVES points (whole rounds of k-steps).
static SyrkGeom syrk_geometry(int row_lo, int nrows, int cus, int64_t n, bool wide = false) {
    SyrkGeom g;
    g.row_lo = row_lo;
    g.nrows = nrows;
    g.tile0 = row_lo * (row_lo + 1) / 2;
    g.ntiles = (row_lo + nrows) * (row_lo + nrows + 1) / 2 - g.tile0;
    g.chunk = wide ? 4 * SYRK_WAVES : KB;
    g.nchunks = 0;
    g.write_through = 0;
    g.wide = wide ? 1 : 0;
    if (n <= 0) return g;
    if (wide) {
        const int want = std::max(1, std::max(8, cus) / g.ntiles);
        int64_t per = (n + want - 1) / want;
        per = std::max<int64_t>(4 * SYRK_WAVES, (per + 4 * SYRK_WAVES - 1) / (4 * SYRK_WAVES) * (4 * SYRK_WAVES));
        g.chunk = (int)per;
        g.nchunks = (int)std::max<int64_t>(1, (n + per - 1) / per);
        return g;
    }
    const int slots = SYRK_BLOCKS_PER_CU * std::max(8, cus);
    int a = 8;
    for (int q = 2; q <= 8; q *= 2)
        if (g.ntiles % q == 0) a = 8 / q;                 // smallest a with (ntiles * a) % 8 == 0
    int want = std::max(1, slots / g.ntiles);
    if (want > a) want = want / a * a;
    int64_t per = (n + want - 1) / want;
    per = std::max<int64_t>(KB, (per + KB - 1) / KB * KB);
    g.chunk = (int)per;
    g.nchunks = (int)std::max<int64_t>(1, (n + per - 1) / per);
    if (g.nchunks > a) g.nchunks = (g.nchunks + a - 1) / a * a;      // (<= want: trailing chunks may be empty, zero slabs)
    return g;
}
static inline size_t syrk_items(const SyrkGeom& g) { return (size_t)g.ntiles * g.nchunks; }
// k_assemble's third grid dimension: 4 (four entries per thread, all chunk loads of a thread in flight at once) while a tile has few
// slabs, 16 (one entry per thread) when the point axis is cut into many chunks
static inline unsigned assemble_z(const SyrkGeom& g) { return g.nchunks <= 24 ? 4u : 16u; }
static void launch_syrk(const SyrkGeom& g, hipStream_t s, const double* Kuf, const double* omega, double* slabs, int Mp, int64_t n,
                        int64_t* stamps, long long* gate, long long gate_value) {
    if (g.wide)
        hipLaunchKernelGGL(k_syrk_direct, dim3((unsigned)((syrk_items(g) + 7) / 8 * 8)), dim3(SYRK_DIRECT_THREADS), 0, s, Kuf, omega, slabs, Mp, n,
                           g, stamps, gate, gate_value);
    else
        hipLaunchKernelGGL(k_syrk_stream, dim3((unsigned)syrk_items(g)), dim3(256), 0, s, Kuf, omega, slabs, Mp, n, g, stamps, gate,
                           gate_value);
}

// ------------------------------------------------------------------------------------------------
// dense building blocks (launch sequences)
// ------------------------------------------------------------------------------------------------
// Winv (may be nullptr): receives W = L^-1.  The diagonal tile of step j is inverted by that step's otherwise idle
// diagonal workgroup; extra workgroups of step j's launch finish block row j - 1 of W and pre-accumulate block row j (see
// winv_row_tile), so that the one short launch after the last step only has two products per tile left for the last row.
// form (may be nullptr): step 0 evaluates the matrix on the fly (Lambda = Lambda0 + W (x) Psi2, see LamForm) instead of
// reading it from A.  Sacc (may be nullptr; needs Winv): collects Sigma = W^T W row by row during the steps
// (sigma_row_tile); pass the same buffer to launch_ata, which then only adds the last block row.
// step_wait (may be nullptr): step_wait[j] != nullptr makes the stream wait for that event in front of step j (data-sharded
// overlapped sweeps: the group of tile columns that step j forms has come back from its all-reduce, see enqueue_stats_overlapped).
// (a sequence object: launch_potrf runs it to the end; the interleaved enqueue of a sweep's two chains -- enqueue_chains_interleaved --
// alternates between two of them)
struct PotrfSeq {
    double* A; int ld, Tn; int* info; int n_valid; double* scratch; hipStream_t s;
    double* Winv; LamForm form; bool has_form; double* Sacc; const double* tv_xi; double* tv_t; const hipEvent_t* step_wait;
    int j = 0;
    PotrfSeq(double* A_, int ld_, int Tn_, int* info_, int n_valid_, double* scratch_, hipStream_t s_, double* Winv_ = nullptr,
             const LamForm* form_ = nullptr, double* Sacc_ = nullptr, const double* tv_xi_ = nullptr, double* tv_t_ = nullptr,
             const hipEvent_t* step_wait_ = nullptr)
        : A(A_), ld(ld_), Tn(Tn_), info(info_), n_valid(n_valid_), scratch(scratch_), s(s_), Winv(Winv_), has_form(form_ != nullptr),
          Sacc(Sacc_), tv_xi(tv_xi_), tv_t(tv_t_), step_wait(step_wait_) {
        if (form_) form = *form_;
        else { memset(&form, 0, sizeof form); }
    }
    // extra workgroups of launch j: (j >= 2) finish block row j - 1 of W, pre-accumulate block row j (not in the last,
    // potrf-free launch j = Tn), with Sacc add block row j - 2's contribution to Sigma = W^T W; (j >= 1, with tv_t) one
    // workgroup computes block j - 1 of the forward solve t = W (P xi)
    int extras(int jj) const {
        if (!Winv) return 0;
        int e = 0;
        if (jj >= 2) {
            e += 2 * (jj - 1) * (jj < Tn ? 2 : 1);
            if (Sacc && jj < Tn) e += (jj - 1) * jj / 2;           // (row Tn - 2 of Sigma: left to the product launch, see k_potrf_step)
        }
        if (tv_t && jj >= 1) e += 1;
        return e;
    }
    bool done() const { return j > Tn; }
    void next() {                                     // launch j: a Cholesky step (j < Tn) or the launch behind the last step (j == Tn)
        LamForm none;
        memset(&none, 0, sizeof none);
        none.trace_chain = has_form ? 1 : 0;
        if (j < Tn) {
            const int nt = Tn - j;
            if (step_wait && step_wait[j]) (void)hipStreamWaitEvent(s, step_wait[j], 0);
            hipLaunchKernelGGL(k_potrf_step, dim3(nt * (nt + 1) / 2 + potrf_twins(Tn, j) + extras(j)), dim3(PSTEP_THREADS), 0, s, A, ld, j, Tn, info,
                               n_valid, scratch, Winv, Sacc, tv_xi, tv_t, has_form ? form : none);   // (every step: a tile column may be formed later than step 0)
        } else if (j == Tn && Winv && extras(Tn) > 0) {
            hipLaunchKernelGGL(k_potrf_step, dim3(extras(Tn)), dim3(PSTEP_THREADS), 0, s, A, ld, Tn, Tn, info, n_valid, scratch, Winv, Sacc,
                               tv_xi, tv_t, none);
        }
        ++j;
    }
};
static void launch_potrf(double* A, int ld, int Tn, int* info, int n_valid, double* scratch, hipStream_t s,
                         double* Winv = nullptr, const LamForm* form = nullptr, double* Sacc = nullptr,
                         const double* tv_xi = nullptr, double* tv_t = nullptr, const hipEvent_t* step_wait = nullptr) {
    PotrfSeq q(A, ld, Tn, info, n_valid, scratch, s, Winv, form, Sacc, tv_xi, tv_t, step_wait);
    while (!q.done()) q.next();
}

static void launch_ata(const double* W, double* C, int ld, int Tn, hipStream_t s, int rev = 0, const double* mu = nullptr,
                       double* R = nullptr, const double* Psi2 = nullptr, const double* Kinv = nullptr,
                       double* trace_part = nullptr, const UvArgs* uv = nullptr, const double* Sacc = nullptr) {
    const int extra = uv ? Tn * Tn : 0;                  // pass 2 of Uv rides in the same launch (uv_cols_role)
    hipLaunchKernelGGL(k_gemm32, dim3(Tn * (Tn + 1) / 2 * 4 + extra), dim3(256), 0, s, W, W, C, ld, Tn, 0, 0, rev, mu, R, Psi2,
                       Kinv, trace_part, uv ? *uv : UvArgs{}, Sacc);
}

#ifdef SGP_WITH_PERSISTENT_CHAIN
// The same factorisation (and ride-along roles) as launch_potrf, with the Cholesky itself as ONE persistent launch
// (sgp_chain.hip.h).  `which`: 0 = K_uu chain (the matrix is evaluated from the scaled inducing inputs), 1 = Lambda chain
// (evaluated from the statistics through `form`).  A receives L; Winv / Sacc / tv_* as in launch_potrf.
static int launch_chain(sgp_handle* h, int which, double* A, double* A_next, int ld, int Tn, int* info, int n_valid, hipStream_t s,
                         double* Winv, const LamForm* form, double* Sacc, const double* tv_xi, double* tv_t, const double* Xus,
                         const Params* P, int M, int D) {
    ChainArgs g;
    memset(&g, 0, sizeof g);
    const int par = (int)(++h->chain_epoch[which] & 1);
    g.A = A; g.A_next = A_next; g.ld = ld; g.Tn = Tn; g.info = info; g.n_valid = n_valid;
    g.Far = h->dChainFar[which][par]; g.Far_next = h->dChainFar[which][par ^ 1];
    g.Ship = h->dChainShip[which][par]; g.Ship_next = h->dChainShip[which][par ^ 1];
    g.rinv_all = h->dChainRinv[which][par]; g.rinv_next = h->dChainRinv[which][par ^ 1];
    g.Winv = Winv;
    g.abortw = h->dChainFlags[which] + CH_F_ABORT + par;
    g.abortw_next = h->dChainFlags[which] + CH_F_ABORT + (par ^ 1);
    g.trace = h->dChainTrace[which];
    if (form) g.form = *form;
    g.Xus = Xus; g.P = P; g.M = M; g.D = D;
    // the device copy of the arguments changes rarely (statistics buffer rebound, prior form): rewritten only then, after
    // waiting for whatever may still be reading it
    if (!h->chain_args_valid[which][par] || memcmp(&g, &h->chain_args_shadow[which][par], sizeof g) != 0) {
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, hipMemcpy(h->dChainArgs[which][par], &g, sizeof g, hipMemcpyHostToDevice));
        h->chain_args_shadow[which][par] = g;
        h->chain_args_valid[which][par] = true;
    }
    hipLaunchKernelGGL(k_chol_chain, dim3(chain_blocks(Tn, Winv != nullptr)), dim3(CH_THREADS), 0, s,
                       (const ChainArgs*)h->dChainArgs[which][par]);
    if (!Winv) return 0;
    for (int j = 1; j <= Tn; ++j) {
        int e = 0;
        if (j >= 2) {
            e += 2 * (j - 1) * (j < Tn ? 2 : 1);
            if (Sacc && j < Tn) e += (j - 1) * j / 2;
        }
        if (tv_t) e += 1;
        if (e > 0) hipLaunchKernelGGL(k_chain_extras, dim3(e), dim3(256), 0, s, A, ld, j, Tn, Winv, Sacc, tv_xi, tv_t);
    }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
extern "C" int sgp_abi_version(void) { return SGP_ABI_VERSION; }

extern "C" const char* sgp_last_error(const sgp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

template <typename T>
static hipError_t dalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
}

extern "C" int sgp_create(const sgp_config* cfg, sgp_handle** out) {
    if (!cfg || !out) return fail(nullptr, SGP_ERR_ARG, "sgp_create: null argument");
    *out = nullptr;
    if (cfg->m < 1 || cfg->d < 1 || cfg->d > MAXD || cfg->d_out < 1 || cfg->d_out > MAXO || cfg->n_max < 1)
        return fail(nullptr, SGP_ERR_ARG, "sgp_create: need m >= 1, 1 <= d <= 32, 1 <= d_out <= 4, n_max >= 1");
    if ((int64_t)cfg->m * cfg->d_out > CU_MAXQ - TB)
        return fail(nullptr, SGP_ERR_ARG, "sgp_create: d_out * m is limited to 4032 in this build");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, SGP_ERR_NODEVICE, "sgp_create: no HIP device visible (the HIP path has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, SGP_ERR_ARG, "sgp_create: bad device ordinal");
    HIPCHK((sgp_handle*)nullptr, hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK((sgp_handle*)nullptr, hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, SGP_ERR_NODEVICE, "sgp_create: device is not gfx950 (MI355X); this library is built for gfx950 only");

    sgp_handle* h = new sgp_handle();
    register_handle(h);
    ++g_live_handles;
    h->cfg = *cfg;
    h->M = cfg->m; h->D = cfg->d; h->dout = cfg->d_out; h->n_max = cfg->n_max;
    h->Mp = round_up(h->M, TB);
    h->Q = h->dout * h->M;
    h->Qp = round_up(h->Q, TB);
    h->T = h->Mp / TB;
    h->TQ = h->Qp / TB;
    h->ntiles = h->T * (h->T + 1) / 2;
    const size_t Mp = h->Mp, Qp = h->Qp, nmax = (size_t)h->n_max;
    h->stats_count = (int64_t)(Mp * Mp + Mp * h->dout + SGP_S_COUNT + (size_t)h->dout * h->dout);
    // worst-case slab count
    {
        hipDeviceProp_t prop;
        h->num_cus = (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
                         ? prop.multiProcessorCount : 256;
    }
    // one slab per workgroup of the SYRK's single round -- and never fewer than one per lower tile (M > 2816: more tiles than slots,
    // the point axis is then not split at all, syrk_geometry)
    h->slab_capacity = (size_t)std::max(SYRK_BLOCKS_PER_CU * std::max(8, h->num_cus), h->ntiles) * TB * TB;
    const size_t nblk_max = (nmax + TB - 1) / TB;

#define ALLOC(ptr, count)                                            \
    do {                                                             \
        hipError_t e_ = dalloc(&(ptr), (count));                     \
        if (e_ != hipSuccess) {                                      \
            g_create_error = std::string("hipMalloc failed for " #ptr ": ") + hipGetErrorString(e_); \
            sgp_destroy(h);                                          \
            return SGP_ERR_NOMEM;                                    \
        }                                                            \
    } while (0)
    ALLOC(h->dXu, (size_t)h->M * h->D);
    ALLOC(h->dXus, Mp * h->D);
    ALLOC(h->dX, nmax * h->D);
    ALLOC(h->dYw, nmax * h->dout);
    ALLOC(h->dY, nmax * h->dout);
    ALLOC(h->dYv, nmax);
    ALLOC(h->dOmega, nmax);
    ALLOC(h->dKuf, Mp * nmax);
    ALLOC(h->dBpart, nblk_max * h->dout * Mp);
    ALLOC(h->dBred, (size_t)h->T * h->dout * 4 * TB);      // k_assemble: where the four waves of a B block meet
    ALLOC(h->dSlabs, h->slab_capacity);
    ALLOC(h->dStatsOwn, (size_t)h->stats_count);
    ALLOC(h->dJoin, WORD_COUNT);
    ALLOC(h->dDataScal, SGP_S_COUNT + (size_t)h->dout * h->dout);
    ALLOC(h->dKuu, Mp * Mp);
    ALLOC(h->dWk, Mp * Mp);
    ALLOC(h->dKinv, Mp * Mp);
    ALLOC(h->dSaccK, Mp * Mp);
    ALLOC(h->dLam, Qp * Qp);
    ALLOC(h->dWl, Qp * Qp);
    ALLOC(h->dSigma, Qp * Qp);
    ALLOC(h->dR, Qp * Qp);
    ALLOC(h->dTmp, Qp * Qp);
    ALLOC(h->dUvT, Qp * Qp);
    ALLOC(h->dScratch, 3 * POTRF_SCRATCH);   // per factorisation chain: L_jj parking tile, two tiles of the next diagonal update, the twins' words
    hipMemset(h->dScratch, 0, sizeof(double) * 3 * POTRF_SCRATCH);   // (the unwritten upper tiles of those are read, never used)
    ALLOC(h->dOut2, SGP_R_COUNT);
    ALLOC(h->dUvWork, (2 + 2 * (size_t)h->TQ) * Qp);
    ALLOC(h->dLambda0, Qp * Qp);
    ALLOC(h->dXi, Qp);
    ALLOC(h->dMu, Qp);
    ALLOC(h->dXi0, Qp);
    ALLOC(h->dOut, SGP_R_COUNT);
    ALLOC(h->dWishart, MAXO * MAXO);
    ALLOC(h->dTrace, (size_t)TRACE_BLOCKS + std::max((size_t)TRACE_BLOCKS * MAXO * MAXO, (size_t)h->TQ * (h->TQ + 1) * 4));
    ALLOC(h->dInfo, 4);
    ALLOC(h->dStamps, STAMP_STRIDE * SGP_T_COUNT);
    ALLOC(h->dStampTotals, SGP_T_COUNT + 1 + 2 * SGP_T_COUNT);      // totals, count, then the last sweep's (begin, end) pairs
    ALLOC(h->dParams, 1);
    ALLOC(h->dParamsK, 1);
    ALLOC(h->dXusK, Mp * h->D);
    {
        // diagnostic switches of the environment, read once (not per sweep)
        h->env_no_gate = getenv("SGP_NO_GATE") != nullptr;
        h->env_join_event = getenv("SGP_JOIN_EVENT") != nullptr;
        h->env_grad_one_stream = getenv("SGP_GRAD_ONE_STREAM") != nullptr;
        if (const char* lim = getenv("SGP_SPIN_LIMIT")) h->spin_limit = std::max(1, atoi(lim));
        if (const char* ov = getenv("SGP_OVERLAP")) h->env_overlap = atoi(ov);
        if (const char* g1 = getenv("SGP_G1_AFTER")) h->env_g1_mode = atoi(g1);
        if (const char* wt = getenv("SGP_SYRK_WT")) h->env_syrk_wt = atoi(wt);
        if (const char* sw = getenv("SGP_SYRK_WIDE")) h->env_syrk_wide = atoi(sw) != 0;
        if (const char* gm = getenv("SGP_GATE_MIN")) h->gate_min = atoll(gm);
        if (const char* zc = getenv("SGP_NO_ZERO_COPY")) h->env_no_zero_copy = atoi(zc) != 0;
        if (const char* ni = getenv("SGP_INTERLEAVE")) h->env_interleave = atoi(ni) != 0;
        if (const char* oc = getenv("SGP_OVERLAP_COLS"))
            for (const char* q = oc; *q;) {
                h->env_overlap_cols.push_back(atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
    }
#ifdef SGP_WITH_PERSISTENT_CHAIN
    {
        // opt-in: one persistent launch per factorisation (sgp_chain.hip.h) instead of one launch per 64-column step
        const char* env = getenv("SGP_CHAIN");
        const bool persistent = (env && strcmp(env, "persistent") == 0) || (cfg->flags & SGP_FLAG_PERSISTENT_CHAIN);
        h->use_chain = persistent && !(cfg->flags & SGP_FLAG_GRAPH) && h->TQ <= CH_TMAX && h->T <= CH_TMAX;
        if (h->use_chain) {
            ALLOC(h->dChainFlags[0], CH_F_COUNT);
            ALLOC(h->dChainFlags[1], CH_F_COUNT);
            ALLOC(h->dKuuAlt, Mp * Mp);
            ALLOC(h->dLamAlt, Qp * Qp);
            for (int par = 0; par < 2; ++par) {
                ALLOC(h->dChainFar[0][par], Mp * Mp);
                ALLOC(h->dChainFar[1][par], Qp * Qp);
                ALLOC(h->dChainShip[0][par], Mp * Mp);
                ALLOC(h->dChainShip[1][par], Qp * Qp);
                ALLOC(h->dChainRinv[0][par], Mp);
                ALLOC(h->dChainRinv[1][par], Qp);
                ALLOC(h->dChainArgs[0][par], 1);
                ALLOC(h->dChainArgs[1][par], 1);
            }
            if (getenv("SGP_CHAIN_TRACE")) {
                ALLOC(h->dChainTrace[0], CH_TMAX * 32);
                ALLOC(h->dChainTrace[1], CH_TMAX * 32);
                hipMemset(h->dChainTrace[0], 0, sizeof(long long) * CH_TMAX * 32);
                hipMemset(h->dChainTrace[1], 0, sizeof(long long) * CH_TMAX * 32);
            }
            hipMemset(h->dChainFlags[0], 0, sizeof(long long) * CH_F_COUNT);
            hipMemset(h->dChainFlags[1], 0, sizeof(long long) * CH_F_COUNT);
            // every mailbox starts out full of sentinels; from then on each launch refills the other parity's
            auto fill = [&](double* p, size_t n) {
                hipLaunchKernelGGL(k_chain_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n);
            };
            fill(h->dKuu, Mp * Mp); fill(h->dKuuAlt, Mp * Mp); fill(h->dLam, Qp * Qp); fill(h->dLamAlt, Qp * Qp);
            for (int par = 0; par < 2; ++par) {
                fill(h->dChainFar[0][par], Mp * Mp); fill(h->dChainFar[1][par], Qp * Qp);
                fill(h->dChainShip[0][par], Mp * Mp); fill(h->dChainShip[1][par], Qp * Qp);
                fill(h->dChainRinv[0][par], Mp); fill(h->dChainRinv[1][par], Qp);
            }
            hipDeviceSynchronize();
        }
    }
#else
    if ((cfg->flags & SGP_FLAG_PERSISTENT_CHAIN) || (getenv("SGP_CHAIN") && strcmp(getenv("SGP_CHAIN"), "persistent") == 0)) {
        g_create_error = "sgp_create: this build does not contain the persistent factorisation launch (build the variant library "
                         "with -DSGP_WITH_PERSISTENT_CHAIN: gaussianprocessnode_amd._build.build(variant=\"chain\"))";
        sgp_destroy(h);
        return SGP_ERR_ARG;
    }
#endif
    if (cfg->flags & SGP_FLAG_KEEP_KUF) {
        // per-point partials of k_quadform_fused: [2 T] rows of each quadratic form, then [4] rows of k_n . mu (one allocation)
        ALLOC(h->dPa, (4 * (size_t)h->T + 4) * nmax);
        h->dPb = h->dPa + 2 * (size_t)h->T * nmax;
        h->dKmu = h->dPb + 2 * (size_t)h->T * nmax;
    }
#undef ALLOC
    h->dStats = h->dStatsOwn;
    if (hipHostMalloc(reinterpret_cast<void**>(&h->hParams), sizeof(Params), hipHostMallocDefault) != hipSuccess) {
        g_create_error = "hipHostMalloc failed for the parameter block";
        sgp_destroy(h);
        return SGP_ERR_NOMEM;
    }
    {   // staging for whole minibatches up to 4 MB; larger data sets go straight from the caller's memory (one-off)
        const size_t per_point = (size_t)h->D + 2 * (size_t)h->dout + 2;
        h->stage_doubles = std::min<size_t>((size_t)h->n_max * per_point, (size_t)1 << 19) + SGP_S_COUNT + (size_t)h->dout * h->dout;
        if (hipHostMalloc(reinterpret_cast<void**>(&h->hStage), sizeof(double) * h->stage_doubles, hipHostMallocDefault) != hipSuccess) {
            h->hStage = nullptr;                            // (not fatal: sgp_set_data then copies from the caller's memory)
            h->stage_doubles = 0;
            (void)hipGetLastError();
        }
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&h->hMirror), sizeof(double) * (SGP_R_COUNT + 2), hipHostMallocDefault) != hipSuccess) {
        h->hMirror = nullptr;                               // (not fatal: the getters copy from the device)
        (void)hipGetLastError();
    } else {
        for (int i = 0; i < SGP_R_COUNT + 2; ++i) h->hMirror[i] = -1.0;
    }
    memset(h->hParams, 0, sizeof(Params));
    h->hParams->sigma2 = 1.0;
    for (int d = 0; d < MAXD; ++d) h->hParams->inv_ell[d] = 1.0;
    h->hParams->W[0] = 1.0;
    h->hParams->prior_iso = 1.0;
    h->params_src = h->hParams;
    if (hipStreamCreateWithFlags(&h->own, hipStreamNonBlocking) != hipSuccess ||
        create_low_priority_stream(&h->side) != hipSuccess ||
        hipEventCreateWithFlags(&h->evSide, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evDone, hipEventDisableTiming) != hipSuccess) {
        g_create_error = "stream/event creation failed";
        sgp_destroy(h);
        return SGP_ERR_HIP;
    }
    for (int g = 0; g < LAM_MAX_GROUPS; ++g)
        if (hipEventCreateWithFlags(&h->evGroup[g], hipEventDisableTiming) != hipSuccess) {
            g_create_error = "event creation failed";
            sgp_destroy(h);
            return SGP_ERR_HIP;
        }
    // The overlapped sweep's statistics streams.  statM is a CU-masked queue (hipExtStreamCreateWithCUMask): bit i of the mask
    // is CU (i / 8 / 4) of shader engine (i / 8) % 4 of XCD i % 8 (measured with tools/cu_mask_probe.hip), so the first
    // 32 k bits are k CUs on every shader engine of every XCD -- a symmetric mask: an uneven one (e.g. 216 bits) leaves some
    // engines with fewer CUs than their equal share of the workgroups and costs a second round.
    int reserved_per_se = RESERVED_CUS_PER_SE;
    if (const char* r = getenv("SGP_RESERVED_PER_SE")) reserved_per_se = std::max(1, atoi(r));     // (A/B switch)
    if (h->env_overlap != 0 && !(cfg->flags & SGP_FLAG_GRAPH) && h->dout == 1 && h->num_cus % 32 == 0 && h->num_cus / 32 > reserved_per_se) {
        const int keep = h->num_cus - 32 * reserved_per_se;
        uint32_t mask[16] = {0};
        for (int i = 0; i < keep && i < 512; ++i) mask[i / 32] |= 1u << (i % 32);
        // (Confining the K_uu chain's stream to the complement, the reserved CUs, was measured and dropped: an unmasked launch
        // -- the Lambda chain's -- is dealt CUs from both ends of each shader engine's list, the reserved ones included, and the
        // two chains then queued for the same few CUs while the rest of the chip idled: 47 instead of 18 us per step.)
        if (hipExtStreamCreateWithCUMask(&h->statM, (uint32_t)((h->num_cus + 31) / 32), mask) == hipSuccess)
            h->stat_cus_masked = keep;
        else {
            (void)hipGetLastError();
            h->statM = nullptr;
        }
    }
    hipMemset(h->dInfo, 0, 4 * sizeof(int));
    hipMemset(h->dJoin, 0, WORD_COUNT * sizeof(long long));
    hipMemset(h->dOut, 0, SGP_R_COUNT * sizeof(double));
    hipMemset(h->dStamps, 0, STAMP_STRIDE * SGP_T_COUNT * sizeof(int64_t));
    hipMemset(h->dStampTotals, 0, (SGP_T_COUNT + 1 + 2 * SGP_T_COUNT) * sizeof(int64_t));
    hipMemset(h->dMu, 0, Qp * sizeof(double));
    hipMemset(h->dXi0, 0, Qp * sizeof(double));
    *out = h;
    return 0;
}

extern "C" int sgp_destroy(sgp_handle* h) {
    if (!h) return 0;
    if (!unregister_handle(h)) return 0;       // destroyed already (by the exit handler, or twice by the caller)
    // the hook's trampoline belongs to the host language: never call it again, whatever is still queued
    h->allreduce = nullptr;
    h->allreduce_ctx = nullptr;
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        // the HIP runtime is already down (a finaliser that ran behind its teardown): its teardown released the device objects,
        // all that is left to free is the host struct
        (void)hipGetLastError();
        --g_live_handles;
        delete h;
        return 0;
    }
#ifdef SGP_WITH_PERSISTENT_CHAIN
    if (g_chain_owner == h) g_chain_owner = nullptr;
#endif
    h->gLocal.reset();
    h->gFinish.reset();
    h->gFinish2.reset();
    h->gKuu.reset();
    void* bufs[] = {h->dXu, h->dXus, h->dX, h->dYw, h->dY, h->dYv, h->dOmega, h->dKuf, h->dBpart, h->dSlabs, h->dStatsOwn,
                    h->dDataScal, h->dKuu, h->dWk, h->dKinv, h->dLam, h->dWl, h->dSigma, h->dR, h->dTmp, h->dLambda0,
                    h->dXi, h->dMu, h->dXi0, h->dOut, h->dWishart, h->dTrace, h->dInfo, h->dStamps, h->dParams, h->dPa,
                    h->dUvT, h->dScratch, h->dParamsK, h->dXusK, h->dOut2, h->dStampTotals, h->dUvWork,
                    h->dGradM, h->dGradPart, h->dGrad, h->dCall, h->dSaccK,
                    h->dTrainX, h->dTrainY, h->dTrain, h->dTrainParams, h->dJoin, h->dPack, h->dBred};
    for (void* b : bufs) if (b) hipFree(b);
#ifdef SGP_WITH_PERSISTENT_CHAIN
    void* cbufs[] = {h->dChainFlags[0], h->dChainFlags[1],
                     h->dChainFar[0][0], h->dChainFar[0][1], h->dChainFar[1][0], h->dChainFar[1][1],
                     h->dChainShip[0][0], h->dChainShip[0][1], h->dChainShip[1][0], h->dChainShip[1][1],
                     h->dChainRinv[0][0], h->dChainRinv[0][1], h->dChainRinv[1][0], h->dChainRinv[1][1],
                     h->dChainTrace[0], h->dChainTrace[1], h->dKuuAlt, h->dLamAlt,
                     h->dChainArgs[0][0], h->dChainArgs[0][1], h->dChainArgs[1][0], h->dChainArgs[1][1]};
    for (void* b : cbufs) if (b) hipFree(b);
#endif
    if (h->hParams) hipHostFree(h->hParams);
    if (h->hStage) hipHostFree(h->hStage);
    if (h->hMirror) hipHostFree(h->hMirror);
    if (h->evSide) hipEventDestroy(h->evSide);
    if (h->evDone) hipEventDestroy(h->evDone);
    for (hipEvent_t e : h->evGroup) if (e) hipEventDestroy(e);
    if (h->own) hipStreamDestroy(h->own);
    if (h->side) hipStreamDestroy(h->side);
    if (h->statM) hipStreamDestroy(h->statM);
    --g_live_handles;
    delete h;
    return 0;
}

extern "C" int sgp_set_inducing(sgp_handle* h, const double* Xu) {
    if (!h || !Xu) return fail(h, SGP_ERR_ARG, "sgp_set_inducing: null argument");
    if (int qrc = quiesce(h)) return qrc;
    h->data_gen++;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpy(h->dXu, Xu, sizeof(double) * h->M * h->D, hipMemcpyHostToDevice));
    h->have_inducing = true;
    h->params_gen++;
    h->swept = h->swept_local = false;
    return 0;
}

// The overlapped sweep: does this problem qualify, and how are the tile columns of P Lambda P grouped?
//   * UniSGP, eager launches, the masked statistics stream exists, at least 3 tile rows, a SYRK that fills the chip (the same
//     bound as the K_uu chain's gate).
//   * group 0 (tile columns [0, c0) of P Lambda P = the LAST c0 tile rows of Psi2) runs on all CUs in front of the chain, the
//     other groups on the masked stream while the chain factors; a group must be complete when the chain reaches its first
//     column.  The cuts are chosen by a small model of that schedule (all plans of up to three groups are tried): a SYRK
//     launch lasts as long as its chunks are long (0.18 us per point of a chunk + 5 us, from the launch's real geometry),
//     ~8 / ~11 us of assembly, word kernel and gaps per unmasked / masked group, ~17 us per chain step, 4 us of margin.  Measured at T
//     (N = 10 000, M = 512; sweeps/s on one box): plain order 3750; cuts {3} 3896, {2} 3832-3950, {2,4} 3925, {1,3} 3656,
//     {2,3,5} 3790, {1,2,4} 3528 -- the model ranks them the same way.  When no plan beats the plain order by 5 us the plain
//     order stays (huge N: the masked groups' lost CUs cost more than the chain's early start saves).
static double model_overlap_end(const sgp_handle* h, int64_t n, const int* cuts, int ncuts, double* classic_end) {
    const int T = h->T;
    // one SYRK launch = one resident round: its duration follows the points per chunk (k_syrk_stream: 0.18 us per point at four
    // workgroups per CU, + ~5 us of launch ramp and tail; k_syrk_direct: 0.029 us per point -- eight waves share the chunk, two to a
    // SIMD, 16 MFMAs of 64 cycles per 4 points each, at the 71 TFLOP/s it attains -- + ~13 us of ramp, first loads, the partial
    // tiles' meeting in LDS and the slab store), whatever the number of tiles -- fewer CUs or an awkward tile count show up as
    // fewer, longer chunks (syrk_geometry)
    auto syrk_us = [&](int row_lo, int nrows, int cus) {
        const SyrkGeom g = syrk_geometry(row_lo, nrows, cus, n, h->syrk_wide);
        return g.wide ? 13.0 + 0.029 * g.chunk : 5.0 + 0.18 * g.chunk;
    };
    // The schedule's constants.  k_syrk_stream launches (small problems never overlap; kept as fitted in round 3): ~8 / ~11 us of
    // assembly, word kernel and gaps behind an unmasked / masked group's SYRK, 4 us of margin in front of a forming step (a step
    // that finds its group's word unset waits and then reads the statistics past the L2, element by element -- far slower than
    // the wait alone), 2.5 us for forming.  k_syrk_direct launches: refitted to six measured plans at T (device sweep, us: {3} 216.3,
    // {2} 216.3, {4} 224.9, {2,5} 213.2, {3,6} 221.5, {2,4} 215.5 -- profiles/r04_ab_log.txt [17]; residuals of this model
    // +-1.3 us): in the kernel's own duration (7.5 + 0.029 us per point; syrk_us above is what HIP events see) group 0's assembly
    // and its two boundaries are 11.5 us, the masked stream's first SYRK starts 8 us behind group 0's, a masked group is usable
    // 5 us behind its SYRK, and a step that forms a group costs 5 us more than one that does not.
    const bool dk = h->syrk_wide;
    const double asm0 = dk ? 11.5 : 8.0, asmm = dk ? 5.0 : 11.0, step = 17.0, margin = dk ? 0.0 : 4.0, forming = dk ? 5.0 : 2.5;
    const double mstart = dk ? 8.0 : 2.0, launch = dk ? 5.5 : 0.0;   // (launch: what syrk_us counts beyond the kernel's own duration)
    if (classic_end) *classic_end = syrk_us(0, T, h->num_cus) - launch + asm0 + step * T;
    double ready[LAM_MAX_COLS];
    double t = syrk_us(T - cuts[0], cuts[0], h->num_cus) - launch + asm0;          // group 0 assembled
    for (int c = 0; c < cuts[0]; ++c) ready[c] = t;
    t -= asm0 - mstart;                                       // the masked stream starts when group 0's assembly does
    for (int g = 0; g < ncuts; ++g) {
        const int c0 = cuts[g], c1 = (g + 1 < ncuts) ? cuts[g + 1] : T;
        t += syrk_us(T - c1, c1 - c0, h->stat_cus_masked) - launch + asmm;
        for (int c = c0; c < c1; ++c) ready[c] = t;
    }
    double end = ready[0];
    for (int j = 0; j < T; ++j) {
        bool forms = false;
        for (int g = 0; g < ncuts; ++g) forms = forms || cuts[g] == j;
        end = std::max(end, ready[j] + (j >= cuts[0] ? margin : 0.0)) + step + (forms ? forming : 0.0);
    }
    return end;
}

static void plan_overlap(sgp_handle* h, int64_t n) {
    h->overlap = false;
    h->ngroups = 0;
    const int T = h->T;
    if (!h->statM || h->env_overlap == 0 || h->dout != 1 || T < 3 || T > LAM_MAX_COLS || h->use_chain || h->training ||
        (h->cfg.flags & SGP_FLAG_GRAPH) || n < 1 || (!h->gate_side && h->env_overlap != 1))
        return;
    std::vector<int> cuts;                                   // group boundaries, ascending, in (0, T)
    for (int c : h->env_overlap_cols) if (c > 0 && c < T && (cuts.empty() || c > cuts.back())) cuts.push_back(c);
    if (cuts.empty()) {
        double classic = 0.0, best = 1e300;
        int cand[2];
        // One or two cuts.  (Round 3 settled on ONE: with its 16 us steps the chain reached a third group's columns before that group
        // was there.  With k_syrk_direct a masked group's launch is a quarter shorter and {2,5} beats {3} at T, 213.2 against 216.3 us.)
        // Among plans the model cannot tell apart (< 0.3 us) the later second cut wins: the measured order at T.
        for (int a = 1; a < T; ++a) {
            cand[0] = a; cand[1] = a;
            const double e = model_overlap_end(h, n, cand, 1, &classic);
            if (e < best - 0.3) { best = e; cuts.assign(cand, cand + 1); }
            if (!h->syrk_wide || h->allreduce) continue;     // (data-sharded: every further group is another collective)
            // (three groups were fitted and validated at eight tile columns; with four -- M = 256 -- the planner's {1,2} lost 8 % against the
            // plain order at N = 20 000 where {1} gains: fewer than six columns keep one cut)
            if (T < 6) continue;
            for (int b = T - 1; b > a; --b) {
                cand[1] = b;
                // Three groups only while the masked launches are short (<= 45 us by the model: T has 38 and 22).  At N = 40 000 the
                // model liked {3,5} (masked launches of 63 and 45 us) and the sweep lost 5 % against {4}: a chain step that finds its
                // group late polls and then reads the statistics past the L2, the masked SYRK beside it slows down (0.038 instead of
                // 0.029 us per point in the timeline), and the next group is later still -- a model that knows no feedback must stay
                // out of that regime (profiles/r04_ab_log.txt [17]).
                const double m1 = 13.0 + 0.029 * syrk_geometry(T - b, b - a, h->stat_cus_masked, n, true).chunk;
                const double m2 = 13.0 + 0.029 * syrk_geometry(0, T - b, h->stat_cus_masked, n, true).chunk;
                if (m1 > 45.0 || m2 > 45.0) continue;
                const double e2 = model_overlap_end(h, n, cand, 2, &classic);
                if (e2 < best - 0.3) { best = e2; cuts.assign(cand, cand + 2); }
            }
        }
        // (the plain order stays unless the model sees a gain: 5 us with the LDS-staged SYRK's constants; 1 us with k_syrk_direct's --
        // at C2 (M = 256, four tile columns) the model sees 1.1 us for the cut {1} and the sweep gains 4.5, 129.2 -> 124.8 us)
        if (h->env_overlap != 1 && best > classic - (h->syrk_wide ? 1.0 : 5.0)) return;
    }
    if ((int)cuts.size() + 1 > LAM_MAX_GROUPS) return;
    size_t off = 0;
    int c0 = 0;
    for (size_t g = 0; g <= cuts.size(); ++g) {
        StatGroup& G = h->grp[g];
        G.c0 = c0;
        G.c1 = (g < cuts.size()) ? cuts[g] : T;
        G.row_lo = T - G.c1;
        G.nrows = G.c1 - G.c0;
        G.masked = g > 0;
        G.form_step = G.c0;
        // (the 16-wave kernel for the masked groups as well: with the 256-thread kernel there the sweep was 1 % slower, profiles/r04_ab_log.txt [9])
        G.geom = syrk_geometry(G.row_lo, G.nrows, G.masked ? h->stat_cus_masked : h->num_cus, n, h->syrk_wide);
        // SGP_SYRK_WT (A/B, see k_syrk_stream): 0 plain slab stores (default), 1 write-through in the masked groups, 2 in all
        G.geom.write_through = (h->env_syrk_wt == 2 || (h->env_syrk_wt == 1 && G.masked)) ? 1 : 0;
        G.ntiles = G.geom.ntiles;
        G.slab_off = off;
        off += syrk_items(G.geom) * TB * TB;
        c0 = G.c1;
    }
    if (off > h->slab_capacity) {
        // (only from the blocking setters: nothing is in flight)
        double* bigger = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&bigger), sizeof(double) * off) != hipSuccess) { (void)hipGetLastError(); return; }
        hipFree(h->dSlabs);
        h->dSlabs = bigger;
        h->slab_capacity = off;
    }
    h->ngroups = (int)cuts.size() + 1;
    h->overlap = true;
}

// the launch geometry of the data-sized kernels for n points
static int set_point_count(sgp_handle* h, int64_t n) {
    h->n = n;
    h->nblk = (int)((n + TB - 1) / TB);
    // split the point axis into one resident round of workgroups (see syrk_chunking), chunk a multiple of the stage size
    // Eager launches with a SYRK big enough to fill the chip: the K_uu chain is held back until this launch is resident
    // (enqueue_kuu), so nothing is reserved for it and the grid is sized for all CUs (T: 25 -> 28 chunks, 62.7 -> 57.4 us; the
    // chain then runs after the SYRK and still ends ~9 us before its join).  Small problems keep the early chain: there the
    // two chains are the sweep, and a late K_uu chain is waited for (C1: -15 %, C5: -2 % with the gate).
    // The threshold (points x lower tiles) was 200 000 through the first half of round 4 -- what the LDS-staged SYRK needed to fill
    // the chip; k_syrk_direct has one workgroup per CU and splits the point axis down to 32 points, so far smaller problems gain
    // from it, from the gate and, from ~50 000 on, from the overlapped order (sweeps/s at the old / new threshold, one box,
    // tools/gate_sweep.py, profiles/r04_ab_log.txt [22]): N = 5 000, M = 512: 4 534 -> 4 970; C2 (N = 10 000, M = 256): 7 333 -> 7 640;
    // 3 000 x 512: 4 819 -> 5 057; 10 000 x 128: 10 523 -> 11 836; C4 (4 000 x 128): 11 205 -> 13 032; C5 (1 500 x 48, one tile:
    // 1 500) loses 2 % if gated and C1 15 %: the threshold is 10 000.
    h->gate_side = n * (int64_t)h->ntiles >= h->gate_min && h->dJoin && !(h->cfg.flags & SGP_FLAG_GRAPH) && !h->use_chain &&
                   !h->env_no_gate;
    // (the in-CU split needs the whole LDS of a CU: only where the K_uu chain is gated behind this launch, i.e. the SYRK fills the chip)
    h->syrk_wide = h->gate_side && h->env_syrk_wide;
    h->geom = syrk_geometry(0, h->T, h->num_cus - (h->gate_side ? 0 : SYRK_RESERVED_CUS), n, h->syrk_wide);
    if (syrk_items(h->geom) * TB * TB > h->slab_capacity)
        return fail(h, SGP_ERR_ARG, "sgp_set_data: internal slab capacity exceeded");
    plan_overlap(h, n);
    return 0;
}

extern "C" int sgp_set_data(sgp_handle* h, const double* X, const double* y_mean, const double* y_var,
                            const double* pt_weight, int64_t n, double n_nodes) {
    if (!h || !X || !y_mean) return fail(h, SGP_ERR_ARG, "sgp_set_data: null argument");
    if (int qrc = quiesce(h)) return qrc;
    h->data_gen++;
    if (n < 0 || n > h->n_max) return fail(h, SGP_ERR_ARG, "sgp_set_data: n outside [0, n_max]");
    if (y_var && h->dout != 1) return fail(h, SGP_ERR_ARG, "sgp_set_data: y_var is only defined for d_out = 1");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int dout = h->dout;
    std::vector<double> yw((size_t)std::max<int64_t>(n, 1) * dout);
    std::vector<double> scal(SGP_S_COUNT + (size_t)dout * dout, 0.0);
    double s_w = 0.0;
    for (int64_t i = 0; i < n; ++i) s_w += pt_weight ? pt_weight[i] : 1.0;
    for (int o = 0; o < dout; ++o)
        for (int64_t i = 0; i < n; ++i) yw[(size_t)o * n + i] = y_mean[(size_t)o * n + i] * (pt_weight ? pt_weight[i] : 1.0);
    // Ryy[a][b] = sum_n omega y_a y_b (+ sum omega v for the scalar case)
    for (int a = 0; a < dout; ++a)
        for (int b = 0; b < dout; ++b) {
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += yw[(size_t)a * n + i] * y_mean[(size_t)b * n + i];
            scal[SGP_S_COUNT + a + (size_t)b * dout] = s;
        }
    double s_yy = scal[SGP_S_COUNT];
    if (y_var)
        for (int64_t i = 0; i < n; ++i) s_yy += (pt_weight ? pt_weight[i] : 1.0) * y_var[i];
    scal[SGP_S_YY] = s_yy;
    if (dout == 1) scal[SGP_S_COUNT] = s_yy;
    scal[SGP_S_W] = s_w;
    scal[SGP_S_N] = (n_nodes > 0) ? n_nodes : (double)n;
    for (int i = 0; i < dout * dout; ++i) h->ryy_data[i] = scal[SGP_S_COUNT + i];
    const size_t need = (size_t)n * ((size_t)h->D + 2 * (size_t)dout + 2) + scal.size();
    if (h->hStage && need <= h->stage_doubles) {
        // a minibatch: everything through the pinned staging block, asynchronous copies, ONE synchronisation (the six blocking
        // copies from pageable memory were ~70 us per minibatch of the host-paced streaming loop, a quarter of the sweep behind them)
        double* st = h->hStage;
        struct Piece { double* dst; const double* src; size_t count; };
        const Piece pieces[] = {{h->dX, X, (size_t)n * h->D},
                                {h->dYw, yw.data(), (size_t)n * dout},
                                {h->dY, y_mean, (size_t)n * dout},
                                {h->dYv, y_var, y_var ? (size_t)n : 0},
                                {h->dOmega, pt_weight, pt_weight ? (size_t)n : 0},
                                {h->dDataScal, scal.data(), scal.size()}};
        for (const Piece& pc : pieces) {
            if (!pc.count) continue;
            memcpy(st, pc.src, sizeof(double) * pc.count);
            HIPCHK(h, hipMemcpyAsync(pc.dst, st, sizeof(double) * pc.count, hipMemcpyHostToDevice, h->own));
            st += pc.count;
        }
        HIPCHK(h, hipStreamSynchronize(h->own));
    } else {
        if (n > 0) {
            HIPCHK(h, hipMemcpy(h->dX, X, sizeof(double) * n * h->D, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(h->dYw, yw.data(), sizeof(double) * n * dout, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(h->dY, y_mean, sizeof(double) * n * dout, hipMemcpyHostToDevice));
            if (y_var) HIPCHK(h, hipMemcpy(h->dYv, y_var, sizeof(double) * n, hipMemcpyHostToDevice));
            if (pt_weight) HIPCHK(h, hipMemcpy(h->dOmega, pt_weight, sizeof(double) * n, hipMemcpyHostToDevice));
        }
        HIPCHK(h, hipMemcpy(h->dDataScal, scal.data(), sizeof(double) * scal.size(), hipMemcpyHostToDevice));
    }
    h->n = n;
    h->n_nodes = scal[SGP_S_N];
    h->has_omega = pt_weight != nullptr;
    h->has_yv = y_var != nullptr;
    h->have_data = true;
    if (int rc = set_point_count(h, n)) return rc;
    h->swept = h->swept_local = false;
    return 0;
}

extern "C" int sgp_set_output_cov_sum(sgp_handle* h, const double* S) {
    if (!h || !S) return fail(h, SGP_ERR_ARG, "sgp_set_output_cov_sum: null argument");
    if (int qrc = quiesce(h)) return qrc;
    if (!h->have_data) return fail(h, SGP_ERR_ARG, "sgp_set_output_cov_sum: call sgp_set_data first");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int dd = h->dout * h->dout;
    std::vector<double> ryy(dd);
    HIPCHK(h, hipMemcpy(ryy.data(), h->dDataScal + SGP_S_COUNT, sizeof(double) * dd, hipMemcpyDeviceToHost));
    for (int i = 0; i < dd; ++i) ryy[i] = h->ryy_data[i] + S[i];
    HIPCHK(h, hipMemcpy(h->dDataScal + SGP_S_COUNT, ryy.data(), sizeof(double) * dd, hipMemcpyHostToDevice));
    if (h->dout == 1) {
        double syy = h->ryy_data[0] + S[0];
        HIPCHK(h, hipMemcpy(h->dDataScal + SGP_S_YY, &syy, sizeof(double), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int sgp_set_kernel(sgp_handle* h, double sigma2, const double* ell, int32_t n_ell, double jitter) {
    if (!h || !ell) return fail(h, SGP_ERR_ARG, "sgp_set_kernel: null argument");
    if (int qrc = quiesce(h)) return qrc;
    if (n_ell != 1 && n_ell != h->D) return fail(h, SGP_ERR_ARG, "sgp_set_kernel: n_ell must be 1 or D");
    if (!(sigma2 > 0.0) || !(jitter >= 0.0)) return fail(h, SGP_ERR_ARG, "sgp_set_kernel: sigma2 must be > 0, jitter >= 0");
    for (int d = 0; d < h->D; ++d) {
        double l = ell[n_ell == 1 ? 0 : d];
        if (!(l > 0.0)) return fail(h, SGP_ERR_ARG, "sgp_set_kernel: lengthscales must be > 0");
        h->hParams->inv_ell[d] = 1.0 / l;
    }
    h->hParams->sigma2 = sigma2;
    h->hParams->jitter = jitter;
    h->params_gen++;
    h->n_ell = n_ell;
    h->have_kernel = true;
    return 0;
}

extern "C" int sgp_set_noise(sgp_handle* h, const double* W, double E_log_w) {
    if (!h || !W) return fail(h, SGP_ERR_ARG, "sgp_set_noise: null argument");
    if (int qrc = quiesce(h)) return qrc;
    for (int i = 0; i < h->dout * h->dout; ++i) h->hParams->W[i] = W[i];
    h->hParams->E_logw = E_log_w;
    h->params_gen++;
    return 0;
}

// upload a Q x Q host matrix into a padded Qp x Qp device buffer (pad = identity)
static int upload_padded(sgp_handle* h, const double* src, double* dst) {
    const size_t Q = h->Q, Qp = h->Qp;
    std::vector<double> tmp(Qp * Qp, 0.0);
    for (size_t j = 0; j < Qp; ++j) {
        if (j < Q) memcpy(&tmp[j * Qp], &src[j * Q], Q * sizeof(double));
        else tmp[j * Qp + j] = 1.0;
    }
    HIPCHK(h, hipMemcpy(dst, tmp.data(), Qp * Qp * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int sgp_set_prior(sgp_handle* h, const double* vec, const double* mat, int32_t form) {
    if (!h || !mat) return fail(h, SGP_ERR_ARG, "sgp_set_prior: null argument");
    if (int qrc = quiesce(h)) return qrc;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t Q = h->Q, Qp = h->Qp;
    if (form == 2) {
        if (!(mat[0] > 0.0)) return fail(h, SGP_ERR_ARG, "sgp_set_prior: isotropic variance must be > 0");
        h->hParams->prior_iso = 1.0 / mat[0];
        h->params_gen++;
        h->prior_form = 2;
        return 0;
    }
    if (!vec) return fail(h, SGP_ERR_ARG, "sgp_set_prior: null mean");
    std::vector<double> v(Qp, 0.0);
    memcpy(v.data(), vec, Q * sizeof(double));
    if (form == 1) {
        int rc = upload_padded(h, mat, h->dLambda0);
        if (rc) return rc;
        HIPCHK(h, hipMemcpy(h->dXi0, v.data(), Qp * sizeof(double), hipMemcpyHostToDevice));
        h->prior_form = 1;
        return 0;
    }
    if (form != 0) return fail(h, SGP_ERR_ARG, "sgp_set_prior: form must be 0, 1 or 2");
    // mean + covariance: Lambda0 = Sigma0^-1 by Cholesky, xi0 = Lambda0 mu0 (ReactiveMP converts the
    // MvNormalMeanCovariance prior before the first product, GPnode/UniSGPnode.jl:62-63)
    int rc = upload_padded(h, mat, h->dTmp);
    if (rc) return rc;
    hipStream_t s = h->own;
    HIPCHK(h, hipMemsetAsync(h->dInfo + 2, 0, sizeof(int), s));
    launch_potrf(h->dTmp, h->Qp, h->TQ, h->dInfo + 2, h->Q, h->dScratch + 2 * POTRF_SCRATCH, s, h->dWl);
    launch_ata(h->dWl, h->dLambda0, h->Qp, h->TQ, s);
    // (the mean is staged in dXi and the factor in dTmp / dWl -- buffers every sweep rewrites before it reads them -- not in dMu:
    // sgp_get_posterior, sgp_predict(mu_v = NULL) and the theta gradient read the last sweep's mean from there)
    HIPCHK(h, hipMemcpyAsync(h->dXi, v.data(), Qp * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_symv, dim3((h->Qp + 3) / 4), dim3(256), 0, s, h->dLambda0, h->dXi, h->dXi0, h->Qp, h->Qp);
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    int info = 0;
    HIPCHK(h, hipMemcpy(&info, h->dInfo + 2, sizeof(int), hipMemcpyDeviceToHost));
    h->prior_form = 1;
    if (info > 0) { h->err = "sgp_set_prior: prior covariance is not positive definite"; return info; }
    return 0;
}

// prior <- posterior in natural form: Lambda0 += W (x) Psi2, xi0 += vec(B W).  The minibatch carry of
// experiments/regression_kin40k.ipynb:205-212 without the round trip through (mu, Sigma) on the host -- and without
// inverting Sigma_v again on the next sweep.
extern "C" int sgp_carry_posterior(sgp_handle* h, void* stream) {
    if (!h) return SGP_ERR_ARG;
    if (!h->swept || h->stats_dirty)
        return fail(h, SGP_ERR_ARG, "sgp_carry_posterior: call it right after a finished sweep (before sgp_theta_objective)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->own;
    // in place: Lambda0 += W (x) Psi2, xi0 += vec(B W), entry by entry (no scratch, no copies)
    hipLaunchKernelGGL(k_form_lambda, dim3(h->TQ, h->TQ), dim3(256), 0, s, h->dStats, h->dLambda0, h->dXi0, h->dLambda0, h->dXi0,
                       h->dParams, h->M, h->Mp, h->dout, h->Q, h->Qp, h->prior_form, 0, (int64_t*)nullptr, (int*)nullptr);
    HIPCHK(h, hipGetLastError());
    h->prior_form = 1;
    h->in_flight = true;                       // (asynchronous: a following setter must wait for it before it touches the prior)
    h->mirror_epoch = -1;
    return 0;
}

extern "C" int sgp_stats_layout(const sgp_handle* h, void** stats_dev, int64_t* count, int32_t* mp) {
    if (!h) return SGP_ERR_ARG;
    if (stats_dev) *stats_dev = h->dStats;
    if (count) *count = h->stats_count;
    if (mp) *mp = h->Mp;
    return 0;
}

extern "C" int sgp_bind_stats(sgp_handle* h, void* stats_dev) {
    if (!h) return SGP_ERR_ARG;
    if (int qrc = quiesce(h)) return qrc;
    h->dStats = stats_dev ? static_cast<double*>(stats_dev) : h->dStatsOwn;
    h->gLocal.valid = false;
    h->gFinish.valid = false;
    h->gFinish2.valid = false;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launch sequences
// ------------------------------------------------------------------------------------------------
// The sweep is four launch sequences:
//   K  (side stream) : K_uu chain -- theta and Xu only: K_uu, its Cholesky factor, inverse factor and inverse;
//                      starts with the sweep, joined before F2                                  [sgp_sweep_local]
//   L  (main stream) : data-sized work -- K_uf, Psi2 / B partials, packed statistics          [sgp_sweep_local]
//   F1 (main stream) : Lambda chain -- Lambda formed in step 0, Cholesky + inverse factor, mu, p, scan, Uv pass 1
//   F2 (main stream) : Sigma, R, both traces, Uv pass 2, scalars                               [sgp_sweep_finish]
// The two chains are latency-bound pivot sequences that use a handful of CUs each; they overlap only when they sit on
// different streams (parallel branches inside ONE captured graph were observed to execute back to back).
// the K_uu chain in three pieces -- [k_prep_xu, Gram] [Cholesky steps + inverse factor] [K_uu^-1, join word] -- so that a sweep can
// enqueue its two chains' steps alternately (enqueue_chains_interleaved)
static void kuu_gram(sgp_handle* h, hipStream_t s, bool gate) {
    if (gate) hipLaunchKernelGGL(k_gram_uu_lds, dim3(h->T, h->T), dim3(256), 0, s, h->dXusK, h->dKuu, h->dParamsK, h->M, h->Mp, h->D);
    else hipLaunchKernelGGL(k_gram_uu, dim3(h->T, h->T), dim3(256), 0, s, h->dXusK, h->dKuu, h->dParamsK, h->M, h->Mp, h->D);
}
static void kuu_tail(sgp_handle* h, hipStream_t s) {
    launch_ata(h->dWk, h->dKinv, h->Mp, h->T, s, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, h->dSaccK);
    if (h->join_by_flag) hipLaunchKernelGGL(k_join_set, dim3(1), dim3(64), 0, s, h->dJoin + WORD_JOIN, h->join_epoch);
}
static void enqueue_kuu_head(sgp_handle* h, hipStream_t s, bool steps_too) {
    const int M = h->M, Mp = h->Mp, D = h->D, T = h->T;
    const bool words = h->dev_words && s == h->side;
    // The gate (the chain's whole-CU workgroups stay off the chip until the SYRK's resident round is on it) sits inside k_prep_xu, in
    // front of K_uu's Gram: k_gram_uu_lds then starts together with the SYRK, gets a CU only as SYRK workgroups leave (it needs LDS, the
    // SYRK holds all of it) and the chain's first step begins a few microseconds after the SYRK has drained.  Round 4 tried to start
    // the chain earlier -- K_uu's Gram at the start of the sweep in front of the gate; an LDS-free Gram beside the SYRK with the
    // factorisation behind the word group 0's k_assemble sets -- and both cost the sweep 3 - 5 %: whatever FP64 work runs beside
    // group 0's SYRK, or whole-CU workgroups in its tail, lengthens that SYRK, which is on the critical path, by more than the
    // K_uu chain gains (profiles/r04_ab_log.txt [1], [9]).
    const bool gate = words && h->gate_side;
    hipLaunchKernelGGL(k_prep_xu, dim3((Mp + 255) / 256), dim3(256), 0, s, h->dXu, h->dXusK, h->params_src,
                       h->dParamsK, h->dInfo + 0, M, Mp, D, (int64_t*)nullptr, 0, 0,
                       words ? (const long long*)(h->dJoin + WORD_DONE) : (const long long*)nullptr, h->done_epoch,
                       gate ? (const long long*)(h->dJoin + WORD_GATE) : (const long long*)nullptr,
                       h->gate_epoch, h->spin_limit, h->dInfo + 3);
#ifdef SGP_WITH_PERSISTENT_CHAIN
    if (h->use_chain) {
        if (h->gate_kuu)
            hipLaunchKernelGGL(k_chain_gate, dim3(1), dim3(64), 0, s, (const long long*)(h->dChainFlags[0] + CH_F_GATE), h->gate_epoch);
        std::swap(h->dKuu, h->dKuuAlt);         // this launch's factor goes to the buffer the last one refilled with sentinels
        launch_chain(h, 0, h->dKuu, h->dKuuAlt, Mp, T, h->dInfo + 0, M, s, h->dWk, nullptr, h->dSaccK, nullptr, nullptr, h->dXusK,
                     h->dParamsK, M, D);
        kuu_tail(h, s);
        return;
    }
#endif
    kuu_gram(h, s, gate);
    if (steps_too) {
        launch_potrf(h->dKuu, Mp, T, h->dInfo + 0, M, h->dScratch, s, h->dWk, nullptr, h->dSaccK);
        kuu_tail(h, s);
    }
}
static void enqueue_kuu(sgp_handle* h, hipStream_t s) { enqueue_kuu_head(h, s, true); }

static void launch_gram(sgp_handle* h, hipStream_t s, bool opens_sweep) {
    int64_t* sweep_begin = opens_sweep ? h->dStamps : nullptr;
    if (h->D <= 8)
        hipLaunchKernelGGL(k_gram_uf<8>, dim3(h->nblk, h->T), dim3(256), 0, s, h->dXus, h->dX, h->dYw, h->dKuf, h->dBpart,
                           h->dParams, h->M, h->Mp, h->D, h->n, h->dout, h->dStamps + STAMP_STRIDE * SGP_T_GRAM, sweep_begin);
    else
        hipLaunchKernelGGL(k_gram_uf<MAXD>, dim3(h->nblk, h->T), dim3(256), 0, s, h->dXus, h->dX, h->dYw, h->dKuf, h->dBpart,
                           h->dParams, h->M, h->Mp, h->D, h->n, h->dout, h->dStamps + STAMP_STRIDE * SGP_T_GRAM, sweep_begin);
}

static void enqueue_local(sgp_handle* h, hipStream_t s) {
    const int M = h->M, Mp = h->Mp, D = h->D, T = h->T;
    // The scaled inducing inputs and the parameter mirror only change when a setter ran: a sweep at unchanged parameters
    // (VMP iterations at fixed theta) starts with the Gram kernel.  (Always in graph mode -- the captured sequence is fixed
    // -- and without data, where no Gram kernel exists to open the sweep's stamps.)
    const bool prep = h->main_prep_gen != h->params_gen || (h->cfg.flags & SGP_FLAG_GRAPH) || h->n <= 0;
    if (prep) {
        hipLaunchKernelGGL(k_prep_xu, dim3((Mp + 255) / 256), dim3(256), 0, s, h->dXu, h->dXus, h->params_src,
                           h->dParams, (int*)nullptr, M, Mp, D, h->dStamps, (int)SGP_T_COUNT, (int)SGP_T_SWEEP,
                           (const long long*)nullptr, 0LL, (const long long*)nullptr, 0LL, h->spin_limit, (int*)nullptr);
        h->main_prep_gen = h->params_gen;
    }
    if (h->n > 0) {
        launch_gram(h, s, true);
        long long* gate = h->gate_side ? h->dJoin + WORD_GATE : nullptr;
#ifdef SGP_WITH_PERSISTENT_CHAIN
        if (h->use_chain) gate = h->dChainFlags[0] + CH_F_GATE;
#endif
        launch_syrk(h->geom, s, h->dKuf, h->has_omega ? h->dOmega : nullptr, h->dSlabs, Mp, h->n, h->dStamps + STAMP_STRIDE * SGP_T_SYRK,
                    gate, h->gate_epoch);
    }
    SyrkGeom ga = h->geom;
    if (h->n <= 0) { ga = syrk_geometry(0, T, h->num_cus, 0); }       // no data: zero chunks, the statistics are zero
    // (data-sharded sweeps write the exchange buffer instead: lower tiles only, see exchange_stats)
    hipLaunchKernelGGL(k_assemble, dim3(T, T + 1, assemble_z(ga)), dim3(256), 0, s, h->dSlabs, h->dBpart, h->dDataScal,
                       h->pack_now ? h->dPack : h->dStats, Mp, T, ga, h->n > 0 ? h->nblk : 0, h->dout,
                       SGP_S_COUNT + h->dout * h->dout, 1, h->dStamps + STAMP_STRIDE * SGP_T_LOCAL, h->dInfo + 1, (long long*)nullptr, 0LL,
                       h->pack_now ? 1 : 0, h->dBred);
}

// The statistics of an overlapped sweep (see plan_overlap): the same kernels, the SYRK and the assembly once per tile-row group.
// Group 0 runs on all CUs on the sweep's own stream, in front of the Lambda chain (plain stream order: no hand-off on the
// critical path); the other groups on the CU-masked stream (statM), so that they cannot take the compute units the
// factorisation chains run on.  No events: statM's first kernel waits for the word the first block of group 0's k_assemble
// sets (group 0's SYRK has drained), a k_join_set behind each masked group's k_assemble writes the sweep's number into the
// group's word (the kernel boundary in front of it makes the statistics visible device-wide), and the chain step that forms the
// group polls that word.
static int exchange_stats(sgp_handle* h, hipStream_t s, int tile0, int ntile, bool with_tail);
// Data-sharded (an all-reduce hook is installed; round 4): the same schedule with ONE reduce per group, on the group's own stream --
// k_assemble writes the group's lower tiles into the exchange buffer, the hook sums that piece over the ranks (group 0's piece
// also carries B and the scalars: the last tile rows of Psi2 sit next to the buffer's tail), k_unpack_stats expands it.  Group 0's
// reduce is in front of the chain in plain stream order; a masked group's is published by an EVENT recorded behind its unpack, and
// the sweep's stream waits for it in front of the chain step that forms the group (launch_potrf's step_wait) -- not by a device
// word: words are waited for with a bounded spin, and a collective that builds its rings or waits for a straggler rank may take
// longer than that.  The chain therefore starts after (statistics of group 0 -> reduce of 0.6 MB) instead of (all statistics ->
// reduce of 1.18 MB), and the second reduce runs beside it.  Every rank calls the hook in the same order (group 0, 1, ...).
static int enqueue_stats_overlapped(sgp_handle* h, hipStream_t own) {
    const int M = h->M, Mp = h->Mp, D = h->D, T = h->T;
    const bool sharded = h->allreduce != nullptr;
    // (as in enqueue_local: a sweep at unchanged parameters starts with the Gram kernel)
    const bool prep = h->main_prep_gen != h->params_gen;
    if (prep) {
        hipLaunchKernelGGL(k_prep_xu, dim3((Mp + 255) / 256), dim3(256), 0, own, h->dXu, h->dXus, h->params_src,
                           h->dParams, (int*)nullptr, M, Mp, D, h->dStamps, (int)SGP_T_COUNT, (int)SGP_T_SWEEP,
                           (const long long*)nullptr, 0LL, (const long long*)nullptr, 0LL, h->spin_limit, (int*)nullptr);
        h->main_prep_gen = h->params_gen;
    }
    launch_gram(h, own, true);
    // When the masked groups may start (A/B switch SGP_G1_AFTER, read in sgp_create): 2 (default) = when group 0's assembly
    // starts, i.e. its SYRK has drained; 0 = as soon as group 0's SYRK has its round on the CUs (the masked SYRK then fills the
    // CUs as they drain, but group 0's assembly shares them with it: 12.7 instead of 8.7 us on the critical path);
    // 1 = when group 0 is assembled (the chains' whole-CU workgroups settle on the idle masked CUs meanwhile and the masked
    // SYRK no longer fits its single round).  Sweeps/s at T on one box: 3978 / 3975 / 3865.
    const int g1_mode = sharded ? 2 : h->env_g1_mode;
    const long long* g1_word = h->dJoin + (g1_mode == 0 ? WORD_GATE : (g1_mode == 1 ? WORD_GROUP0 : WORD_ASM0));
    hipLaunchKernelGGL(k_join_wait, dim3(1), dim3(64), 0, h->statM, g1_word, g1_mode == 0 ? h->gate_epoch : h->stat_epoch,
                       h->spin_limit, h->dInfo + 3, (int)SYNC_LATE_COLUMN, 8);
    for (int g = 0; g < h->ngroups; ++g) {
        const StatGroup& G = h->grp[g];
        hipStream_t s = G.masked ? h->statM : own;
        launch_syrk(G.geom, s, h->dKuf, h->has_omega ? h->dOmega : nullptr, h->dSlabs + G.slab_off, Mp, h->n,
                    h->dStamps + STAMP_STRIDE * SGP_T_SYRK, g == 0 ? h->dJoin + WORD_GATE : (long long*)nullptr, h->gate_epoch);
        hipLaunchKernelGGL(k_assemble, dim3(G.nrows, T + (g == 0 ? 1 : 0), assemble_z(G.geom)), dim3(256), 0, s, h->dSlabs + G.slab_off, h->dBpart,
                           h->dDataScal, sharded ? h->dPack : h->dStats, Mp, T, G.geom, h->nblk, h->dout,
                           SGP_S_COUNT + h->dout * h->dout, g == 0 ? 1 : 0, h->dStamps + STAMP_STRIDE * SGP_T_LOCAL,
                           g == 0 ? h->dInfo + 1 : (int*)nullptr, g == 0 ? h->dJoin + WORD_ASM0 : (long long*)nullptr, h->stat_epoch,
                           sharded ? 1 : 0, h->dBred);
        if (sharded) {
            if (int xrc = exchange_stats(h, s, G.geom.tile0, G.geom.ntiles, g == 0)) return xrc;
            if (G.masked && hipEventRecord(h->evGroup[g], s) != hipSuccess) return fail(h, SGP_ERR_HIP, "hipEventRecord failed (statistics group)");
        } else if (G.masked || g1_mode == 1)
            hipLaunchKernelGGL(k_join_set, dim3(1), dim3(64), 0, s, h->dJoin + WORD_GROUP0 + g, h->stat_epoch);
    }
    return 0;
}

static void enqueue_finish1(sgp_handle* h, hipStream_t s) {
    const int M = h->M, Mp = h->Mp, Q = h->Q, Qp = h->Qp, TQ = h->TQ;
    // Lambda is factored in index-reversed order (P Lambda P = L' L'^T): its inverse factor W' = L'^-1 then IS the upper
    // Cholesky factor of Sigma_v up to the reversal, and Uv follows by a rank-1 update instead of a third potrf.
    // (Lambda = Lambda0 + W (x) Psi2 and xi are formed by step 0 of the factorisation itself; the status word was reset by
    // k_assemble)
    LamForm form;
    memset(&form, 0, sizeof form);
    form.stats = h->dStats; form.Lambda0 = h->dLambda0; form.xi0 = h->dXi0; form.xi = h->dXi; form.P = h->dParams;
    form.M = M; form.Mp = Mp; form.d_out = h->dout; form.Q = Q; form.prior_form = h->prior_form;
    form.stamps = h->dStamps + STAMP_STRIDE * SGP_T_FINISH1;
    form.spin_limit = h->spin_limit;
    form.sync_status = h->dInfo + 3;
    form.trace_chain = 1;
    hipEvent_t step_wait[LAM_MAX_COLS] = {nullptr};
    bool any_wait = false;
    if (h->overlap_now) {
        // the statistics arrive group by group while the chain runs (enqueue_stats_overlapped): step G.form_step forms group G.
        // Group 0 was summed on this very stream: nothing to wait for.  (A step 0 resident from the start of the sweep and
        // waiting for its statistics itself was measured too: on T + 1 CUs it keeps group 0's SYRK from its full single round,
        // 54 instead of 35 us.)
        // (data-sharded: the masked groups arrive through events the stream waits for in front of their forming step -- nothing
        // for the kernel to poll, every column is "in stream order")
        form.col_words = h->dJoin + WORD_GROUP0;
        form.col_need = h->stat_epoch;
        for (int g = 0; g < h->ngroups; ++g) {
            for (int c = h->grp[g].c0; c < h->grp[g].c1; ++c) {
                form.form_step[c] = (unsigned char)h->grp[g].form_step;
                form.col_group[c] = (h->grp[g].masked && !h->allreduce) ? (unsigned char)g : (unsigned char)0xff;
            }
            if (h->grp[g].masked && h->allreduce) { step_wait[h->grp[g].form_step] = h->evGroup[g]; any_wait = true; }
        }
    }
    double* uvt0 = h->dUvWork + 2 * (size_t)Qp;     // t = W' P xi, advanced block by block during the factorisation
#ifdef SGP_WITH_PERSISTENT_CHAIN
    if (h->use_chain) {
        std::swap(h->dLam, h->dLamAlt);
        launch_chain(h, 1, h->dLam, h->dLamAlt, Qp, TQ, h->dInfo + 1, Qp, s, h->dWl, &form, h->dTmp, h->dXi, uvt0, nullptr, nullptr, 0, 0);
    } else
#endif
    {
        PotrfSeq lam(h->dLam, Qp, TQ, h->dInfo + 1, Qp, h->dScratch + POTRF_SCRATCH, s, h->dWl, &form, h->dTmp, h->dXi, uvt0,
                     any_wait ? step_wait : nullptr);
        if (h->kuu_deferred) {
            // (SGP_INTERLEAVE=1, off by default.)  The K_uu chain's Cholesky steps were held back (sweep_local_impl): its launches and the
            // Lambda chain's go out ALTERNATELY, so that on a GPU that is idle when the sweep arrives neither chain waits for the host to
            // get through the other's ten launches (the Lambda chain's first step arrives ~25 us after the GPU is ready for it).  Same
            // launches, same order on each stream.  Measured: the first A/B gained 1.3 % on 20-sweep blocks and 5 % on the sweep; w_stats
            // loop, two repeats on other boxes were neutral to -1 % -- switching streams between launches costs the host ~20 us per
            // sweep (120 instead of 100 us), which on a busy host is exactly what a short block lacks (profiles/r04_ab_log.txt [30]).
            PotrfSeq kuu(h->dKuu, h->Mp, h->T, h->dInfo + 0, h->M, h->dScratch, h->side, h->dWk, nullptr, h->dSaccK);
            while (!lam.done() || !kuu.done()) {
                if (!lam.done()) lam.next();
                if (!kuu.done()) kuu.next();
            }
            kuu_tail(h, h->side);
        } else {
            while (!lam.done()) lam.next();
        }
    }
    // mu = Sigma xi = P W'^T W' P xi as two triangular mat-vecs; their intermediate t IS p = V^-T mu up to the reversal,
    // so the closed-form Uv needs no further solve and nothing here waits for Sigma itself
    double* uvp = h->dXi;                    // xi is consumed by the forward solve; p lands in the same vector afterwards
    double* uvck = h->dUvWork;               // C_kk
    double* uvak = h->dUvWork + Qp;          // p_k / sqrt(alpha_k alpha_{k+1})
    double* uvt = uvt0;
    double* uvpart = uvt + (size_t)TQ * Qp;         // TQ x Qp tile partial sums
    // mu, p, the alpha scan and (extra workgroups) pass 1 of Uv in one launch
    hipLaunchKernelGGL(k_trmv_mu_scan, dim3(Qp / 4 + 1 + TQ * (TQ + 1) / 2), dim3(256), 0, s, (const double*)h->dWl,
                       (const double*)uvt, h->dMu, uvp, uvck, uvak, uvpart, Qp);
}

// after the join with the side stream (K_uu chain): Sigma, R, the traces, Uv pass 2 and the scalars
// k_scalars' pinned mirror (see there): on for eager sweeps outside a device-paced training run (whose loop reads nothing back)
static double* mirror_for(const sgp_handle* h) {
    return (h->hMirror && !(h->cfg.flags & SGP_FLAG_GRAPH) && !h->env_no_zero_copy && !h->training) ? h->hMirror : nullptr;
}

static void enqueue_finish2(sgp_handle* h, hipStream_t s) {
    const int M = h->M, Mp = h->Mp, Q = h->Q, Qp = h->Qp, TQ = h->TQ;
    double* uvp = h->dXi;
    double* uvck = h->dUvWork;
    double* uvak = h->dUvWork + Qp;
    double* uvpart = h->dUvWork + 2 * (size_t)Qp + (size_t)TQ * Qp;
    // Sigma = W'^T W' (index-reversed back), R = Sigma + mu mu^T, -- UniSGP -- the shares of tr(R Psi2) and tr(Kuu^-1 Psi2),
    // and (extra workgroups) pass 2 of Uv = chol(Sigma_v + mu mu^T).U (GPnode/UniSGPnode.jl:67-69): one launch
    const int nata = TQ * (TQ + 1) / 2 * 4;
    double* traceR = h->dTrace + TRACE_BLOCKS;           // UniSGP: [nata] R shares, then [nata] K shares
    const double* partK = h->dTrace;
    int nK = TRACE_BLOCKS, nR = TRACE_BLOCKS;
    UvArgs uv;
    uv.Wp = h->dWl; uv.p = uvp; uv.ck = uvck; uv.ak = uvak; uv.partial = uvpart; uv.LR = h->dUvT;
    uv.stamps = h->dStamps + STAMP_STRIDE * SGP_T_FINISH1;
    uv.join = h->join_by_flag ? h->dJoin + WORD_JOIN : nullptr;
    uv.join_need = h->join_epoch;
    uv.spin_limit = h->spin_limit;
    uv.sync_status = h->dInfo + 3;
    if (h->dout == 1) {
        launch_ata(h->dWl, h->dSigma, Qp, TQ, s, 1, h->dMu, h->dR, h->dStats, h->dKinv, traceR, &uv, h->dTmp);
        partK = traceR + nata;
        nK = nR = nata;
    } else {
        launch_ata(h->dWl, h->dSigma, Qp, TQ, s, 1, h->dMu, h->dR, nullptr, nullptr, nullptr, &uv, h->dTmp);
        hipLaunchKernelGGL(k_trace_kinv, dim3(TRACE_BLOCKS), dim3(256), 0, s, h->dStats, h->dKinv, h->dTrace, M, Mp);
        hipLaunchKernelGGL(k_trace_R, dim3(TRACE_BLOCKS), dim3(256), 0, s, h->dStats, h->dR, traceR, M, Mp, h->dout, Qp,
                           (int64_t*)nullptr);
    }
    hipLaunchKernelGGL(k_scalars, dim3(1), dim3(256), 0, s, h->dStats, partK, nK, (const double*)traceR, nR, h->dMu, h->dKuu,
                       h->dLam, h->dInfo, h->dParams, h->dOut, h->dWishart, M, Mp, h->dout, Q, Qp, Qp - Q,
                       h->dStamps + STAMP_STRIDE * SGP_T_FINISH2, h->dStamps, h->dStampTotals,
                       (h->cfg.flags & SGP_FLAG_GRAPH) ? (long long*)nullptr : h->dJoin + WORD_DONE, h->done_epoch,
                       h->use_chain ? (const double*)nullptr : (const double*)(h->dScratch + POTRF_LOGDET), h->T,
                       h->use_chain ? (const double*)nullptr : (const double*)(h->dScratch + POTRF_SCRATCH + POTRF_LOGDET), TQ,
                       mirror_for(h));
}

static int set_device_checked(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, SGP_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(nullptr, SGP_ERR_ARG, "bad device ordinal");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SGP_ERR_HIP, "hipSetDevice failed");
    return 0;
}

// device temporaries of the stand-alone building blocks: freed on every exit path
namespace {
struct EventPair {                      // a timing pair that is destroyed on every exit path
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
    hipError_t create() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
};
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};
}  // namespace


typedef void (*enqueue_fn)(sgp_handle*, hipStream_t);

static int run_sequence(sgp_handle* h, Graph& g, enqueue_fn fn, hipStream_t s) {
    if (!(h->cfg.flags & SGP_FLAG_GRAPH)) {
        fn(h, s);
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    if (!g.valid || g.key_n != h->n || g.key_prior != h->prior_form || g.key_stats != (void*)h->dStats ||
        g.key_omega != (int)h->has_omega) {
        g.reset();
        // capture on the library's own stream (the caller's stream may be the legacy default stream)
        HIPCHK(h, hipStreamBeginCapture(h->own, hipStreamCaptureModeThreadLocal));
        fn(h, h->own);
        hipError_t e = hipStreamEndCapture(h->own, &g.graph);
        if (e != hipSuccess) { h->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return SGP_ERR_HIP; }
        HIPCHK(h, hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
        g.key_n = h->n; g.key_prior = h->prior_form; g.key_stats = h->dStats; g.key_omega = (int)h->has_omega; g.valid = true;
    }
    HIPCHK(h, hipGraphLaunch(g.exec, s));
    return 0;
}

static int check_ready(sgp_handle* h) {
    if (!h) return SGP_ERR_ARG;
    if (!h->have_inducing || !h->have_data || !h->have_kernel)
        return fail(h, SGP_ERR_ARG, "sweep: set_inducing, set_data and set_kernel must be called first");
    return 0;
}

// `overlapped`: the statistics go to the handle's own statistics streams in tile-row groups and the Lambda chain (sgp_sweep_finish
// on the library's stream) starts on the first group while the others are still being summed (plan_overlap); only from
// sgp_sweep, which owns both halves.
static int sweep_local_impl(sgp_handle* h, void* stream, bool overlapped) {
    int rc = check_ready(h);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->own;
#ifdef SGP_WITH_PERSISTENT_CHAIN
    if (h->use_chain) {
        if (g_chain_owner && g_chain_owner != h) HIPCHK(h, hipDeviceSynchronize());
        g_chain_owner = h;
    }
#endif
    if (h->sync_reported) {
        // (the getter that reported the word had drained the device: nothing is in flight)
        HIPCHK(h, hipMemset(h->dInfo + 3, 0, sizeof(int)));
        h->sync_reported = false;
    }
    h->in_flight = true;
    h->overlap_now = overlapped;
    // The K_uu chain depends on theta and Xu only: it starts on the (low-priority) side stream as soon as the previous
    // sweep has finished with its outputs, runs beside the data-sized kernels, the all-reduce and the Lambda chain, and is
    // joined just before the Sigma launch.  Eager launches meet through device words (the chain's first kernel waits for the
    // previous sweep's done word itself, see enqueue_kuu): an event record or wait between two kernels of a stream was
    // measured at ~6 us of idle time.  Captured graphs use events -- and so does a sweep with an all-reduce hook, in addition:
    // the words are waited for with a BOUNDED spin (~1 s), and a collective that builds its rings or waits for a straggler
    // rank may take longer than that.
    const bool graph = (h->cfg.flags & SGP_FLAG_GRAPH) != 0;
    const bool hooked = h->allreduce != nullptr;
    h->dev_words = !graph;
    // (a caller's stream as well: two-phase callers -- sgp_sweep_local, their own reduce, sgp_sweep_finish -- may sit in a collective
    // between the halves for longer than the bounded words wait, exactly like the hook)
    h->use_events = graph || hooked || s != h->own;
    if (h->use_events) HIPCHK(h, hipStreamWaitEvent(h->side, h->evDone, 0));
    // How F2 will join the K_uu chain: an event wait between two kernels of the main stream costs it ~5 us of idle time even
    // when the event fired long ago, so the UniSGP path lets the Sigma launch's product workgroups poll a device word in
    // front of their epilogue instead (k_join_set behind the chain's last kernel).  Only while that launch leaves enough
    // CUs free for a late chain's workgroups (each needs a whole CU's LDS) -- which nobody can promise when a collective or a
    // second handle's sweep may hold CUs too --, and not inside captured graphs.
    {
        const int grid = h->TQ * (h->TQ + 1) / 2 * 4 + h->TQ * h->TQ;
        h->join_by_flag = h->dJoin && h->dout == 1 && !h->use_events && grid <= h->num_cus - 40 && !h->env_join_event &&
                          g_live_handles.load() <= 1;
        ++h->join_epoch;
    }
    ++h->gate_epoch;
    if (overlapped) ++h->stat_epoch;                            // (what the words of this sweep's statistics groups receive)
#ifdef SGP_WITH_PERSISTENT_CHAIN
    h->gate_kuu = h->use_chain && h->n > 0;                    // (a SYRK launch follows on the main stream and opens the gate)
#endif
    // Host order of the enqueues.  The K_uu chain is 14 launches (~50 us of host time).  Where it is gated behind the SYRK anyway
    // (gate_side) the data-sized kernels go out FIRST: a sweep that starts on an idle GPU -- the drop-in's pattern: sweep, fetch
    // something, next sweep -- otherwise has its Gram kernel wait ~50 us for the host to get through launches the GPU cannot run
    // yet.  (Back-to-back sweeps are enqueued a sweep ahead either way.)  Small problems keep the chain first: there the two chains
    // are the sweep.  With an all-reduce hook too: the hook is a host callback that may block (gloo), the chain should be queued by then.
    const bool stats_first = h->gate_side && h->n > 0 && !graph && !h->allreduce;
    auto enqueue_stats = [&]() -> int {
        if (overlapped) {
            if (int orc = enqueue_stats_overlapped(h, s)) return orc;
            HIPCHK(h, hipGetLastError());
            return 0;
        }
        return run_sequence(h, h->gLocal, enqueue_local, s);
    };
    if (stats_first)
        if (int src = enqueue_stats()) return src;
    // (one-shot sgp_sweep on the library's streams: only the K_uu chain's first two kernels go out here, its steps are enqueued
    // alternately with the Lambda chain's by sgp_sweep_finish -- see enqueue_finish1)
    h->kuu_deferred = h->defer_request && stats_first && !h->use_events && !h->use_chain && s == h->own;
    if (h->kuu_deferred) {
        enqueue_kuu_head(h, h->side, false);
        HIPCHK(h, hipGetLastError());
        rc = 0;
    } else
        rc = run_sequence(h, h->gKuu, enqueue_kuu, h->side);
#ifdef SGP_WITH_PERSISTENT_CHAIN
    h->gate_kuu = false;
#endif
    if (rc) return rc;
    if (!h->kuu_deferred) HIPCHK(h, hipEventRecord(h->evSide, h->side));
    if (!stats_first)
        if (int src = enqueue_stats()) return src;
    h->stats_dirty = false;
    h->swept_params = *h->hParams;
    h->swept_data_gen = h->data_gen;
    h->swept_local = true;
    return 0;
}

extern "C" int sgp_sweep_local(sgp_handle* h, void* stream) { return sweep_local_impl(h, stream, false); }

extern "C" int sgp_sweep_finish(sgp_handle* h, void* stream) {
    if (!h) return SGP_ERR_ARG;
    if (!h->swept_local) return fail(h, SGP_ERR_ARG, "sgp_sweep_finish: call sgp_sweep_local first");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->own;
    h->in_flight = true;
    int rc = run_sequence(h, h->gFinish, enqueue_finish1, s);
    if (rc) return rc;
    if (h->kuu_deferred) {                                   // (the K_uu chain's steps and tail went out inside enqueue_finish1)
        HIPCHK(h, hipEventRecord(h->evSide, h->side));
        h->kuu_deferred = false;
    }
    if (!h->join_by_flag) HIPCHK(h, hipStreamWaitEvent(s, h->evSide, 0));          // join with the K_uu chain
    ++h->done_epoch;                                         // what this sweep's k_scalars writes when it is through
    rc = run_sequence(h, h->gFinish2, enqueue_finish2, s);
    if (rc) return rc;
    h->mirror_epoch = mirror_for(h) ? h->done_epoch : -1;
    // the next sweep's K_uu chain may overwrite K_uu^-1 after this
    if (h->use_events) HIPCHK(h, hipEventRecord(h->evDone, s));
    h->overlap_now = false;
    h->swept = true;
    h->last_stream = s;
    return 0;
}

// The one exchange step of a data-sharded sweep: the ranks sum the exchange buffer k_assemble just wrote -- the LOWER tiles of
// Psi2, B and the scalars: 1.18 MB at M = 512 where the full symmetric statistics are 2.10 MB -- through the hook, on the sweep's
// stream (the K_uu chain keeps running on the side stream meanwhile); one more launch expands the sum into the statistics buffer.
// (tile0, ntile: the lower tiles of the piece, in the row-major triangle order of the exchange buffer; with_tail: B and the scalars,
// which follow the last tile, travel with it -- the whole buffer in the plain order, one piece per statistics group in the
// overlapped order, the first group's piece being [its tiles | B | scalars]: the LAST tile rows of Psi2 sit next to the tail)
static int exchange_stats(sgp_handle* h, hipStream_t s, int tile0, int ntile, bool with_tail) {
    const int tail = (int)(h->Mp * h->dout + SGP_S_COUNT + h->dout * h->dout);
    const int64_t count = (int64_t)ntile * TB * TB + (with_tail ? tail : 0);
    if (with_tail && tile0 + ntile != h->ntiles) return fail(h, SGP_ERR_ARG, "internal: the tail of the exchange buffer follows its last tile");
    if (h->allreduce(h->allreduce_ctx, h->dPack + (size_t)tile0 * TB * TB, count, s))
        return fail(h, SGP_ERR_HIP, "the all-reduce hook failed (statistics)");
    hipLaunchKernelGGL(k_unpack_stats, dim3(ntile + 1), dim3(256), 0, s, (const double*)h->dPack, h->dStats, h->Mp, h->T,
                       with_tail ? tail : 0, tile0, ntile);
    HIPCHK(h, hipGetLastError());
    return 0;
}
static int exchange_stats(sgp_handle* h, hipStream_t s) { return exchange_stats(h, s, 0, h->ntiles, true); }

extern "C" int sgp_sweep(sgp_handle* h, void* stream) {
    if (!h) return SGP_ERR_ARG;
    // single GPU, the library's own streams, a problem that qualifies: statistics and Lambda chain overlapped
    // (with an all-reduce hook as well: one reduce per statistics group, see enqueue_stats_overlapped)
    const bool overlapped = h->overlap && !stream && h->n > 0 && !h->training;
    h->pack_now = h->allreduce != nullptr;
    h->defer_request = h->env_interleave;
    int rc = sweep_local_impl(h, stream, overlapped);
    h->defer_request = false;
    h->pack_now = false;
    if (rc) return rc;
    if (h->allreduce && !overlapped) {
        rc = exchange_stats(h, stream ? static_cast<hipStream_t>(stream) : h->own);
        if (rc) return rc;
    }
    return sgp_sweep_finish(h, stream);
}

static int ensure_pack(sgp_handle* h) {
    if (h->dPack) return 0;
    h->pack_count = (int64_t)h->ntiles * TB * TB + (int64_t)h->Mp * h->dout + SGP_S_COUNT + (int64_t)h->dout * h->dout;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dPack), sizeof(double) * (size_t)h->pack_count));
    return 0;
}

extern "C" int sgp_set_allreduce(sgp_handle* h, sgp_allreduce_fn fn, void* ctx) {
    if (!h) return SGP_ERR_ARG;
    if (int qrc = quiesce(h)) return qrc;
    if (int prc = ensure_pack(h)) return prc;
    h->allreduce = fn;
    h->allreduce_ctx = ctx;
    if (h->n > 0) plan_overlap(h, h->n);         // (a data-sharded sweep pays one collective per statistics group: one cut)
    return 0;
}

// ncclAllReduce(sendbuff, recvbuff, count, ncclDouble = 8, ncclSum = 0, comm, stream), looked up in the running process
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
static nccl_allreduce_t g_nccl_allreduce = nullptr;
static int rccl_hook(void* ctx, void* buf, int64_t count, void* stream) {
    sgp_handle* h = static_cast<sgp_handle*>(ctx);
    return g_nccl_allreduce(buf, buf, (size_t)count, /*ncclDouble*/ 8, /*ncclSum*/ 0, h->rccl_comm, static_cast<hipStream_t>(stream));
}
extern "C" int sgp_use_rccl(sgp_handle* h, void* nccl_comm) {
    if (!h || !nccl_comm) return fail(h, SGP_ERR_ARG, "sgp_use_rccl: null argument");
    if (!g_nccl_allreduce) g_nccl_allreduce = reinterpret_cast<nccl_allreduce_t>(dlsym(RTLD_DEFAULT, "ncclAllReduce"));
    if (!g_nccl_allreduce) return fail(h, SGP_ERR_ARG, "sgp_use_rccl: no ncclAllReduce in this process (load librccl first)");
    if (int qrc = quiesce(h)) return qrc;
    if (int prc = ensure_pack(h)) return prc;
    h->rccl_comm = nccl_comm;
    h->allreduce = rccl_hook;
    h->allreduce_ctx = h;
    if (h->n > 0) plan_overlap(h, h->n);
    return 0;
}

__global__ void k_clock_probe(long long* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999;
    const long long c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) { a = fma(a, b, 1e-9); a = fma(a, b, 1e-9); a = fma(a, b, 1e-9); a = fma(a, b, 1e-9); }
    const long long c1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (long long)a; }
}
extern "C" int sgp_measure_sclk_mhz(int32_t device, double* mhz) {
    if (!mhz) return SGP_ERR_ARG;
    int rc = set_device_checked(device);
    if (rc) return rc;
    sgp_handle* h = nullptr;
    DevBuf b;
    HIPCHK(h, b.alloc(3 * sizeof(long long)));
    hipLaunchKernelGGL(k_clock_probe, dim3(1024), dim3(256), 0, 0, b.as<long long>(), 20000);     // ~1 ms on every CU
    hipLaunchKernelGGL(k_clock_probe, dim3(1024), dim3(256), 0, 0, b.as<long long>(), 20000);
    HIPCHK(h, hipDeviceSynchronize());
    long long v[3];
    HIPCHK(h, hipMemcpy(v, b.p, sizeof v, hipMemcpyDeviceToHost));
    *mhz = v[1] > 0 ? 100.0 * (double)v[0] / (double)v[1] : 0.0;
    return 0;
}

// The same under matrix-core load: every wave issues independent v_mfma_f64_16x16x4_f64 back to back (4 accumulators), one
// resident round of 4 workgroups per CU.  out[0] = shader clock (MHz) during the loop, out[1] = FP64 matrix rate the loop
// attained (TFLOP/s, 2048 flop per instruction), out[2] = shader clock under the v_fma_f64 loop above, out[3] = number of CUs.
__global__ void __launch_bounds__(256) k_clock_probe_mfma(long long* out, int iters) {
    typedef double d4v __attribute__((ext_vector_type(4)));
    d4v c0v = {0.0, 0.0, 0.0, 0.0}, c1v = c0v, c2v = c0v, c3v = c0v;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    const long long c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    // (inline assembly with the accumulators pinned in VGPRs: written with the builtin, the compiler moved all four accumulators
    // between VGPRs and AGPRs around every round of this loop -- 64 copies per 4 MFMAs -- and the "matrix rate of the chip" this
    // probe reported through round 3 and half of round 4, 45 - 48 TFLOP/s, was the rate of THAT loop; the instruction itself
    // sustains 72 - 73, tools/dpp_f64_probe.hip)
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c0v) : "v"(a), "v"(b));
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c1v) : "v"(a), "v"(b));
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c2v) : "v"(a), "v"(b));
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c3v) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // (the last results are in flight: the compiler does not know these are MFMAs)
    const long long c1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (long long)(c0v[0] + c1v[1] + c2v[2] + c3v[3]); }
}
extern "C" int sgp_measure_clocks(int32_t device, double* out) {
    if (!out) return SGP_ERR_ARG;
    int rc = set_device_checked(device);
    if (rc) return rc;
    sgp_handle* h = nullptr;
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount, blocks = 4 * cus, iters = 4000;
    DevBuf b;
    HIPCHK(h, b.alloc(3 * sizeof(long long)));
    EventPair ev;
    HIPCHK(h, ev.create());
    hipLaunchKernelGGL(k_clock_probe_mfma, dim3(blocks), dim3(256), 0, 0, b.as<long long>(), iters);       // warm the clocks
    HIPCHK(h, hipEventRecord(ev.a, 0));
    hipLaunchKernelGGL(k_clock_probe_mfma, dim3(blocks), dim3(256), 0, 0, b.as<long long>(), iters);
    HIPCHK(h, hipEventRecord(ev.b, 0));
    HIPCHK(h, hipDeviceSynchronize());
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, ev.a, ev.b));
    long long v[3];
    HIPCHK(h, hipMemcpy(v, b.p, sizeof v, hipMemcpyDeviceToHost));
    out[0] = v[1] > 0 ? 100.0 * (double)v[0] / (double)v[1] : 0.0;
    out[1] = ms > 0.f ? (double)blocks * 4.0 * iters * 4.0 * 2048.0 / (ms * 1e-3) * 1e-12 : 0.0;
    rc = sgp_measure_sclk_mhz(device, out + 2);
    out[3] = cus;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// results
// ------------------------------------------------------------------------------------------------
static int sync_all(sgp_handle* h) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, drain_device(h));
    h->in_flight = false;
    return 0;
}

// A bounded wait on a device word gave up somewhere since the last check (dInfo[3], see SYNC_LATE_* in sgp_kernels.hip.h): what
// the wait protected -- the buffers the two streams hand each other -- cannot be trusted, so the results are refused by EVERY
// getter, until the next sweep is enqueued (sweep_local_impl clears the word then).  Call after a device synchronisation.
extern "C" int sgp_wait(sgp_handle* h) {
    if (!h) return SGP_ERR_ARG;
    return sync_all(h);
}

// The mirror speaks for the device when the last thing enqueued that can touch the status word is the sweep whose k_scalars wrote
// it (mirror_epoch), that kernel has run (its epoch is in the mirror; the caller has waited for the stream) and no getter has
// reported a give-up since.
static bool mirror_fresh(const sgp_handle* h) {
    return h->hMirror && h->mirror_epoch >= 0 && h->mirror_epoch == h->done_epoch && !h->sync_reported &&
           h->hMirror[SGP_R_COUNT + 1] == (double)h->mirror_epoch;
}

static int check_sync_status(sgp_handle* h) {
    if (mirror_fresh(h) && h->hMirror[SGP_R_COUNT] == 0.0) return 0;        // (a set word takes the slow path: message, sticky flag)
    int bits = 0;
    HIPCHK(h, hipMemcpy(&bits, h->dInfo + 3, sizeof(int), hipMemcpyDeviceToHost));
    if (bits == 0) return 0;
    h->sync_reported = true;                 // (sticky: every getter refuses until the next sweep is enqueued, which clears the word)
    std::string msg = "a bounded device-word wait gave up (stream hand-off not honoured; results refused):";
    if (bits & SYNC_LATE_DONE) msg += " [done word: the next sweep's first kernels started before the previous sweep had finished]";
    if (bits & SYNC_LATE_GRAD_START) msg += " [done word: the K_uu half of the theta gradient started before the sweep had finished]";
    if (bits & SYNC_LATE_KINV) msg += " [join word: the Sigma launch did not get K_uu^-1 from the K_uu chain]";
    if (bits & SYNC_LATE_GRAD_JOIN) msg += " [gradient word: the K_uu half of the theta gradient did not arrive]";
    if (bits & SYNC_LATE_COLUMN) msg += " [statistics words: a Lambda-chain step did not get its tile columns / the masked stream did not get K_uf]";
    h->err = msg;
    return SGP_ERR_HIP;
}

static int download_square(sgp_handle* h, const double* dsrc, int ld, int n, double* dst) {
    HIPCHK(h, hipMemcpy2D(dst, sizeof(double) * n, dsrc, sizeof(double) * ld, sizeof(double) * n, n, hipMemcpyDeviceToHost));
    return 0;
}

// Install an externally given q(v) for the per-point outputs: mu_v and Uv = chol(Sigma_v + mu mu^T).U.  The reference's
// cold rules are called with an arbitrary q_v / meta.Uv (GPtest.jl:173-181,221-229,257-292): with this, sgp_w_stats
// evaluates them at that posterior for whatever points sgp_set_data + sgp_sweep_local put on the device.
extern "C" int sgp_set_posterior(sgp_handle* h, const double* mu_v, const double* Uv) {
    if (!h || !mu_v || !Uv) return fail(h, SGP_ERR_ARG, "sgp_set_posterior: null argument");
    if (int qrc = quiesce(h)) return qrc;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t Q = h->Q, Qp = h->Qp;
    std::vector<double> m(Qp, 0.0), ut(Qp * Qp, 0.0);
    memcpy(m.data(), mu_v, Q * sizeof(double));
    for (size_t k = 0; k < Qp; ++k) {
        if (k >= Q) { ut[k * Qp + k] = 1.0; continue; }
        for (size_t j = k; j < Q; ++j) ut[k * Qp + j] = Uv[k + j * Q];      // column k of Uv^T = row k of Uv (upper part)
    }
    HIPCHK(h, hipMemcpy(h->dMu, m.data(), Qp * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->dUvT, ut.data(), Qp * Qp * sizeof(double), hipMemcpyHostToDevice));
    h->swept = true;
    h->stats_dirty = true;                     // q(v) no longer belongs to the statistics on the device
    return 0;
}

extern "C" int sgp_get_scalars(sgp_handle* h, double* out) {
    if (!h || !out) return fail(h, SGP_ERR_ARG, "sgp_get_scalars: null argument");
    if (!h->swept) return fail(h, SGP_ERR_ARG, "sgp_get_scalars: no finished sweep");
    int rc = sync_all(h);
    if (rc) return rc;
    if (int src = check_sync_status(h)) return src;
    if (mirror_fresh(h)) memcpy(out, h->hMirror, SGP_R_COUNT * sizeof(double));
    else HIPCHK(h, hipMemcpy(out, h->dOut, SGP_R_COUNT * sizeof(double), hipMemcpyDeviceToHost));
    if (out[SGP_R_INFO_KUU] < 0 || out[SGP_R_INFO_LAMBDA] < 0)
        return fail(h, SGP_ERR_HIP, "the persistent factorisation launch gave up waiting (deadlock guard): its workgroups were not all resident");
    if (out[SGP_R_INFO_KUU] > 0) { h->err = "K_uu is not positive definite"; return (int)out[SGP_R_INFO_KUU]; }
    if (out[SGP_R_INFO_LAMBDA] > 0) {
        // Lambda is factored from its last row upwards: report the natural index of the failing pivot
        out[SGP_R_INFO_LAMBDA] = (double)(h->Qp - (int)out[SGP_R_INFO_LAMBDA] + 1);
        h->err = "Lambda is not positive definite";
        return (int)std::max(1.0, out[SGP_R_INFO_LAMBDA]);
    }
    return 0;
}

extern "C" int sgp_get_posterior(sgp_handle* h, double* mu_v, double* Sigma_v, double* Uv) {
    if (!h) return SGP_ERR_ARG;
    if (!h->swept) return fail(h, SGP_ERR_ARG, "sgp_get_posterior: no finished sweep");
    int rc = sync_all(h);
    if (rc) return rc;
    if (int src = check_sync_status(h)) return src;
    int info[4];
    HIPCHK(h, hipMemcpy(info, h->dInfo, sizeof info, hipMemcpyDeviceToHost));
    if (info[0] < 0 || info[1] < 0)
        return fail(h, SGP_ERR_HIP, "the persistent factorisation launch gave up waiting (deadlock guard): its workgroups were not all resident");
    if (info[0] > 0) { h->err = "K_uu is not positive definite"; return info[0]; }
    if (info[1] > 0) { h->err = "Lambda is not positive definite"; return std::max(1, h->Qp - info[1] + 1); }
    const int Q = h->Q, Qp = h->Qp;
    if (mu_v) HIPCHK(h, hipMemcpy(mu_v, h->dMu, sizeof(double) * Q, hipMemcpyDeviceToHost));
    if (Sigma_v) { rc = download_square(h, h->dSigma, Qp, Q, Sigma_v); if (rc) return rc; }
    if (Uv) {
        hipLaunchKernelGGL(k_transpose, dim3(h->TQ, h->TQ), dim3(256), 0, h->own, h->dUvT, h->dTmp, Qp);
        HIPCHK(h, hipStreamSynchronize(h->own));
        rc = download_square(h, h->dTmp, Qp, Q, Uv);
        if (rc) return rc;
        for (int j = 0; j < Q; ++j)
            for (int i = j + 1; i < Q; ++i) Uv[(size_t)j * Q + i] = 0.0;
    }
    return 0;
}

extern "C" int sgp_get_stats(sgp_handle* h, double* Psi2, double* B, double* scalars) {
    if (!h) return SGP_ERR_ARG;
    if (!h->swept_local) return fail(h, SGP_ERR_ARG, "sgp_get_stats: no sweep yet");
    int rc = sync_all(h);
    if (rc) return rc;
    if (int src = check_sync_status(h)) return src;
    const int M = h->M, Mp = h->Mp;
    if (Psi2) { rc = download_square(h, h->dStats, Mp, M, Psi2); if (rc) return rc; }
    if (B)
        HIPCHK(h, hipMemcpy2D(B, sizeof(double) * M, h->dStats + (size_t)Mp * Mp, sizeof(double) * Mp, sizeof(double) * M,
                              h->dout, hipMemcpyDeviceToHost));
    if (scalars)
        HIPCHK(h, hipMemcpy(scalars, h->dStats + (size_t)Mp * Mp + (size_t)Mp * h->dout, SGP_S_COUNT * sizeof(double),
                            hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int sgp_get_kuu_chol(sgp_handle* h, double* KuuL) {
    if (!h || !KuuL) return fail(h, SGP_ERR_ARG, "sgp_get_kuu_chol: null argument");
    if (!h->swept) return fail(h, SGP_ERR_ARG, "sgp_get_kuu_chol: no finished sweep");
    int rc = sync_all(h);
    if (rc) return rc;
    if (int src = check_sync_status(h)) return src;
    rc = download_square(h, h->dKuu, h->Mp, h->M, KuuL);
    if (rc) return rc;
    for (int j = 0; j < h->M; ++j)
        for (int i = 0; i < j; ++i) KuuL[(size_t)j * h->M + i] = 0.0;
    return 0;
}

extern "C" int sgp_get_wishart_invscale(sgp_handle* h, double* S) {
    if (!h || !S) return fail(h, SGP_ERR_ARG, "sgp_get_wishart_invscale: null argument");
    if (!h->swept) return fail(h, SGP_ERR_ARG, "sgp_get_wishart_invscale: no finished sweep");
    if (h->dout == 1) return fail(h, SGP_ERR_ARG, "sgp_get_wishart_invscale: d_out = 1 (use sgp_get_scalars)");
    int rc = sync_all(h);
    if (rc) return rc;
    if (int src = check_sync_status(h)) return src;
    std::vector<double> tmp(MAXO * MAXO);
    HIPCHK(h, hipMemcpy(tmp.data(), h->dWishart, sizeof(double) * MAXO * MAXO, hipMemcpyDeviceToHost));
    for (int i = 0; i < h->dout * h->dout; ++i) S[i] = tmp[i];
    return 0;
}

extern "C" int sgp_get_timestamps(sgp_handle* h, int64_t* out) {
    if (!h || !out) return SGP_ERR_ARG;
    int rc = sync_all(h);
    if (rc) return rc;
    // the last sweep's (begin, end) pairs, kept by the sweep's closing kernel behind the totals (the live records are reset there)
    HIPCHK(h, hipMemcpy(out, h->dStampTotals + SGP_T_COUNT + 1, 2 * SGP_T_COUNT * sizeof(int64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int sgp_get_chain_trace(sgp_handle* h, int32_t which, int64_t* out) {
    if (!h || !out || which < 0 || which > 1) return SGP_ERR_ARG;
#ifdef SGP_WITH_PERSISTENT_CHAIN
    if (!h->dChainTrace[which]) return fail(h, SGP_ERR_ARG, "sgp_get_chain_trace: create the handle with SGP_CHAIN_TRACE set in the environment");
    int rc = sync_all(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(out, h->dChainTrace[which], sizeof(long long) * CH_TMAX * 32, hipMemcpyDeviceToHost));
    return 0;
#else
    return fail(h, SGP_ERR_ARG, "sgp_get_chain_trace: this build does not contain the persistent factorisation launch");
#endif
}

extern "C" int sgp_get_step_trace(int64_t* out) {
    if (!out) return SGP_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return SGP_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_trace), sizeof(long long) * 8 * 64) != hipSuccess) return SGP_ERR_HIP;
    return 0;
}

// diagnostics of the variant library built with -DSGP_SWEEP_TRACE (all zeros otherwise): see g_sweep_trace in sgp_kernels.hip.h
extern "C" int sgp_get_sweep_trace(int64_t* out) {
    if (!out) return SGP_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return SGP_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_trace), sizeof(long long) * TRACE_SLOTS_N * 65) != hipSuccess) return SGP_ERR_HIP;
    return 0;
}

extern "C" int sgp_get_phase_totals(sgp_handle* h, int64_t* totals, int64_t* count, int32_t reset) {
    if (!h || !totals || !count) return SGP_ERR_ARG;
    int rc = sync_all(h);
    if (rc) return rc;
    int64_t buf[SGP_T_COUNT + 1];
    HIPCHK(h, hipMemcpy(buf, h->dStampTotals, sizeof buf, hipMemcpyDeviceToHost));
    for (int i = 0; i < SGP_T_COUNT; ++i) totals[i] = buf[i];
    *count = buf[SGP_T_COUNT];
    if (reset) HIPCHK(h, hipMemset(h->dStampTotals, 0, sizeof buf));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// HIP-event timing of one data-sized kernel of the sweep, launched eagerly `iters` times on `stream`
// (bench.py's roofline leg; both kernels are idempotent on the resident data)
// ------------------------------------------------------------------------------------------------
// the per-point quadratic forms |L_K^-1 k_n|^2, |Uv k_n|^2 and k_n . mu in one pass over the resident K_uf
static void launch_quadform(sgp_handle* h, hipStream_t s) {
    // (grid.y: 2 T work items per point block -- a factor's part of a row tile --, largest first: see k_quadform_fused)
    hipLaunchKernelGGL(k_quadform_fused, dim3(h->nblk, 2 * h->T), dim3(256), 0, s, h->dWk, h->dUvT, h->dKuf, h->dMu, h->dPa, h->dPb, h->dKmu,
                       h->Mp, h->T, h->n);
}

extern "C" int sgp_time_kernel(sgp_handle* h, int32_t which, int32_t iters, void* stream, double* avg_us) {
    if (!h || !avg_us || iters < 1) return fail(h, SGP_ERR_ARG, "sgp_time_kernel: bad argument");
    if (!h->swept_local || h->n == 0) return fail(h, SGP_ERR_ARG, "sgp_time_kernel: run a sweep on non-empty data first");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->own;
    // which = SGP_TIME_GROUP0 + g: the SYRK launch of statistics group g of the overlapped sweep, on the stream (and CUs) it runs on
    const StatGroup* G = nullptr;
    if (which >= SGP_TIME_GROUP0 && which < SGP_TIME_GROUP0 + LAM_MAX_GROUPS) {
        if (!h->overlap || which - SGP_TIME_GROUP0 >= h->ngroups) return fail(h, SGP_ERR_ARG, "sgp_time_kernel: no such statistics group");
        HIPCHK(h, hipDeviceSynchronize());
        G = &h->grp[which - SGP_TIME_GROUP0];
        s = G->masked ? h->statM : h->own;
    }
    // which = SGP_TIME_QUADFORM: k_quadform_fused of sgp_w_stats (|L^-1 k_n|^2 with W_K and |Uv k_n|^2 in one launch)
    const int qmode = (which == SGP_TIME_QUADFORM) ? 0 : -1;
    if (qmode >= 0 && (!(h->cfg.flags & SGP_FLAG_KEEP_KUF) || !h->swept))
        return fail(h, SGP_ERR_ARG, "sgp_time_kernel: the per-point kernels need SGP_FLAG_KEEP_KUF and a finished sweep");
    EventPair ev;
    HIPCHK(h, ev.create());
    hipEvent_t e0 = ev.a, e1 = ev.b;
    auto launch = [&]() {
        if (qmode >= 0)
            launch_quadform(h, s);
        else if (G)
            launch_syrk(G->geom, s, h->dKuf, h->has_omega ? h->dOmega : nullptr, h->dSlabs + G->slab_off, h->Mp, h->n, (int64_t*)nullptr,
                        (long long*)nullptr, 0LL);
        else if (which == SGP_T_GRAM && h->D <= 8)
            hipLaunchKernelGGL(k_gram_uf<8>, dim3(h->nblk, h->T), dim3(256), 0, s, h->dXus, h->dX, h->dYw, h->dKuf, h->dBpart,
                               h->dParams, h->M, h->Mp, h->D, h->n, h->dout, (int64_t*)nullptr, (int64_t*)nullptr);
        else if (which == SGP_T_GRAM)
            hipLaunchKernelGGL(k_gram_uf<MAXD>, dim3(h->nblk, h->T), dim3(256), 0, s, h->dXus, h->dX, h->dYw, h->dKuf, h->dBpart,
                               h->dParams, h->M, h->Mp, h->D, h->n, h->dout, (int64_t*)nullptr, (int64_t*)nullptr);
        else
            launch_syrk(h->geom, s, h->dKuf, h->has_omega ? h->dOmega : nullptr, h->dSlabs, h->Mp, h->n, (int64_t*)nullptr,
                        (long long*)nullptr, 0LL);
    };
    if (!G && qmode < 0 && which != SGP_T_GRAM && which != SGP_T_SYRK)
        return fail(h, SGP_ERR_ARG, "sgp_time_kernel: which must be SGP_T_GRAM, SGP_T_SYRK or SGP_TIME_GROUP0 + g");
    launch();
    HIPCHK(h, hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch();
    HIPCHK(h, hipEventRecord(e1, s));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    *avg_us = 1e3 * (double)ms / iters;
    return 0;
}

// The plan of the overlapped sweep for the resident data (plan_overlap): *ngroups = 0 if the next sgp_sweep will not be an
// overlapped one; else, per group g, info[8 g ..] = first / past-the-last tile column of P Lambda P, lower tiles, point chunks,
// points per chunk, 1 if it runs on the CU-masked stream, CUs it may use, the Lambda-chain step that forms it.
extern "C" int sgp_overlap_plan(const sgp_handle* h, int32_t* ngroups, int32_t* info) {
    if (!h || !ngroups) return SGP_ERR_ARG;
    const bool on = h->overlap && !h->training;
    *ngroups = on ? h->ngroups : 0;
    if (!on || !info) return 0;
    for (int g = 0; g < h->ngroups; ++g) {
        const StatGroup& G = h->grp[g];
        int32_t* o = info + 8 * g;
        o[0] = G.c0; o[1] = G.c1; o[2] = G.ntiles; o[3] = G.geom.nchunks; o[4] = G.geom.chunk; o[5] = G.masked ? 1 : 0;
        o[6] = G.masked ? h->stat_cus_masked : h->num_cus; o[7] = G.form_step;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-point :w quantities
// ------------------------------------------------------------------------------------------------
extern "C" int sgp_w_stats(sgp_handle* h, double* I1, double* I2, void* stream) {
    if (!h) return SGP_ERR_ARG;
    if (!(h->cfg.flags & SGP_FLAG_KEEP_KUF)) return fail(h, SGP_ERR_ARG, "sgp_w_stats: create the handle with SGP_FLAG_KEEP_KUF");
    if (!h->swept) return fail(h, SGP_ERR_ARG, "sgp_w_stats: no finished sweep");
    if (h->dout != 1) return fail(h, SGP_ERR_ARG, "sgp_w_stats: d_out must be 1");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->own;
    const int64_t n = h->n;
    if (n == 0) return 0;
    double *dI1 = nullptr, *dI2 = nullptr;
    // Ordered after the sweep by the stream when that is the stream the sweep's tail was enqueued on (the usual case: both the
    // library's own -- no device-wide wait in front of the launches); any other pairing -- the sweep on a caller's stream and this
    // call on the library's, or the reverse -- is not ordered by anything, so then the device is drained first.  The scratch
    // grows only here and in sgp_predict, both blocking: a reallocation never races with a queued launch.
    if (s != h->last_stream || 2 * (size_t)n > h->call_capacity) HIPCHK(h, hipDeviceSynchronize());
    if (int crc = call_scratch(h, 2 * (size_t)n, &dI1)) return crc;
    dI2 = dI1 + n;
    // |L^-1 k_n|^2 with the explicit inverse factor W_k, |Uv k_n|^2 = |L_R^T k_n|^2 and k_n . mu: ONE pass over the resident K_uf
    // (k_quadform_fused), then the fixed-order sums
    launch_quadform(h, s);
    // Both vectors come back through the pinned staging block when they fit (two blocking copies into pageable memory were ~60 us of
    // a 280 us call at n = 10 000) -- and the finishing kernel writes them THERE (pinned host memory is mapped into the device's
    // address space: 160 KB of coalesced stores over the bus under the kernel, instead of a copy-engine launch behind it)
    const bool staged = h->hStage && 2 * (size_t)n <= h->stage_doubles && !h->env_no_zero_copy;
    hipLaunchKernelGGL(k_w_point_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->dPa, h->dPb, h->dKmu, h->dY,
                       h->has_yv ? h->dYv : nullptr, staged ? h->hStage : dI1, staged ? h->hStage + n : dI2, h->dParams, h->T, n);
    HIPCHK(h, wait_stream(s));
    HIPCHK(h, hipGetLastError());
    if (int src = check_sync_status(h)) return src;          // (the sweep whose q(v) these are: a hand-off that gave up voids them too)
    if (staged) {
        if (I1) memcpy(I1, h->hStage, sizeof(double) * n);
        if (I2) memcpy(I2, h->hStage + n, sizeof(double) * n);
    } else {
        if (I1) HIPCHK(h, hipMemcpy(I1, dI1, sizeof(double) * n, hipMemcpyDeviceToHost));
        if (I2) HIPCHK(h, hipMemcpy(I2, dI2, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// prediction
// ------------------------------------------------------------------------------------------------
template <int DT>
static void launch_predict(sgp_handle* h, const double* dXs, const double* dMu, double* dMean, int64_t ns, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_predict<DT>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, h->dXusK, dXs, dMu,
                       dMean, h->dParamsK, h->M, h->Mp, h->D, ns, h->dout);
}

// (the same on the MAIN stream's copies of the scaled inducing inputs / parameters: the classification training step's forward message)
template <int DT>
static void launch_predict_main(sgp_handle* h, const double* dXs, const double* dMu, double* dMean, int64_t ns, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_predict<DT>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, h->dXus, dXs, dMu,
                       dMean, h->dParams, h->M, h->Mp, h->D, ns, h->dout);
}

extern "C" int sgp_predict(sgp_handle* h, const double* Xstar, int64_t ns, const double* mu_v, double* mean) {
    if (!h || !Xstar || !mean || ns < 0) return fail(h, SGP_ERR_ARG, "sgp_predict: bad argument");
    if (!h->have_inducing || !h->have_kernel) return fail(h, SGP_ERR_ARG, "sgp_predict: set_inducing and set_kernel first");
    if (!mu_v && !h->swept) return fail(h, SGP_ERR_ARG, "sgp_predict: no posterior in the handle and mu_v is NULL");
    if (h->training) return fail(h, SGP_ERR_ARG, "sgp_predict: a device-paced training run is open (sgp_train_end first)");
    if (ns == 0) return 0;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, drain_device(h));
    h->in_flight = false;
    hipStream_t s = h->own;
    double *dXs = nullptr, *dMean = nullptr, *dMuTmp = nullptr;
    if (int crc = call_scratch(h, (size_t)ns * (h->D + h->dout) + (mu_v ? (size_t)h->Q : 0), &dXs)) return crc;
    dMean = dXs + (size_t)ns * h->D;
    HIPCHK(h, hipMemcpy(dXs, Xstar, sizeof(double) * ns * h->D, hipMemcpyHostToDevice));
    const double* dMu = h->dMu;
    if (mu_v) {
        dMuTmp = dMean + (size_t)ns * h->dout;
        HIPCHK(h, hipMemcpy(dMuTmp, mu_v, sizeof(double) * h->Q, hipMemcpyHostToDevice));
        dMu = dMuTmp;
    }
    // The CURRENT kernel parameters go into the K_uu chain's mirror (dParamsK / dXusK: every sweep rewrites it first thing), not
    // into dParams / dXus: those must keep the values of the last sweep, which sgp_carry_posterior and the gradient of
    // sgp_theta_objective at unchanged theta still read (a predict between set_noise and theta_objective scaled
    // the gradient by w_new / w_old twice).
    hipLaunchKernelGGL(k_prep_xu, dim3((h->Mp + 255) / 256), dim3(256), 0, s, h->dXu, h->dXusK, (const Params*)h->hParams,
                       h->dParamsK, (int*)nullptr, h->M, h->Mp, h->D, (int64_t*)nullptr, 0, 0, (const long long*)nullptr, 0LL,
                       (const long long*)nullptr, 0LL, h->spin_limit, (int*)nullptr);
    switch (h->D) {
        case 1: launch_predict<1>(h, dXs, dMu, dMean, ns, s); break;
        case 2: launch_predict<2>(h, dXs, dMu, dMean, ns, s); break;
        case 3: launch_predict<3>(h, dXs, dMu, dMean, ns, s); break;
        case 4: launch_predict<4>(h, dXs, dMu, dMean, ns, s); break;
        case 8: launch_predict<8>(h, dXs, dMu, dMean, ns, s); break;
        default: launch_predict<0>(h, dXs, dMu, dMean, ns, s); break;
    }
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpy(mean, dMean, sizeof(double) * ns * h->dout, hipMemcpyDeviceToHost));
    return 0;
}

// F(theta) = -sum_n [ -w/2 k_nn + w/2 |L^-1 k_n|^2 - w/2 |Uv k_n|^2 + w y_n mu_v'k_n ]  (derivative_helper.jl:23-39)
//          = w/2 [ s_kk - tr(Kuu^-1 Psi2) + tr(R Psi2) ] - w b'mu_v
// evaluated at the CURRENT kernel parameters with q(v) (mu_v, R = Sigma_v + mu mu') held fixed at the last finished
// sweep -- exactly how the notebooks call it (experiments/regression_kin40k.ipynb:212-221: q(v) from the sweep, then the
// gradient step on theta).  Re-uses the sweep's kernels: K_uu chain + Gram/SYRK at theta, then the trace kernels.
// analytic gradient w.r.t. (sigma2, ell_1 .. ell_n_ell) of the objective at the resident K_uf, Psi2, K_uu^-1, R and mu, into
// h->dGrad: one G K_uf GEMM contracted with the kernel derivatives in its
// epilogue, plus the K_uu term through H = Kinv Psi2 Kinv (see k_theta_grad_* in sgp_kernels.hip.h)
static int enqueue_theta_grad(sgp_handle* h, hipStream_t s) {
    const int Mp = h->Mp, T = h->T;
    const int nblk_max = (int)((h->n_max + TB - 1) / TB);
    if (!h->dGradM) {
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dGradM), sizeof(double) * 3 * (size_t)Mp * Mp));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dGradPart),
                            sizeof(double) * ((size_t)std::max(std::max(nblk_max, 1) * T, 512 + T) + (size_t)T * T) * GRAD_SLOTS));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dGrad), sizeof(double) * 2 * GRAD_SLOTS));
    }
    double* dG = h->dGradM;
    double* dT1 = dG + (size_t)Mp * Mp;
    double* dH = dT1 + (size_t)Mp * Mp;
    double* part_uu = h->dGradPart;
    double* part_uf = h->dGradPart + (size_t)T * T * GRAD_SLOTS;
    const size_t cnt = (size_t)Mp * Mp;
    // split the K loop of the G K_uf product when there are few point blocks (minibatches), so that the launch fills the chip
    const int KS = h->n > 0 ? std::max(1, std::min(T, 512 / std::max(1, h->nblk * T))) : 1;
    const int n_uf = h->n > 0 ? h->nblk * T * KS : 0;
    // The gradient has two independent halves: the data half (G = R - K_uu^-1, then G K_uf contracted with the kernel
    // derivatives) and the K_uu half (H = K_uu^-1 Psi2 K_uu^-1 against dK_uu).  On the library's own streams they run side
    // by side -- the K_uu half on the side stream, which idles between two sweeps -- and meet in the finishing kernel through
    // a device word (an event would cost the main stream ~6 us, see sgp_sweep_finish).
    const bool split = s == h->own && h->dJoin && !(h->cfg.flags & SGP_FLAG_GRAPH) && !h->env_grad_one_stream;
    h->mirror_epoch = -1;                      // (the gradient's hand-offs report into the same status word)
    hipStream_t su = split ? h->side : s;
    if (split)
        hipLaunchKernelGGL(k_join_wait, dim3(1), dim3(64), 0, su, (const long long*)(h->dJoin + WORD_DONE), h->done_epoch,
                           h->spin_limit, h->dInfo + 3, (int)SYNC_LATE_GRAD_START, 8);
    hipLaunchKernelGGL(k_gemm32, dim3(T * T * 4), dim3(256), 0, su, (const double*)h->dKinv, (const double*)h->dStats, dT1,
                       Mp, T, 3, 0, 0, (const double*)nullptr, (double*)nullptr, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, UvArgs{}, (const double*)nullptr);
    hipLaunchKernelGGL(k_gemm32, dim3(T * T * 4), dim3(256), 0, su, (const double*)dT1, (const double*)h->dKinv, dH, Mp, T,
                       3, 0, 0, (const double*)nullptr, (double*)nullptr, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, UvArgs{}, (const double*)nullptr);
    hipLaunchKernelGGL(k_theta_grad_uu, dim3(T, T), dim3(256), 0, su, dH, h->dXus, h->dParams, part_uu, h->M, Mp, h->D);
    if (split) hipLaunchKernelGGL(k_join_set, dim3(1), dim3(64), 0, su, h->dJoin + WORD_GRAD, ++h->grad_epoch);
    hipLaunchKernelGGL(k_form_G, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, h->dR, h->dKinv, dG, cnt);
    if (h->n > 0)
        hipLaunchKernelGGL(k_theta_grad_uf, dim3(h->nblk, T, KS), dim3(256), 0, s, dG, h->dKuf, h->dX, h->dXus, h->dYw,
                           h->has_omega ? h->dOmega : nullptr, h->dMu, h->dParams, part_uf, Mp, T, h->D, h->n);
    // Data-sharded run (an all-reduce hook is installed; the statistics in dStats are the reduced ones): the data half above used
    // this rank's K_uf, X and y -- its fixed-order total (1 + D doubles) is summed over the ranks through the same hook; the
    // K_uu half and the s_w term come from the reduced statistics and are replicated.  Every rank ends with the whole gradient.
    const double* uf_src = part_uf;
    int uf_blocks = n_uf;
    if (h->allreduce) {
        double* tot = h->dGrad + GRAD_SLOTS;
        hipLaunchKernelGGL(k_theta_grad_fold, dim3(1), dim3(256), 0, s, (const double*)part_uf, n_uf, tot, h->D);
        if (h->allreduce(h->allreduce_ctx, tot, GRAD_SLOTS, s)) return fail(h, SGP_ERR_HIP, "the all-reduce hook failed (theta gradient)");
        uf_src = tot;
        uf_blocks = 1;
    }
    hipLaunchKernelGGL(k_theta_grad_finish, dim3(1), dim3(256), 0, s, uf_src, uf_blocks, part_uu, T * T,
                       h->dStats + (size_t)Mp * Mp + (size_t)Mp * h->dout, h->dParams, h->dGrad, h->D, h->n_ell,
                       split ? (const long long*)(h->dJoin + WORD_GRAD) : (const long long*)nullptr, h->grad_epoch, h->spin_limit,
                       h->dInfo + 3);
    return 0;
}

static int theta_objective_eval(sgp_handle* h, hipStream_t s, double* value) {
    enqueue_kuu(h, s);
    h->pack_now = h->allreduce != nullptr;
    enqueue_local(h, s);
    h->pack_now = false;
    // (data-sharded run: the statistics re-formed at the new theta are this rank's -- sum them like a sweep's)
    if (h->allreduce)
        if (int xrc = exchange_stats(h, s)) return xrc;
    h->main_prep_gen = 0;       // this evaluation opens phase stamps that no closing kernel folds: let the next sweep's k_prep_xu reset them
    const int M = h->M, Mp = h->Mp, Q = h->Q, Qp = h->Qp;
    hipLaunchKernelGGL(k_trace_kinv, dim3(TRACE_BLOCKS), dim3(256), 0, s, h->dStats, h->dKinv, h->dTrace, M, Mp);
    hipLaunchKernelGGL(k_trace_R, dim3(TRACE_BLOCKS), dim3(256), 0, s, h->dStats, h->dR, h->dTrace + TRACE_BLOCKS, M, Mp, h->dout,
                       Qp, (int64_t*)nullptr);
    hipLaunchKernelGGL(k_scalars, dim3(1), dim3(256), 0, s, h->dStats, (const double*)h->dTrace, (int)TRACE_BLOCKS,
                       (const double*)(h->dTrace + TRACE_BLOCKS), (int)TRACE_BLOCKS, h->dMu, h->dKuu, h->dLam, h->dInfo, h->dParams,
                       h->dOut2, h->dWishart, M, Mp, h->dout, Q, Qp, Qp - Q, (int64_t*)nullptr, (int64_t*)nullptr,
                       (int64_t*)nullptr, (long long*)nullptr, 0LL, (const double*)nullptr, 0, (const double*)nullptr, 0, (double*)nullptr);
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    double out[SGP_R_COUNT], sc[SGP_S_COUNT];
    HIPCHK(h, hipMemcpy(out, h->dOut2, sizeof out, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sc, h->dStats + (size_t)Mp * Mp + (size_t)Mp * h->dout, sizeof sc, hipMemcpyDeviceToHost));
    if (out[SGP_R_INFO_KUU] > 0) { h->err = "K_uu is not positive definite"; return (int)out[SGP_R_INFO_KUU]; }
    *value = 0.5 * h->hParams->W[0] * (out[SGP_R_SUM_I1] + out[SGP_R_SUM_I2] - sc[SGP_S_YY]);
    return 0;
}

extern "C" int sgp_theta_objective(sgp_handle* h, double* value, double* grad) {
    if (!h || !value) return fail(h, SGP_ERR_ARG, "sgp_theta_objective: null argument");
    if (!h->swept || h->dout != 1) return fail(h, SGP_ERR_ARG, "sgp_theta_objective: needs a finished UniSGP sweep (q(v))");
    if (!h->have_data || !h->have_kernel) return fail(h, SGP_ERR_ARG, "sgp_theta_objective: data and kernel must be set");
    if (h->training) return fail(h, SGP_ERR_ARG, "sgp_theta_objective: a device-paced training run is open (sgp_train_end first)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, drain_device(h));
    h->in_flight = false;
    if (int src = check_sync_status(h)) return src;
    hipStream_t s = h->own;
    // Same theta, data and noise as the sweep that produced q(v) -- the notebooks' call pattern
    // (experiments/regression_kin40k.ipynb:205-221 evaluates the gradient at the theta the sweep just used): K_uf, Psi2, b,
    // K_uu^-1 and the traces are still on the device, nothing is recomputed.  With an all-reduce hook installed the statistics
    // are the reduced ones and value and gradient are those of ALL shards (enqueue_theta_grad sums the data half through the
    // hook).  A caller-bound buffer WITHOUT a hook may have been reduced outside the library: then the statistics are
    // re-formed locally, as before.
    bool fresh = !h->stats_dirty && h->swept_data_gen == h->data_gen && (h->dStats == h->dStatsOwn || h->allreduce) &&
                 h->swept_params.sigma2 == h->hParams->sigma2 && h->swept_params.jitter == h->hParams->jitter;
    // the objective is linear in w: a new mean(q_w) (classification_banana.ipynb passes the UPDATED q(w)) only rescales it
    const double wscale = fresh ? h->hParams->W[0] / h->swept_params.W[0] : 1.0;
    for (int d = 0; d < h->D && fresh; ++d) fresh = h->swept_params.inv_ell[d] == h->hParams->inv_ell[d];
    int rc = 0;
    if (fresh) {
        double out[SGP_R_COUNT], sc[SGP_S_COUNT];
        HIPCHK(h, hipMemcpy(out, h->dOut, sizeof out, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(sc, h->dStats + (size_t)h->Mp * h->Mp + (size_t)h->Mp * h->dout, sizeof sc, hipMemcpyDeviceToHost));
        if (out[SGP_R_INFO_KUU] > 0) { h->err = "K_uu is not positive definite"; return (int)out[SGP_R_INFO_KUU]; }
        *value = 0.5 * h->hParams->W[0] * (out[SGP_R_SUM_I1] + out[SGP_R_SUM_I2] - sc[SGP_S_YY]);
        if (!grad) return 0;
    } else {
        h->stats_dirty = true;                   // the statistics now belong to the NEW theta, not to q(v)'s sweep
        rc = theta_objective_eval(h, s, value);
        if (rc || !grad) { h->swept_local = true; return rc; }
    }
    if (int grc = enqueue_theta_grad(h, s)) return grc;
    HIPCHK(h, wait_stream(s));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpy(grad, h->dGrad, sizeof(double) * (1 + h->n_ell), hipMemcpyDeviceToHost));
    for (int i = 0; i <= h->n_ell; ++i) grad[i] *= wscale;
    if (!fresh) h->swept_local = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Device-paced minibatch training: the loop of `PerformInference` (experiments/regression_kin40k.ipynb:196-230) with
// nothing but launches on the host side.  The training set is resident, a minibatch is a window of it, the optimiser
// (Flux's AdaMax, :222) and the softplus map of `kernel_gp` (:108) run on the device, and the next sweep's parameters are
// read from where the optimiser kernel wrote them.  The host never waits inside the loop; factorisation failures are
// counted on the device (the step is then skipped) and reported by sgp_train_end.
// ------------------------------------------------------------------------------------------------
extern "C" int sgp_train_begin(sgp_handle* h, const double* X, const double* y, int64_t n_total, const double* theta_raw,
                               int32_t n_ell, double jitter, double eta, double beta1, double beta2, double eps) {
    if (!h || !X || !y || !theta_raw) return fail(h, SGP_ERR_ARG, "sgp_train_begin: null argument");
    if (int qrc = quiesce(h)) return qrc;
    if (h->dout != 1) return fail(h, SGP_ERR_ARG, "sgp_train_begin: the theta objective is defined for UniSGP (d_out = 1)");
    if (h->cfg.flags & SGP_FLAG_GRAPH) return fail(h, SGP_ERR_ARG, "sgp_train_begin: not with SGP_FLAG_GRAPH (the window moves every step)");
    // (a caller-bound statistics buffer is fine here: inside sgp_train_step nobody but the library -- through the all-reduce hook,
    // if one is installed -- touches the statistics between the two halves of the sweep)
    if (!h->have_inducing) return fail(h, SGP_ERR_ARG, "sgp_train_begin: call sgp_set_inducing first");
    if (n_total < 1) return fail(h, SGP_ERR_ARG, "sgp_train_begin: empty training set");
    if (n_ell != 1 && n_ell != h->D) return fail(h, SGP_ERR_ARG, "sgp_train_begin: n_ell must be 1 or D");
    if (!(jitter >= 0.0) || !(eta > 0.0)) return fail(h, SGP_ERR_ARG, "sgp_train_begin: jitter >= 0 and eta > 0 required");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->train_N != n_total) {
        if (h->dTrainX) hipFree(h->dTrainX);
        if (h->dTrainY) hipFree(h->dTrainY);
        h->dTrainX = h->dTrainY = nullptr;
        h->train_N = 0;
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dTrainX), sizeof(double) * (size_t)n_total * h->D));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dTrainY), sizeof(double) * (size_t)n_total));
        h->train_N = n_total;
    }
    if (!h->dTrain) {
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dTrain), sizeof(TrainState)));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->dTrainParams), sizeof(Params)));
    }
    HIPCHK(h, hipMemcpy(h->dTrainX, X, sizeof(double) * (size_t)n_total * h->D, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->dTrainY, y, sizeof(double) * (size_t)n_total, hipMemcpyHostToDevice));
    TrainState st;
    memset(&st, 0, sizeof st);
    for (int i = 0; i <= n_ell; ++i) st.theta[i] = theta_raw[i];
    st.bp[0] = beta1; st.bp[1] = beta2;
    st.eta = eta; st.beta1 = beta1; st.beta2 = beta2; st.eps = eps;
    HIPCHK(h, hipMemcpy(h->dTrain, &st, sizeof st, hipMemcpyHostToDevice));
    // noise precision, E[log w], the isotropic prior and the jitter stay what the setters left; sigma2 and the lengthscales
    // come from theta on the device
    h->hParams->jitter = jitter;
    HIPCHK(h, hipMemcpy(h->dTrainParams, h->hParams, sizeof(Params), hipMemcpyHostToDevice));
    h->n_ell = n_ell;
    hipLaunchKernelGGL(k_train_adamax, dim3(1), dim3(64), 0, h->own, h->dTrain, (const double*)nullptr, (const double*)nullptr,
                       h->dTrainParams, h->D, n_ell, 0, (const int*)nullptr, (const Params*)nullptr, (const double*)nullptr);
    HIPCHK(h, hipStreamSynchronize(h->own));
    h->params_src = h->dTrainParams;
    h->have_kernel = true;
    h->training = true;
    h->train_probit = false;
    h->swept = h->swept_local = false;
    return 0;
}

// Classification run: call right after sgp_train_begin.  From then on the labels given there (0 / 1) are Probit observations of
// f (experiments/classification_banana.ipynb cell 7), q(w) = Gamma(shape, rate) is carried over the minibatches, and every
// sgp_train_step forms q(f) for its window on the device first (k_predict with the carried posterior mean -- zero before the
// first step --, k_probit_window), updates q(w) after the sweep and takes the optimiser step at the new mean(q_w).
extern "C" int sgp_train_likelihood(sgp_handle* h, int32_t kind, double shape, double rate) {
    if (!h) return SGP_ERR_ARG;
    if (!h->training) return fail(h, SGP_ERR_ARG, "sgp_train_likelihood: call sgp_train_begin first");
    if (kind != SGP_LIKELIHOOD_GAUSSIAN && kind != SGP_LIKELIHOOD_PROBIT) return fail(h, SGP_ERR_ARG, "sgp_train_likelihood: unknown kind");
    if (kind == SGP_LIKELIHOOD_PROBIT && (!(shape > 0.0) || !(rate > 0.0)))
        return fail(h, SGP_ERR_ARG, "sgp_train_likelihood: shape and rate of q(w) must be > 0");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipDeviceSynchronize());
    TrainState st;
    HIPCHK(h, hipMemcpy(&st, h->dTrain, sizeof st, hipMemcpyDeviceToHost));
    st.kind = kind == SGP_LIKELIHOOD_PROBIT ? 1.0 : 0.0;
    st.ga = shape; st.gb = rate;
    HIPCHK(h, hipMemcpy(h->dTrain, &st, sizeof st, hipMemcpyHostToDevice));
    h->train_probit = kind == SGP_LIKELIHOOD_PROBIT;
    if (h->train_probit) {
        Params P;
        HIPCHK(h, hipMemcpy(&P, h->dTrainParams, sizeof P, hipMemcpyDeviceToHost));
        P.W[0] = shape / rate;
        P.E_logw = std::log(shape / rate);
        HIPCHK(h, hipMemcpy(h->dTrainParams, &P, sizeof P, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemset(h->dMu, 0, sizeof(double) * h->Qp));           // the forward message of the first minibatch: k' 0
        double* scratch = nullptr;
        if (int crc = call_scratch(h, (size_t)h->n_max, &scratch)) return crc;   // mz of a window (sized now: the steps only enqueue)
    }
    return 0;
}

extern "C" int sgp_train_get_gamma(sgp_handle* h, double* shape_rate) {
    if (!h || !shape_rate) return SGP_ERR_ARG;
    if (!h->dTrain) return fail(h, SGP_ERR_ARG, "sgp_train_get_gamma: no training run on this handle");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipDeviceSynchronize());
    TrainState st;
    HIPCHK(h, hipMemcpy(&st, h->dTrain, sizeof st, hipMemcpyDeviceToHost));
    shape_rate[0] = st.ga; shape_rate[1] = st.gb;
    return 0;
}

// one minibatch = the window [offset, offset + n) of the resident set: sweep (:205-211), posterior carry (:212), gradient of
// the objective at that q(v) (:214-221) and the optimiser step (:222).  flags: SGP_TRAIN_LEARN takes the optimiser step,
// SGP_TRAIN_RESET_PRIOR first puts the isotropic prior of sgp_set_prior(form 2) back (the per-epoch reset, :203-204 -- the
// prior's form is a launch argument, so this costs nothing).  Asynchronous.
extern "C" int sgp_train_step(sgp_handle* h, int64_t offset, int64_t n, int32_t flags) {
    if (!h) return SGP_ERR_ARG;
    if (!h->training) return fail(h, SGP_ERR_ARG, "sgp_train_step: call sgp_train_begin first");
    if (offset < 0 || n < (h->allreduce ? 0 : 1) || offset + n > h->train_N || n > h->n_max)
        return fail(h, SGP_ERR_ARG, "sgp_train_step: window outside the resident set or larger than n_max");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = h->own;
    const bool learn = (flags & SGP_TRAIN_LEARN) != 0;
    if (flags & SGP_TRAIN_RESET_PRIOR) h->prior_form = 2;
    double *ownX = h->dX, *ownYw = h->dYw, *ownY = h->dY;
    h->dX = h->dTrainX + (size_t)offset * h->D;
    if (!h->train_probit) h->dYw = h->dY = h->dTrainY + offset;      // (classification: the window's q(f) goes to the handle's own buffers)
    h->has_omega = false;
    h->has_yv = h->train_probit;
    h->have_data = true;
    h->n_nodes = (double)n;
    h->data_gen++;
    h->params_gen++;                                           // theta moved: k_prep_xu mirrors the parameters again
    int rc = set_point_count(h, n);
    if (!rc && h->train_probit) {
        // q(f) of the window: forward message from the carried posterior mean at the current theta, then the Probit moments
        hipLaunchKernelGGL(k_prep_xu, dim3((h->Mp + 255) / 256), dim3(256), 0, s, h->dXu, h->dXus, h->params_src, h->dParams, (int*)nullptr,
                           h->M, h->Mp, h->D, h->dStamps, (int)SGP_T_COUNT, (int)SGP_T_SWEEP, (const long long*)nullptr, 0LL,
                           (const long long*)nullptr, 0LL, h->spin_limit, (int*)nullptr);
        h->main_prep_gen = h->params_gen;                      // (the sweep below starts with its Gram kernel)
        double* mz = h->dCall;
        if (n > 0) {
            switch (h->D) {
                case 1: launch_predict_main<1>(h, h->dX, h->dMu, mz, n, s); break;
                case 2: launch_predict_main<2>(h, h->dX, h->dMu, mz, n, s); break;
                case 3: launch_predict_main<3>(h, h->dX, h->dMu, mz, n, s); break;
                case 4: launch_predict_main<4>(h, h->dX, h->dMu, mz, n, s); break;
                case 8: launch_predict_main<8>(h, h->dX, h->dMu, mz, n, s); break;
                default: launch_predict_main<0>(h, h->dX, h->dMu, mz, n, s); break;
            }
        }
        hipLaunchKernelGGL(k_probit_window, dim3(1), dim3(256), 0, s, (const double*)(h->dTrainY + offset), (const double*)mz, n,
                           (const TrainState*)h->dTrain, h->dY, h->dYw, h->dYv, h->dDataScal, (int)SGP_S_COUNT + 1);
    }
    if (!rc) {
        if (!h->train_probit)
            hipLaunchKernelGGL(k_train_window, dim3(1), dim3(256), 0, s, (const double*)h->dY, n, h->dDataScal, (int)SGP_S_COUNT + 1);
        // (the sweep's own entry point: the product path here too.  Data-sharded: the window is this rank's slice of the
        // minibatch, sgp_sweep sums the statistics -- the window's data scalars ride in the same buffer -- through the hook)
        rc = sgp_sweep(h, nullptr);
        if (!rc) rc = sgp_carry_posterior(h, nullptr);
        if (!rc && learn) rc = enqueue_theta_grad(h, s);
        // (without a learning step the kernel only keeps the books: a failed factorisation is counted either way)
        if (!rc)
            hipLaunchKernelGGL(k_train_adamax, dim3(1), dim3(64), 0, s, h->dTrain, (const double*)h->dGrad, (const double*)h->dOut,
                               h->dTrainParams, h->D, h->n_ell, learn ? 1 : 2, (const int*)(h->dInfo + 3), (const Params*)h->dParams,
                               (const double*)(h->dStats + (size_t)h->Mp * h->Mp + (size_t)h->Mp * h->dout + SGP_S_N));
        // the next K_uu chain (side stream) reads the parameters this step wrote and overwrites the K_uu^-1 its gradient read
        if (!rc) hipLaunchKernelGGL(k_join_set, dim3(1), dim3(64), 0, s, h->dJoin + WORD_DONE, ++h->done_epoch);
        if (!rc && h->use_events) HIPCHK(h, hipEventRecord(h->evDone, s));   // (hooked run: the next K_uu chain waits on this, see sweep_local_impl)
    }
    h->dX = ownX; h->dYw = ownYw; h->dY = ownY;
    h->has_yv = false;
    h->have_data = false;                                      // the window is not the handle's data: set_data again after the run
    if (rc) return rc;
    if (hipGetLastError() != hipSuccess) return fail(h, SGP_ERR_HIP, "sgp_train_step: launch failed");
    return 0;
}

// waits for the queued steps; theta_raw[1 + n_ell] = the raw parameters after the last step; counts[2] = optimiser steps
// taken, minibatches skipped because a factorisation failed.  The handle is an ordinary one again afterwards (kernel set
// to softplus(theta), posterior getters valid, data must be set again before the next sweep).
extern "C" int sgp_train_end(sgp_handle* h, double* theta_raw, int64_t* counts) {
    if (!h) return SGP_ERR_ARG;
    if (!h->training) return fail(h, SGP_ERR_ARG, "sgp_train_end: no run is open");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipDeviceSynchronize());           // (the run stays open if this fails: steps may still be queued)
    h->training = false;
    h->params_src = h->hParams;
    h->in_flight = false;
    if (int src = check_sync_status(h)) return src;
    TrainState st;
    HIPCHK(h, hipMemcpy(&st, h->dTrain, sizeof st, hipMemcpyDeviceToHost));
    Params P;
    HIPCHK(h, hipMemcpy(&P, h->dTrainParams, sizeof P, hipMemcpyDeviceToHost));
    h->hParams->sigma2 = P.sigma2;
    for (int d = 0; d < h->D; ++d) h->hParams->inv_ell[d] = P.inv_ell[d];
    if (h->train_probit) { h->hParams->W[0] = P.W[0]; h->hParams->E_logw = P.E_logw; }      // mean(q_w) after the last minibatch
    h->train_probit = false;
    h->params_gen++;
    h->stats_dirty = true;                                     // the resident statistics belong to the previous theta
    h->swept_local = false;
    if (theta_raw) for (int i = 0; i <= h->n_ell; ++i) theta_raw[i] = st.theta[i];
    if (counts) { counts[0] = (int64_t)st.steps; counts[1] = (int64_t)st.rejected; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone building blocks (host in / host out, blocking)
// ------------------------------------------------------------------------------------------------
extern "C" int sgp_kernelmatrix(int32_t device, const double* A, int64_t na, const double* B, int64_t nb, int32_t d,
                                double sigma2, const double* ell, int32_t n_ell, double* K) {
    if (!A || !B || !K || !ell || d < 1 || d > MAXD || (n_ell != 1 && n_ell != d) || na < 0 || nb < 0)
        return fail(nullptr, SGP_ERR_ARG, "sgp_kernelmatrix: bad argument");
    int rc = set_device_checked(device);
    if (rc) return rc;
    if (na == 0 || nb == 0) return 0;
    sgp_handle* h = nullptr;
    Params P;
    memset(&P, 0, sizeof P);
    P.sigma2 = sigma2;
    for (int i = 0; i < d; ++i) P.inv_ell[i] = 1.0 / ell[n_ell == 1 ? 0 : i];
    DevBuf bA, bB, bK, bP;
    HIPCHK(h, bA.alloc(sizeof(double) * na * d));
    HIPCHK(h, bB.alloc(sizeof(double) * nb * d));
    HIPCHK(h, bK.alloc(sizeof(double) * na * nb));
    HIPCHK(h, bP.alloc(sizeof(Params)));
    HIPCHK(h, hipMemcpy(bA.p, A, sizeof(double) * na * d, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(bB.p, B, sizeof(double) * nb * d, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(bP.p, &P, sizeof(Params), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_kernelmatrix, dim3((unsigned)((na * nb + 255) / 256)), dim3(256), 0, 0, bA.as<double>(), bB.as<double>(),
                       bK.as<double>(), bP.as<Params>(), na, nb, d);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpy(K, bK.p, sizeof(double) * na * nb, hipMemcpyDeviceToHost));
    return 0;
}

static int dense_common(int32_t device, const double* A, int32_t n, double* out, bool inverse) {
    if (!A || !out || n < 1) return fail(nullptr, SGP_ERR_ARG, "sgp_potrf/potri: bad argument");
    int rc = set_device_checked(device);
    if (rc) return rc;
    sgp_handle* h = nullptr;
    const int np = round_up(n, TB), Tn = np / TB;
    std::vector<double> tmp((size_t)np * np, 0.0);
    for (int j = 0; j < np; ++j) {
        if (j < n) memcpy(&tmp[(size_t)j * np], &A[(size_t)j * n], sizeof(double) * n);
        else tmp[(size_t)j * np + j] = 1.0;
    }
    const size_t mat = sizeof(double) * np * np;
    DevBuf bA, bInfo, bScr, bW, bC, bL, bFar, bShip, bRinv, bFlags, bArgs;
    HIPCHK(h, bA.alloc(mat));
    HIPCHK(h, bInfo.alloc(sizeof(int)));
    HIPCHK(h, hipMemset(bInfo.p, 0, sizeof(int)));
    HIPCHK(h, hipMemcpy(bA.p, tmp.data(), mat, hipMemcpyHostToDevice));
    if (inverse) {
        HIPCHK(h, bW.alloc(mat));
        HIPCHK(h, bC.alloc(mat));
    }
    const double* factor = bA.as<double>();
#ifdef SGP_WITH_PERSISTENT_CHAIN
    const char* env = getenv("SGP_CHAIN");
    const bool chain = env && strcmp(env, "persistent") == 0 && Tn <= CH_TMAX;
    if (chain) {
        // one persistent launch (sgp_chain.hip.h); the factor goes to a buffer of its own (see ChainArgs::Ain)
        if (g_chain_owner) HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, bL.alloc(mat));
        HIPCHK(h, bFar.alloc(mat));
        HIPCHK(h, bShip.alloc(mat));
        HIPCHK(h, bRinv.alloc(sizeof(double) * np));
        HIPCHK(h, bFlags.alloc(sizeof(long long) * CH_F_COUNT));
        HIPCHK(h, hipMemset(bFlags.p, 0, sizeof(long long) * CH_F_COUNT));
        const unsigned nb = (unsigned)(((size_t)np * np + 255) / 256);
        hipLaunchKernelGGL(k_chain_fill, dim3(nb), dim3(256), 0, 0, bL.as<double>(), (size_t)np * np);
        hipLaunchKernelGGL(k_chain_fill, dim3(nb), dim3(256), 0, 0, bFar.as<double>(), (size_t)np * np);
        hipLaunchKernelGGL(k_chain_fill, dim3(nb), dim3(256), 0, 0, bShip.as<double>(), (size_t)np * np);
        hipLaunchKernelGGL(k_chain_fill, dim3((np + 255) / 256), dim3(256), 0, 0, bRinv.as<double>(), (size_t)np);
        ChainArgs g;
        memset(&g, 0, sizeof g);
        g.A = bL.as<double>(); g.Ain = bA.as<double>(); g.ld = np; g.Tn = Tn; g.info = bInfo.as<int>(); g.n_valid = n;
        g.Winv = inverse ? bW.as<double>() : nullptr; g.Far = bFar.as<double>(); g.Ship = bShip.as<double>();
        g.rinv_all = bRinv.as<double>();
        g.abortw = bFlags.as<long long>() + CH_F_ABORT;
        HIPCHK(h, bArgs.alloc(sizeof(ChainArgs)));
        HIPCHK(h, hipMemcpy(bArgs.p, &g, sizeof g, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_chol_chain, dim3(chain_blocks(Tn, inverse)), dim3(CH_THREADS), 0, 0, (const ChainArgs*)bArgs.as<ChainArgs>());
        if (inverse)
            for (int j = 2; j <= Tn; ++j)
                hipLaunchKernelGGL(k_chain_extras, dim3(2 * (j - 1) * (j < Tn ? 2 : 1)), dim3(256), 0, 0, bL.as<double>(), np, j, Tn,
                                   bW.as<double>(), (double*)nullptr, (const double*)nullptr, (double*)nullptr);
        factor = bL.as<double>();
    } else
#else
    const bool chain = false;
#endif
    {
        HIPCHK(h, bScr.alloc(sizeof(double) * POTRF_SCRATCH));
        HIPCHK(h, hipMemset(bScr.p, 0, sizeof(double) * POTRF_SCRATCH));
        launch_potrf(bA.as<double>(), np, Tn, bInfo.as<int>(), n, bScr.as<double>(), 0, bW.as<double>());
    }
    const double* result = factor;
    if (inverse) {
        launch_ata(bW.as<double>(), bC.as<double>(), np, Tn, 0);
        result = bC.as<double>();
    }
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipGetLastError());
    int info = 0;
    HIPCHK(h, hipMemcpy(&info, bInfo.p, sizeof(int), hipMemcpyDeviceToHost));
#ifdef SGP_WITH_PERSISTENT_CHAIN
    if (chain && getenv("SGP_CHAIN_DUMP")) {              // debugging aid: the mailbox matrices of this call, raw
        std::vector<double> hb((size_t)np * np);
        FILE* f = fopen(getenv("SGP_CHAIN_DUMP"), "wb");
        if (f) {
            int hdr[2] = {np, Tn};
            fwrite(hdr, sizeof(int), 2, f);
            for (void* src : {bL.p, bShip.p, bFar.p}) {
                hipMemcpy(hb.data(), src, mat, hipMemcpyDeviceToHost);
                fwrite(hb.data(), sizeof(double), hb.size(), f);
            }
            fclose(f);
        }
    }
#endif
    (void)chain;
    if (info < 0) return fail(nullptr, SGP_ERR_HIP, "the persistent factorisation launch gave up waiting (deadlock guard)");
    HIPCHK(h, hipMemcpy2D(out, sizeof(double) * n, result, sizeof(double) * np, sizeof(double) * n, n, hipMemcpyDeviceToHost));
    if (!inverse)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i) out[(size_t)j * n + i] = 0.0;
    if (info > 0) { g_create_error = "matrix is not positive definite"; return info; }
    return 0;
}

extern "C" int sgp_potrf(int32_t device, const double* A, int32_t n, double* L) { return dense_common(device, A, n, L, false); }
extern "C" int sgp_potri(int32_t device, const double* A, int32_t n, double* Ainv) { return dense_common(device, A, n, Ainv, true); }
