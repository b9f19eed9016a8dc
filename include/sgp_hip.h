/* sgp_hip.h -- C ABI of the MI355X (gfx950) sparse-GP VMP hot path.
 *
 * Drop-in boundary for the UniSGP / MultiSGP factor nodes of biaslab/GaussianProcessNode.
 * The reference evaluates the node per data point inside ReactiveMP message rules; this library
 * evaluates the same mathematics batched on the GPU.  Every entry point
 * below names the reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - plain C, no C++/torch/HIP types in signatures; `void* stream` is a hipStream_t (NULL = the
 *     library's own stream), `*_dev` pointers are raw device pointers (e.g. pointer(::ROCArray)
 *     from AMDGPU.jl or torch.Tensor.data_ptr()).
 *   - all matrices are Float64, column-major (Julia layout).  Points are packed D x N (one point per
 *     column), i.e. the memory of a C-ordered NumPy (N, D) array.
 *   - every function returns 0 on success; k > 0 = "leading minor k not positive definite"
 *     (LAPACK potrf convention; the Julia shim rethrows PosDefException(k)); negative = error
 *     (see SGP_ERR_*).  sgp_last_error() gives the text.
 *   - a handle is NOT re-entrant (the reference's meta is mutated in place too,
 *     helper_functions/gp_helperfunction.jl:33-44); one handle = one GPU = one process.
 */
#ifndef SGP_HIP_H
#define SGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGP_ABI_VERSION 1

#define SGP_ERR_ARG      (-1)   /* bad argument / state */
#define SGP_ERR_HIP      (-2)   /* HIP runtime error, or a bounded wait between the library's streams gave up (results refused) */
#define SGP_ERR_NODEVICE (-3)   /* no gfx950 device visible */
#define SGP_ERR_NOMEM    (-4)

/* flags for sgp_config.flags */
#define SGP_FLAG_NO_GRAPH   1   /* launch kernels eagerly -- the default since eager launches measured ~20 us per sweep faster
                                 * than hipGraph replay at every size (tools/graph_vs_eager.py); kept as a no-op */
#define SGP_FLAG_KEEP_KUF   2   /* keep K_uf resident for the per-point outputs (sgp_w_stats per_point) */
#define SGP_FLAG_GRAPH      4   /* replay the launch sequences as captured hipGraphs (opt-in; bitwise the same results) */
#define SGP_FLAG_PERSISTENT_CHAIN 8   /* EXPERIMENTAL, only in the variant library built with -DSGP_WITH_PERSISTENT_CHAIN
                                 * (libsgp_hip_chain.so; the default library returns SGP_ERR_ARG from sgp_create): factor K_uu and
                                 * Lambda with one persistent launch per factorisation (csrc/sgp_chain.hip.h) instead of one launch
                                 * per 64-column step.  Correct and deterministic (it differs from the default by rounding: right- vs
                                 * left-looking) but measured SLOWER on MI355X (24 vs 19.5 us per step at M = 512);
                                 * ignored with SGP_FLAG_GRAPH and for matrices of more than 12 tile rows (d_out * M > 768). */

typedef struct sgp_handle sgp_handle;

/* Limits of this build (sgp_create returns SGP_ERR_ARG beyond them): 1 <= d <= 32 (LDS coordinate panels),
 * 1 <= d_out <= 4 (register tiles of the MultiSGP reductions), d_out * m <= 4032 (LDS copy of the forward-solve vector). */
typedef struct sgp_config {
    int64_t n_max;    /* capacity in points of this handle (this rank's shard) */
    int32_t m;        /* inducing points M                 */
    int32_t d;        /* input dimension D (1..32)         */
    int32_t d_out;    /* outputs: 1 = UniSGP, 2..4 = MultiSGP (shared kernel) */
    int32_t device;   /* HIP device ordinal                */
    int32_t flags;    /* SGP_FLAG_*                        */
    int32_t reserved;
} sgp_config;

/* statistics slots appended after Psi2 and B in the packed statistics buffer (sgp_stats_layout) */
enum { SGP_S_YY = 0,      /* sum_n omega_n (mu_y^2 + v_y)   (d_out = 1; MultiSGP keeps Ryy separately) */
       SGP_S_W = 1,       /* sum_n omega_n                  (s_kk = sigma2 * S_W)                      */
       SGP_S_N = 2,       /* number of factor nodes                                                     */
       SGP_S_COUNT = 8 };

/* result scalars of a sweep (sgp_get_scalars) */
enum { SGP_R_SUM_I1 = 0,  /* sum_n I1_n = s_kk - tr(Kuu^-1 Psi2)                */
       SGP_R_SUM_I2 = 1,  /* sum_n I2_n                                          */
       SGP_R_ENERGY = 2,  /* sum_n average energy                                */
       SGP_R_INFO_KUU = 3,/* potrf info of K_uu (0 ok, k = failing minor)        */
       SGP_R_INFO_LAMBDA = 4, /* potrf info of Lambda                            */
       SGP_R_INFO_PRIOR = 5,  /* potrf info of Sigma0 when the prior is a covariance */
       SGP_R_LOGDET_KUU = 6,
       SGP_R_LOGDET_LAMBDA = 7,
       SGP_R_COUNT = 8 };

/* ---- lifetime ------------------------------------------------------------------------------
 * replaces: construction of UniSGPMeta / MultiSGPMeta and their scratch buffers
 * (helper_functions/gp_helperfunction.jl:33-44,55-64; GPCache :16-20,78-90). */
int sgp_abi_version(void);
int sgp_create(const sgp_config* cfg, sgp_handle** out);
int sgp_destroy(sgp_handle* h);
const char* sgp_last_error(const sgp_handle* h);   /* h may be NULL: last error of a failed sgp_create */

/* ---- inputs --------------------------------------------------------------------------------
 * sgp_set_inducing: meta.Xu (helper_functions/gp_helperfunction.jl:35; Vector{Vector} packed D x M). */
int sgp_set_inducing(sgp_handle* h, const double* Xu);
/* sgp_set_data: the data of one `infer` call -- x[i], y[i] of `y[i] ~ UniSGP(x[i], v, w, theta)`
 * (experiments/regression_kin40k.ipynb:147-152).  X is D x n; y_mean is n x d_out (column-major);
 * y_var (n, may be NULL) is var(q_out) of the classification rules (GPnode/UniSGPnode.jl:161-173,219-238);
 * pt_weight (n, may be NULL) are cubature weights omega of uncertain inputs (GPnode/UniSGPnode.jl:11-33,
 * GPnode/MultiSGPnode.jl:11-35); n_nodes = number of factor nodes the n points belong to (n if no cubature). */
int sgp_set_data(sgp_handle* h, const double* X, const double* y_mean, const double* y_var,
                 const double* pt_weight, int64_t n, double n_nodes);
/* sgp_set_output_cov_sum: MultiSGP with Gaussian (not PointMass) q_out: sum over the nodes of cov(q_out)
 * (d_out x d_out), the Sigma_y term of `Ry = Sigma_y + mu_y mu_y'` (GPnode/MultiSGPnode.jl:398-401,566).  Call after
 * sgp_set_data (which resets it to zero). */
int sgp_set_output_cov_sum(sgp_handle* h, const double* S);
/* sgp_set_kernel: kernel(theta) = sigma2 * with_lengthscale(SEKernel(), ell)
 * (GPtest.jl:21; experiments/regression_kin40k.ipynb:108).  n_ell = 1 (isotropic) or D (ARD).
 * jitter is added to diag(K_uu) (0 in kin40k training :183, 1e-8 at prediction :297 and in banana). */
int sgp_set_kernel(sgp_handle* h, double sigma2, const double* ell, int32_t n_ell, double jitter);
/* sgp_set_prior: `v ~ MvNormalMeanCovariance(mu_v, Sigma_v)` (experiments/regression_kin40k.ipynb:148).
 * form: 0 = mean + covariance, 1 = weighted mean xi0 + precision Lambda0, 2 = isotropic N(0, s I) with
 * s = mat[0] (the notebook's per-epoch reset 50 I, :203-204).  Vectors have d_out*M entries. */
int sgp_set_prior(sgp_handle* h, const double* vec, const double* mat, int32_t form);
/* sgp_set_noise: mean(q_w).  UniSGP: W[0] = w_bar and E_log_w = E[log w] (log w_bar for a PointMass,
 * GPnode/UniSGPnode.jl:340,414).  MultiSGP: W is d_out x d_out (mean of the Wishart), E_log_w = E[logdet W]. */
int sgp_set_noise(sgp_handle* h, const double* W, double E_log_w);

/* ---- the sweep -----------------------------------------------------------------------------
 * One VMP sweep over the resident data = what ReactiveMP does for `infer(iterations = 1)`:
 *   phase 1 (local):  K_uu + chol (experiments/regression_kin40k.ipynb:183-184), K_uf and the summed
 *                     :v messages  Psi2 = sum k_n k_n^T, b = sum mu_y k_n   (GPnode/UniSGPnode.jl:144-173)
 *   [multi-GPU: sum-all-reduce of the packed statistics buffer, see sgp_stats_layout]
 *   phase 2 (replicated): the N-fold product + marginal (GPnode/UniSGPnode.jl:62-73): Lambda, Sigma_v,
 *                     mu_v, Uv; the summed :w messages (GPnode/UniSGPnode.jl:196-238) and the summed
 *                     average energy (GPnode/UniSGPnode.jl:337-387,411-436).
 * All calls are asynchronous on `stream`; results are fetched with sgp_get_*.  */
int sgp_sweep_local(sgp_handle* h, void* stream);
int sgp_sweep_finish(sgp_handle* h, void* stream);
int sgp_sweep(sgp_handle* h, void* stream);                 /* local + [all-reduce hook] + finish */

/* ---- multi-GPU: the one exchange step of a sweep --------------------------------------------
 * Points are sharded over the ranks (one process = one GPU = one handle); Xu, theta and the prior are replicated.  The
 * statistics are sums over points (the N-fold message product of GPnode/UniSGPnode.jl:62-63 and the sequential minibatch
 * carry of experiments/regression_kin40k.ipynb:205-212 rely on the same additivity), so a sweep needs ONE sum-all-reduce of
 * the packed statistics buffer (sgp_stats_layout) between its two halves.  With a hook installed, sgp_sweep does
 *     local statistics  ->  hook(ctx, stats_dev, count, stream)  ->  replicated M^3 tail
 * inside the library, on `stream`; the K_uu chain runs beside all three on the library's side stream.
 * The hook must enqueue an in-place sum-all-reduce of `count` doubles at `stats_dev` on `stream` and return 0; it may be
 * RCCL's ncclAllReduce (sgp_use_rccl), MPI on device buffers, torch.distributed through a ctypes callback, or a test double.
 * fn = NULL removes the hook (single GPU).
 * What the hook is handed (always a buffer owned by the library, never the one of sgp_bind_stats):
 *   - inside sgp_sweep / sgp_train_step: the EXCHANGE buffer [lower 64 x 64 tiles of Psi2 | B | scalars] -- T (T + 1) / 2 tiles with
 *     T = ceil(M / 64): 1.18 MB at M = 512 where the full symmetric statistics are 2.10 MB; the sum is expanded into the
 *     statistics layout of sgp_stats_layout afterwards;
 *   - inside sgp_theta_objective / sgp_train_step: the data half of the theta gradient, 33 doubles (the K_uu half and the s_w
 *     term come from the reduced statistics).  With a hook installed value and gradient are those of ALL shards on every
 *     rank and nothing is recomputed. */
typedef int (*sgp_allreduce_fn)(void* ctx, void* stats_dev, int64_t count, void* stream);
int sgp_set_allreduce(sgp_handle* h, sgp_allreduce_fn fn, void* ctx);
/* Convenience: all-reduce with RCCL on the communicator `nccl_comm` (an ncclComm_t created by the host program, e.g. with
 * ncclCommInitRank over xGMI).  The library does not link RCCL: it looks ncclAllReduce up in the process (dlsym), so the
 * RCCL the host program already loaded is the one that is used.  Returns SGP_ERR_ARG if no RCCL is loaded. */
int sgp_use_rccl(sgp_handle* h, void* nccl_comm);
/* shader clock the chip holds under a short FP64 load (MHz): Delta s_memtime / Delta s_memrealtime x 100 MHz (bench.py reports it
 * next to the roofline fractions) */
int sgp_measure_sclk_mhz(int32_t device, double* mhz);
/* sgp_measure_clocks: the same under matrix-core load.  out[0] = shader clock (MHz) while every wave issues independent
 * v_mfma_f64_16x16x4_f64 back to back (one resident round, 4 workgroups per CU), out[1] = the FP64 matrix rate that loop
 * attains (TFLOP/s) -- the attainable peak the SYRK roofline can be priced against --, out[2] = shader clock under the
 * v_fma_f64 loop of sgp_measure_sclk_mhz, out[3] = number of CUs. */
int sgp_measure_clocks(int32_t device, double* out /* 4 */);

/* packed statistics buffer (device): [Psi2: Mp*Mp | B: Mp*d_out | scalars: SGP_S_COUNT (+ d_out*d_out Ryy)]
 * Mp = M rounded up to the tile size; count = total doubles to all-reduce. */
int sgp_stats_layout(const sgp_handle* h, void** stats_dev, int64_t* count, int32_t* mp);
/* let the caller own the statistics buffer (e.g. a torch tensor that torch.distributed all-reduces) */
int sgp_bind_stats(sgp_handle* h, void* stats_dev);

/* ---- results -------------------------------------------------------------------------------
 * sgp_get_posterior: mean_cov(qv) + meta.Uv (GPnode/UniSGPnode.jl:66-69).  Any pointer may be NULL.
 * mu_v: d_out*M; Sigma_v, Uv: (d_out*M)^2 column-major; Uv upper-triangular with Uv' Uv = Sigma_v + mu mu'. */
int sgp_get_posterior(sgp_handle* h, double* mu_v, double* Sigma_v, double* Uv);
/* (sgp_get_scalars waits for the handle's work, polled, and then reads a pinned block the sweep's last kernel wrote the scalars and the
 * hand-off status to -- no device-to-host copy behind the wait; SGP_NO_ZERO_COPY=1 restores the copies) */
int sgp_get_scalars(sgp_handle* h, double* out /* SGP_R_COUNT */);
/* sgp_get_stats: the reduced statistics (tests, theta-gradient): Psi2 M x M, B M x d_out, scalars SGP_S_COUNT */
int sgp_get_stats(sgp_handle* h, double* Psi2, double* B, double* scalars);
/* sgp_get_kuu_chol: meta.KuuL (helper_functions/gp_helperfunction.jl:39), lower, M x M */
int sgp_get_kuu_chol(sgp_handle* h, double* KuuL);
/* MultiSGP: inverse scale sum_t (I1_t + I2_t) of the Wishart messages (GPnode/MultiSGPnode.jl:391-404), d_out^2 */
int sgp_get_wishart_invscale(sgp_handle* h, double* S);

/* sgp_carry_posterior: prior <- posterior of the last finished sweep, on the device and in natural form
 * (Lambda0 += W (x) Psi2, xi0 += vec(B W)): the minibatch carry of experiments/regression_kin40k.ipynb:205-212
 * (`mu_v, Sigma_v = mean_cov(q_v)` fed back as the next prior).  Call after sgp_sweep and before sgp_theta_objective
 * (which re-evaluates the statistics at the new theta). */
int sgp_carry_posterior(sgp_handle* h, void* stream);

/* sgp_set_posterior: install an externally given q(v) -- mu_v (d_out*M) and Uv = chol(Sigma_v + mu mu').U (Q x Q
 * column-major, upper) -- for the per-point outputs.  The reference's cold rules are called with an arbitrary q_v /
 * meta.Uv (GPnode/UniSGPnode.jl:107-122,177-192,242-287; GPtest.jl:173-181,221-229,257-292): after sgp_set_data +
 * sgp_sweep_local (K_uu chain and K_uf at the current theta) + this call, sgp_w_stats evaluates I1_n / I2_n there. */
int sgp_set_posterior(sgp_handle* h, const double* mu_v, const double* Uv);

/* sgp_w_stats: per-point :w rule quantities (GPnode/UniSGPnode.jl:196-238):
 * I1_n = k_nn - |L^-1 k_n|^2 (the Q_ff diagonal term) and I2_n.  Needs SGP_FLAG_KEEP_KUF and a finished sweep.
 * Either output may be NULL. */
int sgp_w_stats(sgp_handle* h, double* I1 /* n */, double* I2 /* n */, void* stream);

/* sgp_predict: batched @call_rule UniSGP(:out) (GPnode/UniSGPnode.jl:96-104; loop
 * experiments/regression_kin40k.ipynb:288-304): mean[s] = K(x*_s, Xu) mu_v^(d).  Xstar is D x ns (host),
 * mu_v (host, d_out*M) or NULL to use the handle's current posterior; mean is ns x d_out. */
int sgp_predict(sgp_handle* h, const double* Xstar, int64_t ns, const double* mu_v, double* mean);

/* sgp_wait: returns when everything this handle has enqueued -- on its own streams or the caller's -- has finished: what a caller
 * does between `infer` calls when it wants the sweep to be over but none of its results yet.  The library's streams are polled
 * (hipStreamQuery, up to ~2 ms, then the blocking call): a blocking hipDeviceSynchronize may put the thread to sleep until an
 * interrupt, tens of microseconds behind a 0.22 ms sweep.  The getters wait the same way. */
int sgp_wait(sgp_handle* h);

/* sgp_theta_objective: neg_log_backwardmess_fast (helper_functions/derivative_helper.jl:23-39) evaluated at the CURRENT
 * kernel parameters (sgp_set_kernel) with q(v) -- mu_v and Uv'Uv -- held at the last finished sweep, as the notebooks use
 * it (experiments/regression_kin40k.ipynb:212-221).  grad (may be NULL): d/d(sigma2, ell_1..ell_n_ell), 1 + n_ell
 * entries (grad_llh_new!, derivative_helper.jl:59-63 uses ForwardDiff; here the analytic kernel-derivative contraction). */
int sgp_theta_objective(sgp_handle* h, double* value, double* grad);

/* ---- device-paced minibatch training: `PerformInference` of experiments/regression_kin40k.ipynb:196-230 ---------------
 * The reference's loop is, per minibatch, infer(iterations = 1) (:205-211), q(v) carried over as the next prior (:212),
 * grad_llh_new! at that q(v) (:214-221) and Flux.Optimise.update!(AdaMax, theta, grad) (:222) with
 * theta -> softplus(theta) inside `kernel_gp` (:108).  sgp_train_* keeps all of it on the device: the training set is
 * uploaded once, a minibatch is a window of it, the optimiser state lives in device memory and every sweep reads its
 * kernel parameters from where the optimiser kernel wrote them, so the host only enqueues and never waits in the loop.
 *   sgp_train_begin  X is n_total x D point-major (row i = point i), y n_total; theta_raw[1 + n_ell] the raw
 *                    (pre-softplus) parameters (sigma2 first); noise, prior and inducing inputs as the setters left them;
 *                    AdaMax(eta, (beta1, beta2), eps) starts from zero state.  UniSGP handles without SGP_FLAG_GRAPH.  With an
 *                    all-reduce hook installed (sgp_set_allreduce / sgp_use_rccl) the run is data-sharded: every rank
 *                    passes ITS slice of each minibatch to sgp_train_step (possibly empty), statistics and the data half
 *                    of the gradient are summed through the hook, AdaMax runs replicated on identical gradients and every
 *                    rank ends a step with the same theta (and, for SGP_LIKELIHOOD_PROBIT, the same q(w): its shape
 *                    counts the whole minibatch).  Until sgp_train_end every setter, sgp_predict and
 *                    sgp_theta_objective return SGP_ERR_ARG.
 *   sgp_train_step   one minibatch = points [offset, offset + n), n <= n_max.  flags: SGP_TRAIN_LEARN = gradient and
 *                    optimiser step (without it theta stays); SGP_TRAIN_RESET_PRIOR = before this minibatch the prior
 *                    goes back to the isotropic N(0, variance I) last given to sgp_set_prior(form 2) (the per-epoch
 *                    reset, :203-204).  Asynchronous.  A minibatch whose K_uu or Lambda is not positive definite leaves
 *                    theta alone and is counted.
 *   sgp_train_end    waits; theta_raw out (may be NULL); counts[0] = optimiser steps taken, counts[1] = minibatches
 *                    skipped (may be NULL).  The posterior getters then return the last minibatch's q(v); the kernel is
 *                    set to softplus(theta); sgp_set_data is needed again before another sgp_sweep. */
int sgp_train_begin(sgp_handle* h, const double* X, const double* y, int64_t n_total, const double* theta_raw,
                    int32_t n_ell, double jitter, double eta, double beta1, double beta2, double eps);
enum { SGP_TRAIN_LEARN = 1, SGP_TRAIN_RESET_PRIOR = 2 };
int sgp_train_step(sgp_handle* h, int64_t offset, int64_t n, int32_t flags);
/* Classification runs -- `PerformInference` of experiments/classification_banana.ipynb (cell 9; model cell 7:
 * `f[i] ~ UniSGP(x[i], v, w, theta); y[i] ~ Probit(f[i])`, mean-field q(f) q(v) q(w)).  sgp_train_likelihood, called right after
 * sgp_train_begin with kind = SGP_LIKELIHOOD_PROBIT, declares the y given there to be labels in {0, 1} and q(w) =
 * GammaShapeRate(shape, rate).  Every sgp_train_step then does, on the device: q(f_i) of its window from the :out message
 * N(k_i' mu_v, 1 / mean(q_w)) (GPnode/UniSGPnode.jl:96-104) with the carried posterior mean and the Probit likelihood; the sweep
 * with q_out = q(f) (the classification :v / :w rules, :161-173, :219-238); q(w) <- Gamma(shape + n / 2, rate + (sum I1 + sum I2) / 2);
 * the posterior carry; the optimiser step at the NEW mean(q_w).  q(v) and q(w) are never reset (no SGP_TRAIN_RESET_PRIOR).
 * sgp_train_get_gamma (after sgp_train_end): the final (shape, rate). */
enum { SGP_LIKELIHOOD_GAUSSIAN = 0, SGP_LIKELIHOOD_PROBIT = 1 };
int sgp_train_likelihood(sgp_handle* h, int32_t kind, double shape, double rate);
int sgp_train_get_gamma(sgp_handle* h, double* shape_rate /* 2 */);
int sgp_train_end(sgp_handle* h, double* theta_raw, int64_t* counts /* 2 */);

/* ---- building blocks exposed for tests / other callers (host pointers, blocking) ------------
 * K = sigma2 * exp(-0.5 |(a-b)/ell|^2): kernelmatrix(kernel(theta), A, B) of KernelFunctions.jl as called at
 * GPnode/UniSGPnode.jl:102,153; A is D x na, B is D x nb, K is na x nb column-major. */
int sgp_kernelmatrix(int32_t device, const double* A, int64_t na, const double* B, int64_t nb, int32_t d,
                     double sigma2, const double* ell, int32_t n_ell, double* K);
/* dense FP64 factorisations on the device (fastcholesky / cholinv call sites: GPnode/UniSGPnode.jl:68,
 * experiments/regression_kin40k.ipynb:184): A is n x n column-major symmetric; L lower; Ainv full. */
int sgp_potrf(int32_t device, const double* A, int32_t n, double* L);
int sgp_potri(int32_t device, const double* A, int32_t n, double* Ainv);

/* timing hooks for bench.py: device-side timestamps (100 MHz s_memrealtime) of the last sweep:
 * out[2*i], out[2*i+1] = begin/end ticks of phase i (SGP_T_SWEEP: whole sweep; SGP_T_GRAM / SGP_T_SYRK:
 * first-block-in / last-block-out of the K_uf and streaming-SYRK kernels; SGP_T_LOCAL: sweep begin .. statistics
 * assembled; SGP_T_FINISH1: Lambda formed .. Uv written; SGP_T_FINISH2: traces .. scalars; SGP_T_GAP_LOCAL_FINISH:
 * idle time on the main stream between LOCAL and FINISH1 -- launch latency plus, multi-GPU, the all-reduce). */
enum { SGP_T_SWEEP = 0, SGP_T_GRAM = 1, SGP_T_SYRK = 2, SGP_T_FINISH1 = 3, SGP_T_FINISH2 = 4,
       SGP_T_GAP_LOCAL_FINISH = 5, SGP_T_KUU = 6, SGP_T_LOCAL = 7, SGP_T_COUNT = 8 };
int sgp_get_timestamps(sgp_handle* h, int64_t* out /* 2*SGP_T_COUNT */);
/* Running totals of the per-sweep phase durations (same slots, 100 MHz ticks) over all sweeps since the last reset, and
 * the number of sweeps counted: the per-launch averages of the kernels INSIDE the timed sweeps. */
int sgp_get_phase_totals(sgp_handle* h, int64_t* totals /* SGP_T_COUNT */, int64_t* count, int32_t reset);
/* Diagnostics of the persistent factorisation launch (csrc/sgp_chain.hip.h), recorded when the handle was created with the
 * environment variable SGP_CHAIN_TRACE set: 12 steps x 32 ticks (100 MHz) during the last sweep.  Critical workgroup: [0] step
 * begins, [1] last pivot run of the diagonal tile done, [5] lower tile awaited, [2] lower tile arrived, [3] its last block
 * solved, [4] its share of the next diagonal tile's update applied, [6] / [7] waves 0 / 1 done with theirs, [8] far part of
 * the next diagonal tile received.  Feeder whose shipment this step consumes: [13] accumulation done, [9] last block solved,
 * [14] last product slice awaited, [10] ... done, [11] / [12] shipment stored by waves 0 / 4.
 * [16 + 4 cb ..]: that feeder's wave 0 at column block cb: block in registers, solved, published, updates applied.
 * which: 0 = K_uu chain, 1 = Lambda chain. */
/* diagnostics of the Cholesky step kernel (all zeros unless the library was built with -DSGP_STEP_TRACE): out[512],
 * 100 MHz stamps of one panel workgroup of the Lambda chain (the owner of tile (j + 1, j), or the block chosen with
 * -DSGP_STEP_TRACE_A=a); slot 64 j + 32 g + e = event e of wave group g (0 factoring, 1 solve) in step j < 8.  Events: see
 * tools/step_trace.py. */
int sgp_get_step_trace(int64_t* out /* 512 */);
/* diagnostics of the variant library built with -DSGP_SWEEP_TRACE (all zeros otherwise): out[65 s] = begin, out[65 s + 1 .. 65 s + 64]
 * = exit ticks (100 MHz; take the maximum) of trace slot s of the last sweep, 256 slots: 0 k_prep_xu, 2 k_gram_uf, 16 + j step j of the
 * Lambda chain, 40 + j of the K_uu chain, 64 + first tile of a SYRK launch (k_syrk_direct / k_syrk_stream), 128 + first tile row of a k_assemble launch,
 * 200 + j the moment step j had its statistics, ... (csrc/sgp_kernels.hip.h, g_sweep_trace; tools/sweep_trace.py prints them). */
int sgp_get_sweep_trace(int64_t* out /* 256 * 65 */);
int sgp_get_chain_trace(sgp_handle* h, int32_t which, int64_t* out /* 384 */);
/* HIP-event timing of one data-sized kernel (which = SGP_T_GRAM or SGP_T_SYRK) launched eagerly `iters` times on
 * `stream` with the resident data of the last sweep; returns the average launch duration in microseconds. */
int sgp_time_kernel(sgp_handle* h, int32_t which, int32_t iters, void* stream, double* avg_us);
/* ... which = SGP_TIME_GROUP0 + g: the SYRK launch of statistics group g of the overlapped sweep (sgp_overlap_plan), on the stream
 * and the compute units it uses inside the sweep (`stream` is ignored). */
#define SGP_TIME_GROUP0 100
/* ... which = SGP_TIME_QUADFORM: the per-point quadratic-form kernel of sgp_w_stats -- |L^-1 k_n|^2 (the Q_ff diagonal of
 * GPnode/UniSGPnode.jl:205-212), |Uv k_n|^2 (:214) and k_n . mu in ONE pass over the resident K_uf: 2 n M (M + 64) flop per launch
 * on the matrix cores. */
#define SGP_TIME_QUADFORM 120
/* The overlapped sweep.  For UniSGP problems whose SYRK fills the chip, sgp_sweep(h, NULL) without an all-reduce hook produces the
 * statistics in groups of tile rows of Psi2 -- group 0 on all compute units, the others on a CU-masked queue that leaves 2 CUs
 * per shader engine to the factorisation chains -- and starts the Lambda chain (GPnode/UniSGPnode.jl:62-71: the N-fold product is
 * a sum, so its Cholesky can begin on the tile columns that are complete) while the later groups are still being summed.
 * Results are those of the plain order up to rounding (the tiles collect their rank-64 updates in a different order) and are
 * bitwise reproducible.  sgp_overlap_plan reports what the next sgp_sweep will do: *ngroups = 0 (plain order) or the number
 * of groups with, per group, info[8 g ..] = {first, past-the-last tile column of P Lambda P, lower tiles, point chunks, points
 * per chunk, masked (0/1), CUs available, the Lambda-chain step that forms the group}.  Environment: SGP_OVERLAP=0 turns it
 * off, SGP_OVERLAP=1 forces it wherever it is possible, SGP_OVERLAP_COLS="3" / "2,4" sets the group boundaries.
 * "Fills the chip" = points x lower tiles >= 10 000 (SGP_GATE_MIN overrides): from there on the SYRK is k_syrk_direct (one
 * workgroup per CU, no LDS staging; SGP_SYRK_WIDE=0: the LDS-staged k_syrk_stream everywhere) and the K_uu chain is held back
 * until its single round is resident.  The planner places one cut, or two from six tile columns on while the masked launches
 * are short; a data-sharded sweep (hook installed) keeps one cut -- every group is a collective.
 * Host order of the launches: sgp_sweep on the library's streams enqueues the launches of the K_uu chain and of the Lambda chain
 * alternately -- a sweep that starts on an idle device (the first of a block, every sweep of a caller that fetches something in
 * between) then does not have its Lambda chain wait for the host to get through the other chain's 14 launches; once the host is a
 * sweep ahead the order makes no difference (SGP_INTERLEAVE=0: chain after chain).  Same kernels, same results either way. */
int sgp_overlap_plan(const sgp_handle* h, int32_t* ngroups, int32_t* info /* 8 per group, up to 8 groups; may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* SGP_HIP_H */
