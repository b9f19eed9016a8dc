# SGPHip.jl -- the binding a GaussianProcessNode maintainer would add to route the UniSGP / MultiSGP hot path through
# libsgp_hip.so (include/sgp_hip.h).  WRITTEN BLIND: Julia is not available in the build pipeline, so this file has never
# been executed.  It mirrors, call for call, the Python host mirror (gaussianprocessnode_amd/unisgp.py, multisgp.py),
# which IS tested against the reference's rule tests on the GPU.
#
# The reference's rule bodies are not edited.  The only change in user code is the meta constructor:
#
#     include("GPnode/UniSGPnode.jl"); include("GPnode/MultiSGPnode.jl")
#     include("SGPHip.jl"); using .SGPHip
#     @meta function meta_gp_regression(Xu, Ψ0, Ψ1_trans, Ψ2, KuuL, kernel, Uv)
#         UniSGP() -> HipSGPMeta(UniSGPMeta(nothing, Xu, Ψ0, Ψ1_trans, Ψ2, KuuL, kernel, Uv, 0, batch_size);
#                                kernel_params = θ -> (softplus(θ[1]), softplus.(θ[2:end])))
#     end
#
# `HipSGPMeta` wraps the reference's meta and owns a device handle.  The methods below dispatch on it: the PointMass-input
# rules (the hot path: `:v`, the N-fold product, `:w`, the average energy, `:out`) run on the device; every other rule
# (uncertain inputs, `:in`, `:θ`) forwards to the reference's own method with the wrapped `UniSGPMeta`, whose `Uv` and
# `counter` the device path keeps up to date exactly as GPnode/UniSGPnode.jl:64-71 does.  `HipMultiSGPMeta` does the same
# for the MultiSGP node: the cubature Ψ-statistics of a step (GPnode/MultiSGPnode.jl:11-35, 5 Gram columns and 5 rank-1
# M × M updates per step in the reference) come from the device, the D × D algebra around them stays as the reference has it.
module SGPHip

using ReactiveMP, LinearAlgebra
import ReactiveMP: @rule, @average_energy, @call_rule, GenericProd, PointMass, MvNormalMeanCovariance,
                   MvNormalMeanPrecision, MvNormalWeightedMeanPrecision, NormalMeanPrecision, GammaShapeRate, Wishart,
                   NormalDistributionsFamily, UnivariateGaussianDistributionsFamily, MultivariateNormalDistributionsFamily,
                   MultivariateGaussianDistributionsFamily, mean, var, cov, mean_cov, AverageEnergy
import ..UniSGP, ..UniSGPMeta, ..MultiSGP, ..MultiSGPMeta, ..WishartFast, ..approximate_kernel_expectation!

export HipSGPMeta, HipMultiSGPMeta, predict, theta_objective, carry_posterior!, multisgp_sweep!

const LIB = get(ENV, "SGP_HIP_LIB", "libsgp_hip.so")

# ------------------------------------------------------------------------------------------------------------------
# the C ABI (include/sgp_hip.h)
# ------------------------------------------------------------------------------------------------------------------
struct SGPConfig
    n_max::Int64; m::Int32; d::Int32; d_out::Int32; device::Int32; flags::Int32; reserved::Int32
end
const SGP_FLAG_KEEP_KUF = Int32(2)
const SGP_S_COUNT = 8
const SGP_R_COUNT = 8

function check(rc::Cint, h)
    rc == 0 && return
    msg = unsafe_string(ccall((:sgp_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    rc > 0 ? throw(PosDefException(rc)) : error("libsgp_hip: status $rc: $msg")     # fastcholesky's failure mode
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    m::Int; d::Int; d_out::Int; n_max::Int
end

function Handle(n_max, Xu::Matrix{Float64}, d_out; device = 0, flags = SGP_FLAG_KEEP_KUF)
    D, M = size(Xu)                                            # D × M column-major = M points of D doubles: the ABI layout
    cfg = Ref(SGPConfig(n_max, M, D, d_out, device, flags, 0))
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:sgp_create, LIB), Cint, (Ref{SGPConfig}, Ref{Ptr{Cvoid}}), cfg, p), C_NULL)
    h = Handle(p[], M, D, d_out, n_max)
    check(ccall((:sgp_set_inducing, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, Xu), h.ptr)
    finalizer(x -> ccall((:sgp_destroy, LIB), Cint, (Ptr{Cvoid},), x.ptr), h)
    return h
end

inducing_matrix(Xu) = Xu[1] isa Number ? reshape(collect(Float64, Xu), 1, :) : reduce(hcat, [collect(Float64, u) for u in Xu])

set_data!(h::Handle, X::Matrix{Float64}, y, yv, ω, n_nodes) =
    check(ccall((:sgp_set_data, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64),
                h.ptr, X, y, yv === nothing ? C_NULL : yv, ω === nothing ? C_NULL : ω, size(X, 2), n_nodes), h.ptr)
set_kernel!(h::Handle, σ², ℓ::Vector{Float64}, jitter) =
    check(ccall((:sgp_set_kernel, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Int32, Float64), h.ptr, σ², ℓ, length(ℓ), jitter), h.ptr)
set_noise!(h::Handle, W::Matrix{Float64}, ElogW) =
    check(ccall((:sgp_set_noise, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64), h.ptr, W, ElogW), h.ptr)
set_prior!(h::Handle, vec, mat::Matrix{Float64}, form) =
    check(ccall((:sgp_set_prior, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32), h.ptr, vec, mat, form), h.ptr)
sweep!(h::Handle) = check(ccall((:sgp_sweep, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, C_NULL), h.ptr)
sweep_local!(h::Handle) = check(ccall((:sgp_sweep_local, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, C_NULL), h.ptr)

function posterior(h::Handle)
    Q = h.m * h.d_out
    μ = zeros(Q); Σ = zeros(Q, Q); Uv = zeros(Q, Q)
    check(ccall((:sgp_get_posterior, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h.ptr, μ, Σ, Uv), h.ptr)
    return μ, Σ, Uv
end
function stats(h::Handle)
    Ψ2 = zeros(h.m, h.m); B = zeros(h.m, h.d_out); sc = zeros(SGP_S_COUNT)
    check(ccall((:sgp_get_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h.ptr, Ψ2, B, sc), h.ptr)
    return Ψ2, B, sc
end
function scalars(h::Handle)
    out = zeros(SGP_R_COUNT)
    check(ccall((:sgp_get_scalars, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, out), h.ptr)
    return out                                                 # [ΣI1, ΣI2, energy, info Kuu, info Λ, info prior, logdet Kuu, logdet Λ]
end
function w_stats(h::Handle, n)
    I1 = zeros(n); I2 = zeros(n)
    check(ccall((:sgp_w_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), h.ptr, I1, I2, C_NULL), h.ptr)
    return I1, I2
end
function wishart_invscale(h::Handle)
    S = zeros(h.d_out, h.d_out)
    check(ccall((:sgp_get_wishart_invscale, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, S), h.ptr)
    return S
end
function predict_mean(h::Handle, Xstar::Matrix{Float64}, μ_v::Vector{Float64})
    out = zeros(size(Xstar, 2), h.d_out)
    check(ccall((:sgp_predict, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}), h.ptr, Xstar, size(Xstar, 2), μ_v, out), h.ptr)
    return out
end

# ------------------------------------------------------------------------------------------------------------------
# UniSGP
# ------------------------------------------------------------------------------------------------------------------
mutable struct HipSGPMeta{R<:UniSGPMeta}
    ref::R                            # the reference's meta: Xu, kernel, N, and the `Uv` / `counter` fields the hook maintains
    handle::Handle
    kernel_params::Function           # θ -> (σ², ℓ::Vector{Float64}) of `ref.kernel`
    jitter::Float64
    xs::Vector{Vector{Float64}}       # points of the batch being folded
    ys::Vector{Float64}
    vs::Vector{Float64}
    prior::Any
    w::Float64
    Elogw::Float64
    θ::Vector{Float64}
    index::Dict{Vector{Float64},Int}  # point -> column of the last swept batch
    I1::Vector{Float64}               # per-point :w quantities of the last sweep (fetched on first use)
    I2::Vector{Float64}
end

function HipSGPMeta(ref::UniSGPMeta; kernel_params, jitter = 0.0, device = 0)
    h = Handle(ref.N, inducing_matrix(ref.Xu), 1; device = device)
    return HipSGPMeta(ref, h, kernel_params, Float64(jitter), Vector{Float64}[], Float64[], Float64[], nothing, 1.0, 0.0,
                      Float64[], Dict{Vector{Float64},Int}(), Float64[], Float64[])
end

# the token the per-point :v rule returns instead of an M × M message (replaces BufferUniSGP, GPnode/UniSGPnode.jl:56-60)
struct HipBuffer{T<:HipSGPMeta}
    index::Int
    meta::T
end

point(q) = collect(Float64, mean(q) isa Number ? [mean(q)] : mean(q))
elog(q_w) = q_w isa GammaShapeRate ? mean(log, q_w) : log(mean(q_w))

# ---- :v  (GPnode/UniSGPnode.jl:144-158 PointMass output, :161-173 Gaussian output): O(1) token -------------------
function push_point!(meta::HipSGPMeta, q_out, q_in, q_w, q_θ)
    push!(meta.xs, point(q_in)); push!(meta.ys, mean(q_out)); push!(meta.vs, q_out isa PointMass ? 0.0 : var(q_out))
    meta.w = mean(q_w); meta.Elogw = elog(q_w); meta.θ = collect(Float64, mean(q_θ))
    return HipBuffer(length(meta.xs), meta)
end
@rule UniSGP(:v, Marginalisation) (q_out::PointMass, q_in::PointMass, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) =
    push_point!(meta, q_out, q_in, q_w, q_θ)
@rule UniSGP(:v, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) =
    push_point!(meta, q_out, q_in, q_w, q_θ)

# ---- the N-fold product (GPnode/UniSGPnode.jl:62-73): one device sweep when counter == N -------------------------
function ReactiveMP.prod(::GenericProd, left::NormalDistributionsFamily, right::HipBuffer)
    meta = right.meta; ref = meta.ref
    ref.counter += 1                                           # :63-64
    ref.counter == 1 && (meta.prior = left)
    ref.counter < ref.N && return meta.prior                   # nothing consumes the partial products
    h = meta.handle
    X = reduce(hcat, meta.xs)
    set_data!(h, X, meta.ys, any(!iszero, meta.vs) ? meta.vs : nothing, nothing, -1.0)
    σ², ℓ = meta.kernel_params(meta.θ)
    set_kernel!(h, σ², collect(Float64, ℓ), meta.jitter)
    set_noise!(h, fill(meta.w, 1, 1), meta.Elogw)
    μ0, Σ0 = mean_cov(meta.prior)
    set_prior!(h, collect(Float64, μ0), Matrix{Float64}(Σ0), 0)
    sweep!(h)
    μ, Σ, Uv = posterior(h)
    ref.Uv = UpperTriangular(Uv)                               # :67-69  meta.Uv = chol(Σ + μμ').U
    ref.counter = 0                                            # :70
    meta.index = Dict(x => i for (i, x) in enumerate(meta.xs)); meta.I1 = Float64[]; meta.I2 = Float64[]
    empty!(meta.xs); empty!(meta.ys); empty!(meta.vs)
    return MvNormalMeanCovariance(μ, Σ)
end

# ---- :w and the average energy at PointMass inputs (GPnode/UniSGPnode.jl:196-238, 337-387, 411-436) ---------------
function point_stats(q_in::PointMass, meta::HipSGPMeta)
    isempty(meta.I1) && ((meta.I1, meta.I2) = w_stats(meta.handle, length(meta.index)))
    i = meta.index[point(q_in)]
    return meta.I1[i], meta.I2[i]
end
@rule UniSGP(:w, Marginalisation) (q_out::PointMass, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipSGPMeta) =
    (I = point_stats(q_in, meta); GammaShapeRate(1.5, 0.5 * (I[1] + I[2])))
@rule UniSGP(:w, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipSGPMeta) =
    (I = point_stats(q_in, meta); GammaShapeRate(1.5, 0.5 * (I[1] + I[2])))

hip_energy(q_in, q_w, meta) = (I = point_stats(q_in, meta); w = mean(q_w); 0.5 * (I[1] * w - elog(q_w) + log(2π) + I[2] * w))
@average_energy UniSGP (q_out::PointMass, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::GammaShapeRate, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_in, q_w, meta)
@average_energy UniSGP (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::GammaShapeRate, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_in, q_w, meta)
@average_energy UniSGP (q_out::PointMass, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::PointMass, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_in, q_w, meta)
@average_energy UniSGP (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::PointMass, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_in, q_w, meta)

# ---- :out at a PointMass input (GPnode/UniSGPnode.jl:96-104); `predict` does a whole test set in one call ---------
@rule UniSGP(:out, Marginalisation) (q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) = begin
    m = predict(meta, reshape(point(q_in), :, 1), collect(Float64, mean(q_v)), mean(q_θ))
    return NormalMeanPrecision(m[1], mean(q_w))
end
function predict(meta::HipSGPMeta, Xstar::Matrix{Float64}, μ_v::Vector{Float64}, θ)      # experiments/regression_kin40k.ipynb:288-304
    σ², ℓ = meta.kernel_params(θ)
    set_kernel!(meta.handle, σ², collect(Float64, ℓ), meta.jitter)
    return vec(predict_mean(meta.handle, Xstar, μ_v))
end

# ---- everything else: the reference's own methods, with the wrapped meta -----------------------------------------
@rule UniSGP(:out, Marginalisation) (q_in::UnivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) =
    @call_rule UniSGP(:out, Marginalisation) (q_in = q_in, q_v = q_v, q_w = q_w, q_θ = q_θ, meta = meta.ref)
@rule UniSGP(:in, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) =
    @call_rule UniSGP(:in, Marginalisation) (q_out = q_out, q_v = q_v, q_w = q_w, q_θ = q_θ, meta = meta.ref)
@rule UniSGP(:v, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_in::UnivariateGaussianDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) =
    @call_rule UniSGP(:v, Marginalisation) (q_out = q_out, q_in = q_in, q_w = q_w, q_θ = q_θ, meta = meta.ref)
@rule UniSGP(:w, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_in::UnivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipSGPMeta) =
    @call_rule UniSGP(:w, Marginalisation) (q_out = q_out, q_in = q_in, q_v = q_v, q_θ = q_θ, meta = meta.ref)
@rule UniSGP(:θ, Marginalisation) (q_out::Any, q_in::Any, q_v::MultivariateNormalDistributionsFamily, q_w::Any, meta::HipSGPMeta) =
    @call_rule UniSGP(:θ, Marginalisation) (q_out = q_out, q_in = q_in, q_v = q_v, q_w = q_w, meta = meta.ref)

# ---- hyper-parameter objective and gradient (neg_log_backwardmess_fast / grad_llh_new!,
# helper_functions/derivative_helper.jl:23-39,55-63) at the θ of the last sweep, q(v) fixed ---------------------------
# Returns (F, dF/d(σ², ℓ...)); the caller applies the chain rule of its kernel_gp(θ) (softplus in the notebooks).
function theta_objective(meta::HipSGPMeta, nparams::Int)
    f = Ref{Float64}(0.0); g = zeros(nparams)
    check(ccall((:sgp_theta_objective, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}), meta.handle.ptr, f, g), meta.handle.ptr)
    return f[], g
end

# ---- minibatch loops (experiments/regression_kin40k.ipynb:205-212): prior <- posterior without leaving the device --
carry_posterior!(meta::HipSGPMeta) =
    check(ccall((:sgp_carry_posterior, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), meta.handle.ptr, C_NULL), meta.handle.ptr)

# ---- the whole PerformInference loop on the device (experiments/regression_kin40k.ipynb:196-230; sgp_train_*) -------
# X is D × N (one point per column), θ the raw (pre-softplus) parameters of kernel_gp.  Returns θ after the last step.
function train!(meta::HipSGPMeta, X::Matrix{Float64}, y::Vector{Float64}, θ::Vector{Float64}; batch, epochs, w, prior_var = 50.0,
                η = 1e-3, β = (0.9, 0.999), ϵ = 1e-8)
    h = meta.handle; N = length(y)
    set_noise!(h, fill(Float64(w), 1, 1), log(w))
    set_prior!(h, C_NULL, fill(Float64(prior_var), 1, 1), 2)
    check(ccall((:sgp_train_begin, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Int32, Float64, Float64, Float64, Float64, Float64),
                h.ptr, X, y, N, θ, length(θ) - 1, meta.jitter, η, β[1], β[2], ϵ), h.ptr)
    for _ in 1:epochs, o in 0:batch:N-1
        flags = Int32(1) | (o == 0 ? Int32(2) : Int32(0))       # SGP_TRAIN_LEARN | SGP_TRAIN_RESET_PRIOR (:203-204)
        check(ccall((:sgp_train_step, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Int32), h.ptr, o, min(batch, N - o), flags), h.ptr)
    end
    θout = similar(θ); counts = zeros(Int64, 2)
    check(ccall((:sgp_train_end, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), h.ptr, θout, counts), h.ptr)
    counts[2] == 0 || throw(PosDefException(Int(counts[2])))
    return θout
end

# The classification loop (experiments/classification_banana.ipynb cell 9: `y[i] ~ Probit(f[i])`, q(w) = GammaShapeRate carried
# over the minibatches, q(v) never reset): labels 0 / 1 in y; returns (θ, shape, rate).  sgp_train_likelihood turns the run
# opened by sgp_train_begin into this loop -- forward message, Probit moment matching, Gamma update and AdaMax on the device.
function train_classification!(meta::HipSGPMeta, X::Matrix{Float64}, y::Vector{Float64}, θ::Vector{Float64}; batch, epochs,
                               shape = 0.01, rate = 0.01, prior_var = 50.0, η = 1e-3, β = (0.9, 0.999), ϵ = 1e-8)
    h = meta.handle; N = length(y)
    set_prior!(h, C_NULL, fill(Float64(prior_var), 1, 1), 2)
    check(ccall((:sgp_train_begin, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Int32, Float64, Float64, Float64, Float64, Float64),
                h.ptr, X, y, N, θ, length(θ) - 1, meta.jitter, η, β[1], β[2], ϵ), h.ptr)
    check(ccall((:sgp_train_likelihood, LIB), Cint, (Ptr{Cvoid}, Int32, Float64, Float64), h.ptr, Int32(1), shape, rate), h.ptr)   # SGP_LIKELIHOOD_PROBIT
    for _ in 1:epochs, o in 0:batch:N-1
        check(ccall((:sgp_train_step, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Int32), h.ptr, o, min(batch, N - o), Int32(1)), h.ptr)
    end
    θout = similar(θ); counts = zeros(Int64, 2); ab = zeros(2)
    check(ccall((:sgp_train_end, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), h.ptr, θout, counts), h.ptr)
    check(ccall((:sgp_train_get_gamma, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, ab), h.ptr)
    counts[2] == 0 || throw(PosDefException(Int(counts[2])))
    return θout, ab[1], ab[2]
end

# Multi-GPU (one Julia process per GPU, e.g. under MPI.jl): register the communicator once; sgp_sweep / sgp_theta_objective /
# sgp_train_step then sum what has to be summed inside the library -- the exchange buffer [lower tiles of Ψ2 | B | scalars] and
# the data half of the θ gradient -- and every rank sees the statistics, the objective and the gradient of all shards.
use_rccl!(h::Handle, comm::Ptr{Cvoid}) = check(ccall((:sgp_use_rccl, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, comm), h.ptr)

# ------------------------------------------------------------------------------------------------------------------
# MultiSGP
# ------------------------------------------------------------------------------------------------------------------
mutable struct HipMultiSGPMeta{R<:MultiSGPMeta}
    ref::R
    d_out::Int
    handle::Handle                    # d_out outputs: the batched sweep (multisgp_sweep!)
    step::Handle                      # single-output handle for the per-step Ψ-statistics
    kernel_params::Function
    jitter::Float64
end

function HipMultiSGPMeta(ref::MultiSGPMeta, d_out::Int; kernel_params, n_steps, jitter = 0.0, device = 0)
    Xu = inducing_matrix(ref.Xu)
    npts = 2 * size(Xu, 1) + 1                                 # srcubature: 2 d_in + 1 points per step
    return HipMultiSGPMeta(ref, d_out, Handle(n_steps * npts, Xu, d_out; device = device), Handle(npts, Xu, 1; device = device),
                           kernel_params, Float64(jitter))
end

# cubature points / weights of q_in as the reference's approximate_kernel_expectation! walks them (GPnode/MultiSGPnode.jl:11-35)
function cubature(meta::HipMultiSGPMeta, q_in)
    m, P = mean_cov(q_in)
    weights = ReactiveMP.getweights(meta.ref.method, m, P)
    points = ReactiveMP.getpoints(meta.ref.method, m, P)
    return reduce(hcat, [collect(Float64, p) for p in points]), collect(Float64, weights)
end

# Ψ0, Ψ1, Ψ2 of ONE step on the device: the step's cubature points as weighted data, statistics without a posterior
function psi_statistics!(meta::HipMultiSGPMeta, q_in, θ)
    X, ω = cubature(meta, q_in)
    h = meta.step
    set_data!(h, X, ones(length(ω)), nothing, ω, 1.0)
    σ², ℓ = meta.kernel_params(θ)
    set_kernel!(h, σ², collect(Float64, ℓ), meta.jitter)
    sweep_local!(h)
    Ψ2, B, sc = stats(h)
    ref = meta.ref
    ref.Ψ0 .= σ² * sc[2]; ref.Ψ1_trans .= B; ref.Ψ2 .= Ψ2     # the buffers the reference's rules fill (:299-302)
    return ref.Ψ0, ref.Ψ1_trans, ref.Ψ2
end

# ---- :v for one step (GPnode/MultiSGPnode.jl:290-328): ξ = vcat(Ψ1 (μ_y' W)_d), Λ = kron(W, Ψ2) ---------------------
function hip_multi_v(q_out, q_in, q_w, q_θ, meta::HipMultiSGPMeta)
    W = mean(q_w); μ_y = mean(q_out)
    _, Ψ1, Ψ2 = psi_statistics!(meta, q_in, mean(q_θ))
    row = μ_y' * W
    return MvNormalWeightedMeanPrecision(vcat([vec(Ψ1) .* row[d] for d in 1:length(μ_y)]...), kron(W, Ψ2))
end
@rule MultiSGP(:v, Marginalisation) (q_out::MultivariateGaussianDistributionsFamily, q_in::MultivariateGaussianDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) =
    hip_multi_v(q_out, q_in, q_w, q_θ, meta)
@rule MultiSGP(:v, Marginalisation) (q_out::PointMass, q_in::MultivariateGaussianDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) =
    hip_multi_v(q_out, q_in, q_w, q_θ, meta)

# ---- :w for one step (GPnode/MultiSGPnode.jl:367-444): WishartFast(D + 2, I1 + I2) ---------------------------------
# I1 = (Ψ0 - tr(Kuu^-1 Ψ2)) I,  I2[i,j] = Σ_y[i,j] + μ_y μ_y' - μ_y E' - E μ_y' + tr(Rv_blk[i][j] Ψ2),  E_d = Ψ1 · μ_v^(d)
function hip_multi_w(q_out, q_in, q_v, q_θ, meta::HipMultiSGPMeta)
    Ψ0, Ψ1, Ψ2 = psi_statistics!(meta, q_in, mean(q_θ))
    μ_y = mean(q_out); D = length(μ_y); M = meta.step.m
    Σ_y = q_out isa PointMass ? zeros(D, D) : cov(q_out)
    μ_v, Σ_v = mean_cov(q_v)
    Rv = Σ_v + μ_v * μ_v'
    E = [dot(vec(Ψ1), view(μ_v, (d-1)*M+1:d*M)) for d in 1:D]
    Ψ4 = [sum(view(Rv, (i-1)*M+1:i*M, (j-1)*M+1:j*M) .* Ψ2') for i in 1:D, j in 1:D]
    I1 = (Ψ0[1] - tr(meta.ref.Kuu_inverse * Ψ2)) * Matrix(I, D, D)
    tmp = μ_y * E'
    return WishartFast(D + 2, Ψ4 + Σ_y + μ_y * μ_y' - tmp - tmp' + I1)
end
@rule MultiSGP(:w, Marginalisation) (q_out::MultivariateNormalDistributionsFamily, q_in::MultivariateNormalDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipMultiSGPMeta) =
    hip_multi_w(q_out, q_in, q_v, q_θ, meta)
@rule MultiSGP(:w, Marginalisation) (q_out::PointMass, q_in::MultivariateNormalDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipMultiSGPMeta) =
    hip_multi_w(q_out, q_in, q_v, q_θ, meta)

# ---- :out for one step (GPnode/MultiSGPnode.jl:90-120): mean_d = Ψ1 · μ_v^(d), precision mean(q_w) -----------------
@rule MultiSGP(:out, Marginalisation) (q_in::MultivariateNormalDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) = begin
    X, ω = cubature(meta, q_in)
    σ², ℓ = meta.kernel_params(mean(q_θ))
    set_kernel!(meta.handle, σ², collect(Float64, ℓ), meta.jitter)
    F = predict_mean(meta.handle, X, collect(Float64, mean(q_v)))            # points × d_out
    return MvNormalMeanPrecision(vec(ω' * F), mean(q_w))
end

# ---- :in, :θ and the per-step energies: the reference's own methods with the wrapped meta ---------------------------
@rule MultiSGP(:in, Marginalisation) (q_out::Any, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) =
    @call_rule MultiSGP(:in, Marginalisation) (q_out = q_out, q_v = q_v, q_w = q_w, q_θ = q_θ, meta = meta.ref)
@rule MultiSGP(:in, Marginalisation) (q_out::PointMass, q_in::MultivariateGaussianDistributionsFamily, q_v::MultivariateGaussianDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) =
    @call_rule MultiSGP(:in, Marginalisation) (q_out = q_out, q_in = q_in, q_v = q_v, q_w = q_w, q_θ = q_θ, meta = meta.ref)
@rule MultiSGP(:θ, Marginalisation) (q_out::Any, q_in::MultivariateGaussianDistributionsFamily, q_v::MultivariateGaussianDistributionsFamily, q_w::Any, meta::HipMultiSGPMeta) =
    @call_rule MultiSGP(:θ, Marginalisation) (q_out = q_out, q_in = q_in, q_v = q_v, q_w = q_w, meta = meta.ref)
@average_energy MultiSGP (q_out::Any, q_in::MultivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) =
    ReactiveMP.score(AverageEnergy(), MultiSGP, Val{(:out, :in, :v, :w, :θ)}(), (q_out, q_in, q_v, q_w, q_θ), meta.ref)

# ---- the batched form: ALL steps in one sweep (what gaussianprocessnode_amd.multisgp.sweep does) -------------------
# q(v) from every step's :v message folded with `prior`; also returns Σ_t (I1_t + I2_t) (the Wishart inverse scales add,
# GPnode/MultiSGPnode.jl:367-444) and the summed average energy (:544-632), so a smoother's M-step is one call.
function multisgp_sweep!(meta::HipMultiSGPMeta, q_outs, q_ins, q_w, q_θ::PointMass, prior)
    h = meta.handle; D = meta.d_out
    Xs = Matrix{Float64}[]; ωs = Vector{Float64}[]; Ys = Matrix{Float64}[]; Σsum = zeros(D, D)
    for (q_in, q_out) in zip(q_ins, q_outs)
        X, ω = cubature(meta, q_in)
        push!(Xs, X); push!(ωs, ω); push!(Ys, repeat(collect(Float64, mean(q_out))', length(ω)))
        q_out isa PointMass || (Σsum .+= cov(q_out))
    end
    X = reduce(hcat, Xs); ω = reduce(vcat, ωs); Y = reduce(vcat, Ys)            # Y: points × d_out, column-major = per-output blocks
    set_data!(h, X, vec(Y), nothing, ω, Float64(length(q_ins)))
    any(!iszero, Σsum) && check(ccall((:sgp_set_output_cov_sum, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, Σsum), h.ptr)
    σ², ℓ = meta.kernel_params(mean(q_θ))
    set_kernel!(h, σ², collect(Float64, ℓ), meta.jitter)
    W = Matrix{Float64}(mean(q_w))
    set_noise!(h, W, q_w isa Union{Wishart,WishartFast} ? mean(logdet, q_w) : logdet(W))
    μ0, Σ0 = mean_cov(prior)
    set_prior!(h, collect(Float64, μ0), Matrix{Float64}(Σ0), 0)
    sweep!(h)
    μ, Σ, _ = posterior(h)
    return MvNormalMeanCovariance(μ, Σ), wishart_invscale(h), scalars(h)[3]
end

end # module
