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
# `HipSGPMeta` wraps the reference's meta and owns a device handle.  The methods below dispatch on it.  The PointMass-input
# rules (the hot path: `:v`, the N-fold product, `:w`, the average energy, `:out`) run on the device inside the sweep.  The
# cold rules run on the device too, call for call like gaussianprocessnode_amd/unisgp.py:183-387: uncertain inputs enter `:v`
# and `:out` as cubature-weighted data, and the per-node clamped `:w` / average energy of an uncertain input, the `:in`
# closure and the three `:θ` closures are stand-alone per-point evaluations (`stats_at`: sgp_set_data + sgp_sweep_local +
# sgp_set_posterior + sgp_w_stats on an auxiliary handle) at whatever q_v / meta.Uv the caller passes -- the wrapped
# `UniSGPMeta`'s `Uv`, `KuuL` and `counter` are kept up to date exactly as GPnode/UniSGPnode.jl:64-71 does.
# `HipMultiSGPMeta` does the same for the MultiSGP node: the cubature Ψ-statistics of a step (GPnode/MultiSGPnode.jl:11-35,
# 5 Gram columns and 5 rank-1 M × M updates per step in the reference) come from the device, and so do the `:in` and `:θ`
# closures (multisgp.py:169-278); the D × D algebra around them stays as the reference has it.
module SGPHip

using ReactiveMP, LinearAlgebra
import Optim                                   # (the reference's MultiSGP(:in) Laplace rule uses it, GPnode/MultiSGPnode.jl:229)
import ReactiveMP: @rule, @average_energy, @call_rule, GenericProd, PointMass, MvNormalMeanCovariance,
                   MvNormalMeanPrecision, MvNormalWeightedMeanPrecision, NormalMeanPrecision, GammaShapeRate, Wishart,
                   NormalDistributionsFamily, UnivariateGaussianDistributionsFamily, MultivariateNormalDistributionsFamily,
                   MultivariateGaussianDistributionsFamily, UnivariateNormalDistributionsFamily, mean, var, cov, mean_cov,
                   mean_var, AverageEnergy, ContinuousUnivariateLogPdf, ContinuousMultivariateLogPdf, UnspecifiedDomain
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
wait!(h::Handle) = check(ccall((:sgp_wait, LIB), Cint, (Ptr{Cvoid},), h.ptr), h.ptr)      # polled drain of the handle's streams

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

# q(v) installed from outside for the per-point outputs (include/sgp_hip.h, sgp_set_posterior): mean and the upper factor
# Uv = chol(Σ_v + μ μ').U, column-major Q × Q -- a Julia Matrix as it is
set_posterior!(h::Handle, μ_v::Vector{Float64}, Uv::Matrix{Float64}) =
    check(ccall((:sgp_set_posterior, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h.ptr, μ_v, Uv), h.ptr)
function kuu_chol(h::Handle)
    L = zeros(h.m, h.m)
    check(ccall((:sgp_get_kuu_chol, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, L), h.ptr)
    return L
end
# stand-alone dense building blocks on the device (host in / host out): lower Cholesky factor, SPD inverse, K(A, B)
function potrf(A::Matrix{Float64}; device = 0)
    n = size(A, 1); L = zeros(n, n)
    check(ccall((:sgp_potrf, LIB), Cint, (Int32, Ptr{Float64}, Int32, Ptr{Float64}), device, A, n, L), C_NULL)
    return L
end
function potri(A::Matrix{Float64}; device = 0)
    n = size(A, 1); Ai = zeros(n, n)
    check(ccall((:sgp_potri, LIB), Cint, (Int32, Ptr{Float64}, Int32, Ptr{Float64}), device, A, n, Ai), C_NULL)
    return Ai
end
function kernelmatrix_dev(A::Matrix{Float64}, B::Matrix{Float64}, σ², ℓ::Vector{Float64}; device = 0)      # A: D × na, B: D × nb
    K = zeros(size(A, 2), size(B, 2))
    check(ccall((:sgp_kernelmatrix, LIB), Cint,
                (Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Int32, Float64, Ptr{Float64}, Int32, Ptr{Float64}),
                device, A, size(A, 2), B, size(B, 2), size(A, 1), σ², ℓ, length(ℓ), K), C_NULL)
    return K
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
    ωs::Vector{Float64}               # cubature weights of the batch being folded (uncertain inputs: one entry per cubature point)
    uncertain::Bool                   # the batch being folded consists of uncertain-input nodes (GPnode/UniSGPnode.jl:125-140)
    aux::Union{Nothing,Handle}        # second handle for stand-alone rule evaluations (it replaces data and posterior)
    device::Int
    μ_last::Vector{Float64}           # mean of the last swept q(v): the per-point pass belongs to it
end

function HipSGPMeta(ref::UniSGPMeta; kernel_params, jitter = 0.0, device = 0, cubature_points = 21)
    # (uncertain inputs enter the sweep as `cubature_points` weighted points per node: size the handle for them)
    h = Handle(ref.N * max(1, cubature_points), inducing_matrix(ref.Xu), 1; device = device)
    return HipSGPMeta(ref, h, kernel_params, Float64(jitter), Vector{Float64}[], Float64[], Float64[], nothing, 1.0, 0.0,
                      Float64[], Dict{Vector{Float64},Int}(), Float64[], Float64[], Float64[], false, nothing, device, Float64[])
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
    μ0, Σ0 = mean_cov(meta.prior)
    if meta.uncertain
        # cubature points as weighted data, N nodes (unisgp.py:138-147); every one of the N messages carries Ψ2 + 1e-8 I
        # (GPnode/UniSGPnode.jl:135,138): the prior's precision gains 1e-8 w N on the diagonal
        set_data!(h, X, meta.ys, nothing, meta.ωs, Float64(ref.N))
        Λ0 = potri(Matrix{Float64}(Σ0); device = meta.device)
        ξ0 = Λ0 * collect(Float64, μ0)
        Λ0 += 1e-8 * meta.w * ref.N * I
        set_prior!(h, ξ0, Matrix{Float64}(Λ0), 1)
    else
        set_data!(h, X, meta.ys, any(!iszero, meta.vs) ? meta.vs : nothing, nothing, -1.0)
        set_prior!(h, collect(Float64, μ0), Matrix{Float64}(Σ0), 0)
    end
    σ², ℓ = meta.kernel_params(meta.θ)
    set_kernel!(h, σ², collect(Float64, ℓ), meta.jitter)
    set_noise!(h, fill(meta.w, 1, 1), meta.Elogw)
    sweep!(h)
    μ, Σ, Uv = posterior(h)
    ref.Uv = UpperTriangular(Uv)                               # :67-69  meta.Uv = chol(Σ + μμ').U
    ref.KuuL = LowerTriangular(kuu_chol(h))                    # (the cold rules read meta.KuuL: the factor this sweep used)
    ref.counter = 0                                            # :70
    meta.index = meta.uncertain ? Dict{Vector{Float64},Int}() : Dict(x => i for (i, x) in enumerate(meta.xs))
    meta.I1 = Float64[]; meta.I2 = Float64[]; meta.μ_last = copy(μ)
    empty!(meta.xs); empty!(meta.ys); empty!(meta.vs); empty!(meta.ωs); meta.uncertain = false
    return MvNormalMeanCovariance(μ, Σ)
end

# ---- :v with an uncertain input (GPnode/UniSGPnode.jl:125-140): the node's cubature points join the batch as weighted data --
cubature_1d(meta::HipSGPMeta, q_in) = (m = mean(q_in); v = var(q_in);
    (collect(Float64, ReactiveMP.getpoints(meta.ref.method, m, v)), collect(Float64, ReactiveMP.getweights(meta.ref.method, m, v))))
@rule UniSGP(:v, Marginalisation) (q_out::UnivariateNormalDistributionsFamily, q_in::UnivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) = begin
    isempty(meta.xs) || meta.uncertain || error("PointMass and uncertain inputs mixed in one graph")
    pts, ω = cubature_1d(meta, q_in)
    for (p, wt) in zip(pts, ω)
        push!(meta.xs, [p]); push!(meta.ys, mean(q_out)); push!(meta.vs, 0.0); push!(meta.ωs, wt)
    end
    meta.uncertain = true
    meta.w = mean(q_w); meta.Elogw = elog(q_w); meta.θ = collect(Float64, mean(q_θ))
    return HipBuffer(length(meta.xs), meta)
end

# ---- :w and the average energy at PointMass inputs (GPnode/UniSGPnode.jl:196-238, 337-387, 411-436) ---------------
# (I1_n, I2_n) of one node: from the per-point pass over the last swept batch when the point and q_v belong to it, else
# evaluated stand-alone at (q_v, meta.Uv) like the reference's rule would (unisgp.py: _point_stats)
function point_stats(q_out, q_in::PointMass, q_v, q_θ, meta::HipSGPMeta)
    x = point(q_in)
    if haskey(meta.index, x) && isapprox(collect(Float64, mean(q_v)), meta.μ_last; rtol = 1e-12, atol = 0)
        isempty(meta.I1) && ((meta.I1, meta.I2) = w_stats(meta.handle, length(meta.index)))
        i = meta.index[x]
        return meta.I1[i], meta.I2[i]
    end
    yv = q_out isa PointMass ? nothing : [Float64(var(q_out))]
    I1, I2 = stats_at(meta, mean(q_θ), reshape(x, :, 1), [Float64(mean(q_out))], yv, collect(Float64, mean(q_v)), Matrix{Float64}(meta.ref.Uv))
    return I1[1], I2[1]
end
@rule UniSGP(:w, Marginalisation) (q_out::PointMass, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipSGPMeta) =
    (I = point_stats(q_out, q_in, q_v, q_θ, meta); GammaShapeRate(1.5, 0.5 * (I[1] + I[2])))
@rule UniSGP(:w, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipSGPMeta) =
    (I = point_stats(q_out, q_in, q_v, q_θ, meta); GammaShapeRate(1.5, 0.5 * (I[1] + I[2])))

hip_energy(q_out, q_in, q_v, q_w, q_θ, meta) = (I = point_stats(q_out, q_in, q_v, q_θ, meta); w = mean(q_w); 0.5 * (I[1] * w - elog(q_w) + log(2π) + I[2] * w))
@average_energy UniSGP (q_out::PointMass, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::GammaShapeRate, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_out, q_in, q_v, q_w, q_θ, meta)
@average_energy UniSGP (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::GammaShapeRate, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_out, q_in, q_v, q_w, q_θ, meta)
@average_energy UniSGP (q_out::PointMass, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::PointMass, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_out, q_in, q_v, q_w, q_θ, meta)
@average_energy UniSGP (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::PointMass, q_θ::PointMass, meta::HipSGPMeta) = hip_energy(q_out, q_in, q_v, q_w, q_θ, meta)

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

# ---- the cold rules, device-backed (unisgp.py:183-387; VERDICT r3 item 7) -----------------------------------------------
# Stand-alone per-point (I1_n, I2_n) of GPnode/UniSGPnode.jl:196-238 for arbitrary points X (D × n), at kernel(θ) and at the
# q(v) given by (μ_v, Uv = chol(Σ_v + μ μ').U): K_uu chain and K_uf on the device, then the per-point quadratic forms.  On an
# auxiliary handle: the main one holds the last swept batch.
function aux_handle!(meta::HipSGPMeta, n::Int)
    if meta.aux === nothing || meta.aux.n_max < n
        meta.aux = Handle(max(n, 64), inducing_matrix(meta.ref.Xu), 1; device = meta.device)
    end
    return meta.aux
end
function stats_at(meta::HipSGPMeta, θ, X::Matrix{Float64}, y::Vector{Float64}, yv, μ_v::Vector{Float64}, Uv)
    h = aux_handle!(meta, size(X, 2))
    set_data!(h, X, y, yv, nothing, -1.0)
    σ², ℓ = meta.kernel_params(θ)
    set_kernel!(h, σ², collect(Float64, ℓ), meta.jitter)
    sweep_local!(h)
    set_posterior!(h, μ_v, Matrix{Float64}(Uv))
    return w_stats(h, size(X, 2))
end
# chol(Σ_v + μ μ').U of an explicit q_v, factored on the device
uv_of(q_v, meta::HipSGPMeta) = (μ = mean(q_v); Matrix(potrf(Matrix{Float64}(cov(q_v) + μ * μ'); device = meta.device)'))

# (I1, I2) of ONE node whose input is uncertain: Ψ-statistics by the meta's cubature, Ψ2 + jitter_psi2 I, clamped like the
# reference (GPnode/UniSGPnode.jl:186-190)
function node_I(q_out, q_in, μ_v, Uv, θ, meta::HipSGPMeta, jitter_psi2, clamped)
    pts, ω = cubature_1d(meta, q_in)
    μ_y = mean(q_out); v_y = q_out isa PointMass ? 0.0 : var(q_out)
    I1q, I2q = stats_at(meta, θ, reshape(pts, 1, :), fill(Float64(μ_y), length(ω)), nothing, collect(Float64, μ_v), Uv)
    I1 = dot(ω, I1q)
    I2 = dot(ω, I2q) + μ_y^2 * (1.0 - sum(ω)) + v_y
    if jitter_psi2 != 0
        KuuL = Matrix{Float64}(meta.ref.KuuL)
        I1 -= jitter_psi2 * tr(potri(KuuL * KuuL'; device = meta.device))      # tr(Kuu^-1 (jitter I))
        I2 += jitter_psi2 * sum(abs2, Uv)                                      # tr(Uv'Uv (jitter I))
    end
    clamped && ((I1, I2) = (clamp(I1, 1e-12, 1e12), clamp(I2, 1e-12, 1e12)))
    return I1, I2
end

# :out with an uncertain input (GPnode/UniSGPnode.jl:85-93): Ψ1 = Σ_s ω_s K(Xu, x_s), mean = Ψ1 · μ_v
@rule UniSGP(:out, Marginalisation) (q_in::UnivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) = begin
    pts, ω = cubature_1d(meta, q_in)
    f = predict(meta, reshape(pts, 1, :), collect(Float64, mean(q_v)), mean(q_θ))
    return NormalMeanPrecision(dot(ω, f), mean(q_w))
end
# :w with an uncertain input (GPnode/UniSGPnode.jl:177-192): meta.Uv, Ψ2 + 1e-8 I, clamped
@rule UniSGP(:w, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_in::UnivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_θ::PointMass, meta::HipSGPMeta) =
    (I = node_I(q_out, q_in, mean(q_v), Matrix{Float64}(meta.ref.Uv), mean(q_θ), meta, 1e-8, true); GammaShapeRate(1.5, 0.5 * (I[1] + I[2])))
# average energy with an uncertain input: Gamma q_w (GPnode/UniSGPnode.jl:290-313: meta.KuuL / meta.Uv, Ψ2 + 1e-8 I, clamped) and
# PointMass q_w (:390-409: Σ_v + μ μ' from q_v itself; the reference's `.+ 1e-8` on every entry is not reproduced, < 1e-6)
@average_energy UniSGP (q_out::UnivariateNormalDistributionsFamily, q_in::UnivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::GammaShapeRate, q_θ::PointMass, meta::HipSGPMeta) = begin
    I1, I2 = node_I(q_out, q_in, mean(q_v), Matrix{Float64}(meta.ref.Uv), mean(q_θ), meta, 1e-8, true)
    w = mean(q_w)
    return 0.5 * (I1 * w - elog(q_w) + log(2π) + I2 * w)
end
@average_energy UniSGP (q_out::UnivariateNormalDistributionsFamily, q_in::UnivariateGaussianDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::PointMass, q_θ::PointMass, meta::HipSGPMeta) = begin
    I1, I2 = node_I(q_out, q_in, mean(q_v), uv_of(q_v, meta), mean(q_θ), meta, 0.0, true)
    w = mean(q_w)
    return 0.5 * (I1 * w - elog(q_w) + log(2π) + I2 * w)
end
# :in (GPnode/UniSGPnode.jl:107-122): x -> -w/2 A(x) + w μ_y B(x)·μ_v - w/2 |Uv B(x)|² = -w/2 (I1(x) + I2(x) - μ_y²), every
# evaluation one device pass (K_uu chain, K_uf column, the two quadratic forms); meta.Uv as the reference's rule reads it
@rule UniSGP(:in, Marginalisation) (q_out::UnivariateNormalDistributionsFamily, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipSGPMeta) = begin
    w = mean(q_w); μ_y = mean(q_out); μ_v = collect(Float64, mean(q_v)); θ = mean(q_θ); Uv = Matrix{Float64}(meta.ref.Uv)
    log_backwardmess = (x) -> begin
        I1, I2 = stats_at(meta, θ, reshape(collect(Float64, x), :, 1), [Float64(μ_y)], nothing, μ_v, Uv)
        -0.5 * w * (I1[1] + I2[1] - μ_y^2)
    end
    return ContinuousUnivariateLogPdf(log_backwardmess)
end
# :θ (GPnode/UniSGPnode.jl:242-287, three methods): θ -> w μ_y Ψ1(θ)·μ_v - w/2 (Ψ0(θ) + tr(Ψ2(θ) (Rv - Kuu^-1(θ))))
#    = -w/2 (Σ_q ω_q (I1_q + I2_q) - μ_y² Σ_q ω_q),   Rv from q_v itself
function theta_closure(q_out, q_in, q_v, q_w, meta::HipSGPMeta)
    w = mean(q_w); μ_y = Float64(mean(q_out)); μ_v = collect(Float64, mean(q_v)); Uv = uv_of(q_v, meta)
    pts, ω = q_in isa PointMass ? (point(q_in), [1.0]) : cubature_1d(meta, q_in)
    X = q_in isa PointMass ? reshape(pts, :, 1) : reshape(pts, 1, :)
    log_backwardmess = (θ) -> begin
        I1, I2 = stats_at(meta, θ, X, fill(μ_y, length(ω)), nothing, μ_v, Uv)
        -0.5 * w * (dot(ω, I1 .+ I2) - μ_y^2 * sum(ω))
    end
    return ContinuousMultivariateLogPdf(UnspecifiedDomain(), log_backwardmess)
end
@rule UniSGP(:θ, Marginalisation) (q_out::PointMass, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::Any, meta::HipSGPMeta) =
    theta_closure(q_out, q_in, q_v, q_w, meta)
@rule UniSGP(:θ, Marginalisation) (q_out::UnivariateGaussianDistributionsFamily, q_in::PointMass, q_v::MultivariateNormalDistributionsFamily, q_w::Any, meta::HipSGPMeta) =
    theta_closure(q_out, q_in, q_v, q_w, meta)
@rule UniSGP(:θ, Marginalisation) (q_out::UnivariateNormalDistributionsFamily, q_in::UnivariateNormalDistributionsFamily, q_v::MultivariateGaussianDistributionsFamily, q_w::Any, meta::HipSGPMeta) =
    theta_closure(q_out, q_in, q_v, q_w, meta)
# (The Gaussian × log-pdf product of GPnode/UniSGPnode.jl:39-54 dispatches on the message types, not on the meta: the reference's
# own method applies unchanged -- its 21 closure evaluations are 21 device passes.)

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
    aux::Union{Nothing,Handle}        # single-output handle for the :in / :θ closures (it replaces data and posterior)
    device::Int
end

function HipMultiSGPMeta(ref::MultiSGPMeta, d_out::Int; kernel_params, n_steps, jitter = 0.0, device = 0)
    Xu = inducing_matrix(ref.Xu)
    npts = 2 * size(Xu, 1) + 1                                 # srcubature: 2 d_in + 1 points per step
    return HipMultiSGPMeta(ref, d_out, Handle(n_steps * npts, Xu, d_out; device = device), Handle(npts, Xu, 1; device = device),
                           kernel_params, Float64(jitter), nothing, device)
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

# ---- :in and :θ (multisgp.py:169-278) -- both closures are per-point device quantities at a PSEUDO-posterior ------------------
# s = Σ_d μ_v^(d) (μ_y' W)_d  (sum_diagonal_M) and S = Σ_ij W_ij Rv_blk[i][j]  (create_blockmatrix), Rv = Σ_v + μ μ': what the
# closures keep of q(v) (GPnode/MultiSGPnode.jl:176-179).  With mean s, factor chol(S).U and pseudo-observations y = 1,
# sgp_w_stats returns I1(x) = k(x,x) - k' Kuu^-1 k and I2(x) = 1 - 2 s·k + k' S k.
function second_moment_contraction(q_out, q_v, W, M)
    μ_y = collect(Float64, mean(q_out)); D = length(μ_y)
    μ_v, Σ_v = mean_cov(q_v)
    Rv = Σ_v + μ_v * μ_v'
    row = μ_y' * W
    s = sum(view(μ_v, (d-1)*M+1:d*M) .* row[d] for d in 1:D)
    S = sum(view(Rv, (i-1)*M+1:i*M, (j-1)*M+1:j*M) .* W[i, j] for i in 1:D, j in 1:D)
    return collect(Float64, s), Matrix{Float64}(0.5 * (S + S'))
end
function aux_handle!(meta::HipMultiSGPMeta, n::Int)
    if meta.aux === nothing || meta.aux.n_max < n
        meta.aux = Handle(max(n, 64), inducing_matrix(meta.ref.Xu), 1; device = meta.device)
    end
    return meta.aux
end
function pseudo_stats(meta::HipMultiSGPMeta, X::Matrix{Float64}, σ², ℓ, jitter, s, US)
    h = aux_handle!(meta, size(X, 2))
    set_data!(h, X, ones(size(X, 2)), nothing, nothing, -1.0)
    set_kernel!(h, σ², collect(Float64, ℓ), jitter)
    sweep_local!(h)
    set_posterior!(h, s, US)
    return w_stats(h, size(X, 2))
end
# :in (GPnode/MultiSGPnode.jl:162-208): x -> -1/2 tr(W) I1(x) + s·k(x) - 1/2 k(x)' S k(x) = -1/2 tr(W) I1 - 1/2 (I2 - 1)
function multi_in_closure(q_out, q_v, q_w, q_θ, meta::HipMultiSGPMeta)
    W = Matrix{Float64}(mean(q_w)); M = meta.step.m
    s, S = second_moment_contraction(q_out, q_v, W, M)
    US = Matrix(potrf(S; device = meta.device)')                # upper factor: |US k|² = k' S k
    σ², ℓ = meta.kernel_params(mean(q_θ))
    trW = tr(W)
    return (x) -> begin
        I1, I2 = pseudo_stats(meta, reshape(collect(Float64, x), :, 1), σ², ℓ, meta.jitter, s, US)
        -0.5 * trW * I1[1] - 0.5 * (I2[1] - 1.0)
    end
end
@rule MultiSGP(:in, Marginalisation) (q_out::Any, q_v::MultivariateNormalDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) =
    ContinuousMultivariateLogPdf(UnspecifiedDomain(), multi_in_closure(q_out, q_v, q_w, q_θ, meta))
# the Laplace variant (:210-236: q_out::PointMass, Gaussian q_in, point-mass q_w) minimises the same closure; the reference's own
# optimiser code runs unchanged on it -- every evaluation is one device pass
@rule MultiSGP(:in, Marginalisation) (q_out::PointMass, q_in::MultivariateGaussianDistributionsFamily, q_v::MultivariateGaussianDistributionsFamily, q_w::Any, q_θ::PointMass, meta::HipMultiSGPMeta) = begin
    f = multi_in_closure(q_out, q_v, q_w, q_θ, meta)
    neg = (x) -> -f(x)
    D = length(mean(q_in)); E = Matrix{Float64}(I, D, D); h1 = 1e-5
    # (the reference differentiates with ForwardDiff / Zygote, :227-231; a ccall cannot be traced, so gradient and Hessian are
    # central differences of the device-evaluated closure, as in multisgp.py: rule_in_laplace)
    grad! = (G, x) -> (for a in 1:D; G[a] = (neg(x + h1 * E[:, a]) - neg(x - h1 * E[:, a])) / (2 * h1); end; G)
    m_z = Optim.optimize(neg, grad!, collect(Float64, mean(q_in)), Optim.LBFGS(), Optim.Options(iterations = 20); inplace = true).minimizer   # :229
    hh = 1e-4
    W_z = [(neg(m_z + hh * (E[:, a] + E[:, b])) - neg(m_z + hh * (E[:, a] - E[:, b])) - neg(m_z - hh * (E[:, a] - E[:, b])) + neg(m_z - hh * (E[:, a] + E[:, b]))) / (4 * hh^2)
           for a in 1:D, b in 1:D]                               # Hessian at the minimiser by central differences (:231-233 uses ForwardDiff)
    return MvNormalWeightedMeanPrecision(W_z * m_z, W_z)        # :235
end
# :θ (GPnode/MultiSGPnode.jl:447-466): θ -> -1/2 tr(W) (Ψ0 - tr(Kuu^-1 Ψ2')) + Ψ1·s - 1/2 tr(Ψ2' S), Ψ2' = Ψ2 + 1e-7 I (:458),
# Kuu(θ) without jitter (:455): per cubature point the :in closure; the 1e-7 I term adds 1e-7 (tr(W) tr(Kuu^-1) - tr(S)) / 2
@rule MultiSGP(:θ, Marginalisation) (q_out::Any, q_in::MultivariateGaussianDistributionsFamily, q_v::MultivariateGaussianDistributionsFamily, q_w::Any, meta::HipMultiSGPMeta) = begin
    W = Matrix{Float64}(mean(q_w)); M = meta.step.m
    s, S = second_moment_contraction(q_out, q_v, W, M)
    US = Matrix(potrf(S; device = meta.device)')
    trW, trS = tr(W), tr(S)
    X, ω = cubature(meta, q_in)
    Xu = inducing_matrix(meta.ref.Xu)
    log_backwardmess = (θ) -> begin
        σ², ℓ = meta.kernel_params(θ)
        I1, I2 = pseudo_stats(meta, X, σ², ℓ, 0.0, s, US)
        tr_kinv = tr(potri(kernelmatrix_dev(Xu, Xu, σ², collect(Float64, ℓ); device = meta.device); device = meta.device))
        dot(ω, -0.5 * trW .* I1 .- 0.5 .* (I2 .- 1.0)) + 0.5e-7 * (trW * tr_kinv - trS)
    end
    return ContinuousMultivariateLogPdf(UnspecifiedDomain(), log_backwardmess)
end
# the per-step average energy: the reference's own method with the wrapped meta (its Ψ-statistics buffers were filled by the
# device through psi_statistics! in the :v / :w rules of the same step)
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
