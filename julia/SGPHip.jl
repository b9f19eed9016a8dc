# SGPHip.jl -- the binding a GaussianProcessNode maintainer would add to route the UniSGP node's hot path
# through libsgp_hip.so (include/sgp_hip.h).  WRITTEN BLIND: Julia is not available in the build pipeline, so this
# file has never been executed.  It mirrors, call for call, the Python host mirror
# (gaussianprocessnode_amd/unisgp.py), which IS tested against the reference's rule tests on the GPU.
#
# Usage (after `include("GPnode/UniSGPnode.jl")`):
#     include("SGPHip.jl"); using .SGPHip
#     meta = UniSGPMeta(nothing, Xu, Ψ0, Ψ1_trans, Ψ2, KuuL, kernel_gp, Lu, 0, batch_size)   # unchanged
#     SGPHip.attach!(meta; jitter = 0.0, kernel_params = θ -> (softplus(θ[1]), softplus.(θ[2:end])))
# The @rule / prod methods below are MORE SPECIFIC than the reference's (they dispatch on HipMeta), so the model
# code `y[i] ~ UniSGP(x[i], v, w, θ)` and `@meta UniSGP() -> ...` stay as they are.
module SGPHip

using ReactiveMP, LinearAlgebra
import ReactiveMP: @rule, @average_energy, GenericProd, PointMass, MvNormalMeanCovariance, GammaShapeRate, mean, mean_cov
import ..UniSGP, ..UniSGPMeta, ..BufferUniSGP

const LIB = get(ENV, "SGP_HIP_LIB", "libsgp_hip.so")

struct SGPConfig
    n_max::Int64; m::Int32; d::Int32; d_out::Int32; device::Int32; flags::Int32; reserved::Int32
end

mutable struct HipState
    handle::Ptr{Cvoid}
    kernel_params::Function           # θ -> (σ², ℓ::Vector)
    jitter::Float64
    xs::Vector{Vector{Float64}}       # pending points of the current batch
    ys::Vector{Float64}
    vs::Vector{Float64}
    prior::Any
    w::Float64
    Elogw::Float64
    θ::Vector{Float64}
    I1::Vector{Float64}
    I2::Vector{Float64}
    index::Dict{Vector{Float64},Int}
end

const STATE = IdDict{Any,HipState}()   # meta => device state (the reference's meta struct has no spare field)

function check(rc::Cint, h)
    rc == 0 && return
    msg = unsafe_string(ccall((:sgp_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    rc > 0 ? throw(PosDefException(rc)) : error("libsgp_hip: status $rc: $msg")
end

function attach!(meta::UniSGPMeta; kernel_params, jitter = 0.0, device = 0)
    M, D = length(meta.Xu), length(meta.Xu[1])
    cfg = Ref(SGPConfig(meta.N, M, D, 1, device, 2, 0))        # SGP_FLAG_KEEP_KUF
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:sgp_create, LIB), Cint, (Ref{SGPConfig}, Ref{Ptr{Cvoid}}), cfg, h), C_NULL)
    Xu = reduce(hcat, meta.Xu)                                  # D × M, column-major: exactly the ABI layout
    check(ccall((:sgp_set_inducing, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], Xu), h[])
    st = HipState(h[], kernel_params, jitter, [], [], [], nothing, 1.0, 0.0, Float64[], Float64[], Float64[], Dict())
    finalizer(s -> ccall((:sgp_destroy, LIB), Cint, (Ptr{Cvoid},), s.handle), st)
    STATE[meta] = st
    return meta
end

iship(meta) = haskey(STATE, meta)

# ---- :v  (replaces GPnode/UniSGPnode.jl:144-158 and :161-173): O(1) token, no M×M message -------------------
function hip_rule_v(q_out, q_in::PointMass, q_w, q_θ::PointMass, meta::UniSGPMeta)
    st = STATE[meta]
    push!(st.xs, collect(Float64, mean(q_in))); push!(st.ys, mean(q_out))
    push!(st.vs, q_out isa PointMass ? 0.0 : var(q_out))
    st.w = mean(q_w); st.θ = collect(Float64, mean(q_θ))
    st.Elogw = q_w isa GammaShapeRate ? mean(log, q_w) : log(mean(q_w))
    return BufferUniSGP(length(st.xs), meta)
end

# ---- prod (replaces GPnode/UniSGPnode.jl:62-73): one device sweep when counter == N -------------------------
function hip_prod(left, right::BufferUniSGP)
    meta = right.meta; st = STATE[meta]
    meta.counter += 1
    meta.counter == 1 && (st.prior = left)
    meta.counter < meta.N && return st.prior                    # nothing consumes the partial product
    h = st.handle
    X = reduce(hcat, st.xs); n = length(st.ys)
    yv = any(!iszero, st.vs) ? st.vs : C_NULL
    check(ccall((:sgp_set_data, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64),
                h, X, st.ys, yv, C_NULL, n, -1.0), h)
    σ², ℓ = st.kernel_params(st.θ)
    check(ccall((:sgp_set_kernel, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Int32, Float64), h, σ², ℓ, length(ℓ), st.jitter), h)
    check(ccall((:sgp_set_noise, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64), h, [st.w], st.Elogw), h)
    μ0, Σ0 = mean_cov(st.prior)
    check(ccall((:sgp_set_prior, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32), h, μ0, Matrix(Σ0), 0), h)
    check(ccall((:sgp_sweep, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, C_NULL), h)
    M = length(μ0); μ = zeros(M); Σ = zeros(M, M); Uv = zeros(M, M)
    check(ccall((:sgp_get_posterior, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, μ, Σ, Uv), h)
    meta.Uv = UpperTriangular(Uv)                               # :69
    meta.counter = 0                                            # :70
    st.index = Dict(x => i for (i, x) in enumerate(st.xs)); st.I1 = Float64[]; st.I2 = Float64[]
    empty!(st.xs); empty!(st.ys); empty!(st.vs)
    return MvNormalMeanCovariance(μ, Σ)
end

# ---- :w and average energy (replace GPnode/UniSGPnode.jl:196-238, 337-387, 411-436) -------------------------
function point_stats(q_in::PointMass, meta)
    st = STATE[meta]
    if isempty(st.I1)
        n = length(st.index); st.I1 = zeros(n); st.I2 = zeros(n)
        check(ccall((:sgp_w_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), st.handle, st.I1, st.I2, C_NULL), st.handle)
    end
    i = st.index[collect(Float64, mean(q_in))]
    return st.I1[i], st.I2[i]
end
hip_rule_w(q_in, meta) = (I = point_stats(q_in, meta); GammaShapeRate(1.5, 0.5 * (I[1] + I[2])))
function hip_energy(q_in, q_w, meta)
    I1, I2 = point_stats(q_in, meta); w = mean(q_w)
    Elogw = q_w isa GammaShapeRate ? mean(log, q_w) : log(w)
    return 0.5 * (I1 * w - Elogw + log(2π) + I2 * w)
end

# ---- :out (replaces GPnode/UniSGPnode.jl:96-104); use `predict` for whole test sets -------------------------
function predict(meta, Xstar::Matrix{Float64}, μ_v::Vector{Float64}, θ)
    st = STATE[meta]; σ², ℓ = st.kernel_params(θ)
    check(ccall((:sgp_set_kernel, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Int32, Float64), st.handle, σ², ℓ, length(ℓ), st.jitter), st.handle)
    out = zeros(size(Xstar, 2))
    check(ccall((:sgp_predict, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}), st.handle, Xstar, size(Xstar, 2), μ_v, out), st.handle)
    return out
end

# ---- hyper-parameter objective and gradient (replace neg_log_backwardmess_fast / grad_llh_new!,
# helper_functions/derivative_helper.jl:23-39,55-63) at the theta of the last sweep, q(v) fixed --------------------
# Returns (F, dF/d(sigma2, ell...)); the caller applies the chain rule of its kernel_gp(theta) (softplus in the notebooks).
function theta_objective(meta, nparams::Int)
    st = STATE[meta]; f = Ref{Float64}(0.0); g = zeros(nparams)
    check(ccall((:sgp_theta_objective, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}), st.handle, f, g), st.handle)
    return f[], g
end

# ---- minibatch loops (experiments/regression_kin40k.ipynb:205-212): prior <- posterior without leaving the device
carry_posterior!(meta) = (st = STATE[meta]; check(ccall((:sgp_carry_posterior, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), st.handle, C_NULL), st.handle))

end # module

# ---- dispatch glue: the reference's rules stay for metas that are not attached ------------------------------
# (method bodies of GPnode/UniSGPnode.jl gain one line each, e.g.)
#   @rule UniSGP(:v, Marginalisation) (q_out::PointMass, q_in::PointMass, q_w::Any, q_θ::PointMass, meta::UniSGPMeta) = begin
#       SGPHip.iship(meta) && return SGPHip.hip_rule_v(q_out, q_in, q_w, q_θ, meta)
#       ... reference body ...
#   end
#   ReactiveMP.prod(::GenericProd, left::NormalDistributionsFamily, right::BufferUniSGP) =
#       SGPHip.iship(right.meta) ? SGPHip.hip_prod(left, right) : <reference body>
