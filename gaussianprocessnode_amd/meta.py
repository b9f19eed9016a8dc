"""Host-side mirrors of the reference's meta structs and helpers
(helper_functions/gp_helperfunction.jl): `UniSGPMeta`, `MultiSGPMeta`, `GPCache`, accessor functions,
`split2batch`, `SMSE`, `num_error`, `error_rate`, `jdotavx`, `create_blockmatrix`.

The scratch matrices of the reference's metas (Psi0 / Psi1_trans / Psi2) are kept as fields for interface
compatibility but are not used: the batched device path never materialises per-point M x M messages.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np


# ---- kernel parameterisation ------------------------------------------------------------------
def softplus(x):
    return np.logaddexp(0.0, np.asarray(x, dtype=np.float64))


class SEARDKernel:
    """theta -> (sigma2, lengthscales) for `theta[1] * with_lengthscale(SEKernel(), theta[2:end])`.

    softplus_params=True reproduces the notebooks' `kernel_gp` (experiments/regression_kin40k.ipynb:108),
    False the tests' `kernel` (GPtest.jl:21)."""

    def __init__(self, softplus_params: bool = False):
        self.softplus_params = softplus_params

    def __call__(self, theta):
        theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
        if self.softplus_params:
            theta = softplus(theta)
        return float(theta[0]), theta[1:].copy()


# ---- GPCache (helper_functions/gp_helperfunction.jl:16-20,78-123) ------------------------------
class GPCache:
    """Scratch-buffer dictionary of the reference.  The device path owns its scratch in HBM; this mirror only keeps
    the interface (`getcache`, `mul_A_B!` ...) for callers that use it directly."""

    def __init__(self):
        self.cache_matrices = {}
        self.cache_vectors = {}
        self.cache_LowerTriangular = {}


def getcache(cache: GPCache, label):
    sym, size = label
    if isinstance(size, tuple):
        return cache.cache_matrices.setdefault(label, np.empty(size))
    return cache.cache_vectors.setdefault(label, np.empty(size))


def mul_A_B(cache: GPCache, A, B, *sizes):
    """mul_A_B! (:92-101)"""
    out = getcache(cache, (":AB" if len(sizes) == 1 else ":ABdiff", (A.shape[0], B.shape[1])))
    np.matmul(A, B, out=out)
    return out


def mul_A_B_A(cache: GPCache, A, B, size1=None):
    """mul_A_B_A! (:103-110)"""
    return A @ B @ A


def mul_A_B_At(cache: GPCache, A, B, *sizes):
    """mul_A_B_At! (:112-119)"""
    return A @ B @ A.T


def mul_A_v(cache: GPCache, A, v, size=None):
    """mul_A_v! (:120-123)"""
    return A @ v


def jdotavx(a, b):
    """:125-131"""
    return float(np.dot(np.ravel(a), np.ravel(b)))


def create_blockmatrix(A, d, M):
    """:133-135 -- d x d array of M x M views, [i][j] = A[iM:(i+1)M, jM:(j+1)M]"""
    return [[A[i * M:(i + 1) * M, j * M:(j + 1) * M] for j in range(d)] for i in range(d)]


# ---- metas -------------------------------------------------------------------------------------
@dataclass
class UniSGPMeta:
    """helper_functions/gp_helperfunction.jl:33-44 -- same ten positional fields, same order.

    Extra keyword fields configure the device path: `jitter` (added to diag(K_uu); the reference adds it where it
    builds Kuu, experiments/classification_banana.ipynb:163), `device`, and `engine` (an object with the
    `SGPDevice` interface; created lazily on the first sweep -- raises without a gfx950 GPU)."""
    method: object
    Xu: np.ndarray
    Psi0: Optional[np.ndarray]
    Psi1_trans: Optional[np.ndarray]
    Psi2: Optional[np.ndarray]
    KuuL: Optional[np.ndarray]
    kernel: Callable
    Uv: Optional[np.ndarray]
    counter: int = 0
    N: int = 0
    jitter: float = 0.0
    device: int = 0
    engine: object = None
    # per-`infer` state (the reference keeps the analogous state inside ReactiveMP's graph)
    _pending: list = field(default_factory=list, repr=False)
    _prior: object = field(default=None, repr=False)
    _batch: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        Xu = np.asarray(self.Xu, dtype=np.float64)
        self.Xu = Xu[:, None] if Xu.ndim == 1 else Xu          # 1-D inducing inputs as a vector (GPtest.jl:19)


def make_uni_meta(method, Xu, kernel, N, Psi0=None, Psi1_trans=None, Psi2=None, KuuL=None, Uv=None, **kw) -> UniSGPMeta:
    """Convenience constructor: 1-D inducing inputs may be given as a vector (GPtest.jl:19 `Xu = collect(1:Nu)`)."""
    return UniSGPMeta(method, Xu, Psi0, Psi1_trans, Psi2, KuuL, kernel, Uv, 0, N, **kw)


@dataclass
class MultiSGPMeta:
    """helper_functions/gp_helperfunction.jl:55-64"""
    method: object
    Xu: np.ndarray
    Psi0: Optional[np.ndarray]
    Psi1_trans: Optional[np.ndarray]
    Psi2: Optional[np.ndarray]
    Kuu_inverse: Optional[np.ndarray]
    kernel: Callable
    GPCache: Optional[GPCache] = None
    jitter: float = 0.0
    device: int = 0
    engine: object = None


def getmethod(meta):
    return meta.method


def getInducingInput(meta):
    return meta.Xu


def getKernel(meta):
    return meta.kernel


def getPsi0(meta):
    return meta.Psi0


def getPsi1_trans(meta):
    return meta.Psi1_trans


def getPsi2(meta):
    return meta.Psi2


def getUv(meta):
    return meta.Uv


def getKuuInverse(meta):
    return meta.Kuu_inverse


def getGPCache(meta):
    return meta.GPCache


# ---- batching and metrics (helper_functions/gp_helperfunction.jl:137-158) ----------------------
def split2batch(data, batch_size):
    """:137-142"""
    x, y = data
    xb = [x[i:i + batch_size] for i in range(0, len(x), batch_size)]
    yb = [y[i:i + batch_size] for i in range(0, len(y), batch_size)]
    return xb, yb


def SMSE(y_true, y_approx):
    """:145-149 (Julia `var` = unbiased sample variance)"""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_approx = np.asarray(y_approx, dtype=np.float64)
    N = y_true.size
    mse = float(np.linalg.norm(y_true - y_approx) ** 2) / N
    return mse / float(np.var(y_true, ddof=1))


def num_error(ytrue, y):
    """:152-154"""
    return float(np.sum(np.abs(np.asarray(y, dtype=np.float64) - np.asarray(ytrue, dtype=np.float64))))


def error_rate(ytrue, y):
    """:156-158"""
    return num_error(ytrue, y) / len(ytrue)
