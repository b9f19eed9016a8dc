"""The handful of ReactiveMP / ExponentialFamily distribution types that cross the node's message
interface (GPnode/UniSGPnode.jl, GPnode/MultiSGPnode.jl).  Containers with the accessors the reference's
rules call (`mean`, `var`, `cov`, `mean_cov`, `mean_var`, `shape`, `rate`) -- no inference engine here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
from scipy.special import digamma


@dataclass
class PointMass:
    value: object

    def mean(self):
        return self.value

    def var(self):
        return 0.0 if np.ndim(self.value) == 0 else np.zeros_like(np.asarray(self.value, dtype=np.float64))

    def mean_var(self):
        return self.mean(), self.var()


@dataclass
class NormalMeanVariance:
    m: float
    v: float

    def mean(self):
        return self.m

    def var(self):
        return self.v

    def mean_var(self):
        return self.m, self.v


@dataclass
class NormalMeanPrecision:
    m: float
    w: float

    def mean(self):
        return self.m

    def var(self):
        return 1.0 / self.w

    def precision(self):
        return self.w

    def mean_var(self):
        return self.m, 1.0 / self.w


@dataclass
class MvNormalMeanCovariance:
    m: np.ndarray
    S: np.ndarray

    def mean(self):
        return self.m

    def cov(self):
        return self.S

    def mean_cov(self):
        return self.m, self.S


@dataclass
class MvNormalMeanPrecision:
    m: np.ndarray
    W: np.ndarray

    def mean(self):
        return self.m

    def precision(self):
        return self.W

    def cov(self):
        return np.linalg.inv(self.W)

    def mean_cov(self):
        return self.m, self.cov()


@dataclass
class MvNormalWeightedMeanPrecision:
    xi: np.ndarray
    W: np.ndarray

    def weightedmean(self):
        return self.xi

    def precision(self):
        return self.W

    def cov(self):
        return np.linalg.inv(self.W)

    def mean(self):
        return np.linalg.solve(self.W, self.xi)

    def mean_cov(self):
        S = self.cov()
        return S @ self.xi, S


@dataclass
class GammaShapeRate:
    a: float
    b: float

    def shape(self):
        return self.a

    def rate(self):
        return self.b

    def mean(self):
        return self.a / self.b

    def mean_log(self):
        """mean(log, q) of the reference (GPnode/UniSGPnode.jl:340)."""
        return float(digamma(self.a) - math.log(self.b))

    def prod(self, other: "GammaShapeRate") -> "GammaShapeRate":
        """Product of two Gamma densities (ExponentialFamily): shapes add minus one, rates add."""
        return GammaShapeRate(self.a + other.a - 1.0, self.b + other.b)


@dataclass
class WishartFast:
    """Wishart parameterised by its inverse scale (ReactiveMP.WishartFast; GPnode/MultiSGPnode.jl:404)."""
    nu: float
    invS: np.ndarray

    def params(self):
        return self.nu, np.linalg.inv(self.invS)

    def mean(self):
        return self.nu * np.linalg.inv(self.invS)
