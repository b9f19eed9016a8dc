"""Cubature rules the reference takes from ReactiveMP.jl (`ghcubature(p)`, `srcubature()`); call sites
GPnode/UniSGPnode.jl:11-33 and GPnode/MultiSGPnode.jl:11-35 (`getweights` / `getpoints`).  ReactiveMP is not vendored
by the reference, so these are restated from the published definitions (parity unpinned).  Host code:
they only generate the points and weights that go to the device as weighted data."""
from __future__ import annotations

import math

import numpy as np


class GaussHermiteCubature:
    """ghcubature(p): for N(m, P) (univariate) points m + sqrt(2 P) x_i, weights w_i / sqrt(pi)."""

    def __init__(self, p: int):
        self.p = p
        self._x, self._w = np.polynomial.hermite.hermgauss(p)

    def points_weights(self, m, P):
        m, P = float(np.ravel(m)[0]), float(np.ravel(P)[0])
        return (m + math.sqrt(2.0 * P) * self._x)[:, None], self._w / math.sqrt(math.pi)


class SphericalRadialCubature:
    """srcubature(): 2d + 1 points m +/- sqrt(d + 1) L e_i (weight 1 / (2 (d + 1))) and m (weight 1 / (d + 1)), L = chol(P).L."""

    def points_weights(self, m, P):
        m = np.asarray(m, dtype=np.float64).ravel()
        d = m.size
        L = np.linalg.cholesky(np.asarray(P, dtype=np.float64).reshape(d, d))
        r = math.sqrt(d + 1.0)
        pts = [m + r * L[:, i] for i in range(d)] + [m - r * L[:, i] for i in range(d)] + [m.copy()]
        w = np.full(2 * d + 1, 1.0 / (2.0 * (d + 1.0)))
        w[-1] = 1.0 / (d + 1.0)
        return np.stack(pts), w


def ghcubature(p: int) -> GaussHermiteCubature:
    return GaussHermiteCubature(p)


def srcubature() -> SphericalRadialCubature:
    return SphericalRadialCubature()
