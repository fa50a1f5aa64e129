"""Matern covariances on the HIP path -- counterpart of gpmp/kernel/matern.py.

``maternp_covariance(x, y, p, param, pairwise)`` keeps the reference signature and dispatch
(``y is x or y is None`` -> ii/tt path with the 10 * sigma^2 * eps nugget, matern.py:124-141); the
distance, the Matern polynomial x exponential and the diagonal add are one fused kernel
(gpmp_matern_gram) instead of cdist + ~2p+3 temporaries + eye(n).
"""
import ctypes
import math

import torch

from .. import _lib
from .. import num as gnp


def matern32_kernel(h):
    """gpmp/kernel/matern.py:10-29."""
    return maternp_kernel(1, h)


def maternp_kernel(p: int, h):
    """gpmp/kernel/matern.py:32-64 -- half-integer Matern correlation on a device array."""
    lib = _lib.load()
    h = gnp.asarray(h).to(torch.float64).contiguous()
    out = torch.empty_like(h)
    _lib.check(lib.gpmp_maternp_kernel(gnp._ptr(h), h.numel(), int(p), gnp._ptr(out), gnp._stream()), "gpmp_maternp_kernel")
    return out


def _gram(x, y, p, param, noise, diag_add, lower_only=False):
    lib = _lib.load()
    x = gnp._points(x)
    n, d = x.shape
    th = gnp._host_params(param)
    if th.shape[0] != d + (2 if noise else 1):
        raise ValueError(f"covparam has {th.shape[0]} entries, expected {d + (2 if noise else 1)}")
    if y is None:
        m, yp = n, None
    else:
        y = gnp._points(y)
        if y.shape[1] != d:
            raise ValueError("x and y must have the same number of columns")
        m, yp = y.shape[0], y
    K = gnp.alloc_matrix(n, m)
    hv = _lib.host_vec(th)
    _lib.check(
        lib.gpmp_matern_gram(gnp._ptr(x), gnp._ptr(yp), n, m, d, int(p), hv, 1 if noise else 0, float(diag_add),
                             1 if lower_only else 0, gnp._ptr(K), gnp._ld(K), gnp._stream()),
        "gpmp_matern_gram",
    )
    return K


def _pairwise(x, y, p, param, noise):
    lib = _lib.load()
    x = gnp._points(x)
    n, d = x.shape
    th = gnp._host_params(param)
    yp = None if y is None else gnp._points(y)
    out = torch.empty(n, dtype=torch.float64, device=x.device)
    hv = _lib.host_vec(th)
    _lib.check(lib.gpmp_matern_pairwise(gnp._ptr(x), gnp._ptr(yp), n, d, int(p), hv, 1 if noise else 0, gnp._ptr(out), gnp._stream()),
               "gpmp_matern_pairwise")
    return out


def maternp_covariance_ii_or_tt(x, p, param, pairwise=False):
    """gpmp/kernel/matern.py:67-94."""
    th = gnp._host_params(param)
    sigma2 = math.exp(th[0])
    if pairwise:
        return sigma2 * gnp.ones((x.shape[0],))
    nugget = 10.0 * sigma2 * gnp.eps
    return _gram(x, None, p, th, False, nugget)


def maternp_covariance_it(x, y, p, param, pairwise=False):
    """gpmp/kernel/matern.py:97-121."""
    if pairwise:
        return _pairwise(x, y, p, param, False)
    return _gram(x, y, p, param, False, 0.0)


def maternp_covariance(x, y, p, param, pairwise=False):
    """gpmp/kernel/matern.py:124-141 (identity test, not equality)."""
    if y is x or y is None:
        return maternp_covariance_ii_or_tt(x, p, param, pairwise)
    return maternp_covariance_it(x, y, p, param, pairwise)


class MaternCovariance:
    """Covariance callable ``k(x, y, covparam, pairwise=False)`` with a declared structure.

    Equivalent to the closures every reference example builds around ``maternp_covariance``
    (examples/gpmp_example02_1d_interpolation.py:41-43); ``noise=True`` gives the
    ``[log s2, log s2_noise, log 1/rho...]`` kernel of examples/gpmp_example07_nd_regression.py:95-131.
    Declaring (p, noise) is what lets the criteria use the analytic gradient and the lower-triangle
    Gram build; any other callable still works through the generic path (finite-difference gradient).
    """

    def __init__(self, p: int, noise: bool = False):
        self.p, self.noise = int(p), bool(noise)

    def __call__(self, x, y, covparam, pairwise=False):
        if not self.noise:
            return maternp_covariance(x, y, self.p, covparam, pairwise)
        th = gnp._host_params(covparam)
        if y is x or y is None:
            if pairwise:
                return math.exp(th[0]) * gnp.ones((x.shape[0],))
            return _gram(x, None, self.p, th, True, math.exp(th[1]))
        if pairwise:
            return _pairwise(x, y, self.p, th, True)
        return _gram(x, y, self.p, th, True, 0.0)

    def gram_lower(self, x, covparam):
        """K(x, x) with only the tiles on/below the diagonal written (input of the Cholesky)."""
        th = gnp._host_params(covparam)
        diag = math.exp(th[1]) if self.noise else 10.0 * math.exp(th[0]) * gnp.eps
        return _gram(x, None, self.p, th, self.noise, diag, lower_only=True)

    def __repr__(self):
        return f"MaternCovariance(p={self.p}, noise={self.noise})"
