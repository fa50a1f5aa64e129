"""Analytic gradients of the ML / REML criteria with respect to the covariance parameters.

The reference differentiates these criteria by torch autograd through cdist / exp / cholesky /
solve_triangular (gpmp/num/torch_backend.py:574-604) or, on the NumPy backend, by SciPy finite
differences (numpy_backend.py:333).  Here, for a declared Matern covariance:

    d NLL / d theta_j  = 1/2 tr( (K^-1 - a a^T) dK/dtheta_j ),           a = K^-1 z
    d REML / d theta_j = 1/2 tr( (Qinv - b b^T) dK/dtheta_j ),           b = Qinv z,
                         Qinv = K^-1 - U S^-1 U^T,  U = K^-1 P,  S = P^T U

K^-1 comes from potrf -> trtri -> T^T T on the MFMA GEMM; the trace against dK/dtheta_j is one fused
pass (gpmp_matern_grad_trace) that recomputes the scaled differences on the fly and subtracts the
low-rank part  sum_a F[i,a] G[k,a]  in registers -- K and dK are never stored.
"""
import math

import numpy
import torch

from .. import _lib
from .. import num as gnp
from .utils import mean_values as _mean_values
from ..kernel.matern import MaternCovariance
from .linalg import MeanSpace, covariance_factor


def _grad_trace(cov: MaternCovariance, Kinv, xi, covparam, F, G):
    lib = _lib.load()
    xi = gnp._points(xi)
    n, d = xi.shape
    th = gnp._host_params(covparam)
    r = F.shape[1]
    Fm, Gm = gnp.as_matrix(F), gnp.as_matrix(G)
    if gnp._ld(Fm) != gnp._ld(Gm):
        Gm = gnp.as_matrix(Gm, copy=True)
        Fm = gnp.as_matrix(Fm, copy=True)
    g = torch.zeros(len(th), dtype=torch.float64, device=xi.device)
    ws = torch.empty(int(lib.gpmp_grad_ws_elems(n, d)), dtype=torch.float64, device=xi.device)
    hv = _lib.host_vec(th)
    _lib.check(
        lib.gpmp_matern_grad_trace(gnp._ptr(Kinv), gnp._ld(Kinv), gnp._ptr(xi), n, d, cov.p, hv, 1 if cov.noise else 0,
                                   gnp._ptr(Fm), gnp._ptr(Gm), r, gnp._ld(Fm), gnp._ptr(g), gnp._ptr(ws), gnp._stream()),
        "gpmp_matern_grad_trace",
    )
    return 0.5 * gnp.to_np(g)


class MLZeroMeanAnalytic:
    """negative_log_likelihood_zero_mean (likelihood.py:18-52) value + gradient."""

    def __init__(self, model, mean_offset=None):
        self.model = model
        self.mean_offset = mean_offset  # callable xi -> prior mean vector (parameterized mean), or None

    def value_and_state(self, covparam, xi, zi):
        xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
        if self.mean_offset is not None:
            zi = zi - self.mean_offset(xi)
        n = xi.shape[0]
        F = covariance_factor(self.model, xi, covparam)
        w = F.solve_lower(zi)
        norm2 = float(gnp.sum(w * w).item())
        value = 0.5 * (n * math.log(2.0 * math.pi) + F.logdet() + norm2)
        return value, (F, w, xi, numpy.array(covparam, dtype=numpy.float64))

    def gradient_from_state(self, state):
        F, w, xi, covparam = state
        alpha = F.solve_lower(w, trans=True).reshape(-1, 1)
        Kinv = F.inverse_lower()
        return _grad_trace(self.model.covariance, Kinv, xi, covparam, alpha, alpha)


class REMLAnalytic:
    """negative_log_restricted_likelihood (likelihood.py:92-129) value + gradient."""

    def __init__(self, model):
        self.model = model

    def value_and_state(self, covparam, xi, zi):
        xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
        F = covariance_factor(self.model, xi, covparam)
        P = _mean_values(self.model, xi, self.model.meanparam)
        n, q = P.shape
        ms = MeanSpace(F, zi, P)
        value = 0.5 * ((n - q) * math.log(2.0 * math.pi) + ms.logdet_contrast() + ms.quad())
        return value, (F, ms, xi, numpy.array(covparam, dtype=numpy.float64))

    def gradient_from_state(self, state):
        F, ms, xi, covparam = state
        X = F.solve_lower(ms.W, trans=True)           # K^-1 [z, P]
        alpha, U = X[:, 0], X[:, 1:]
        Sinv = numpy.linalg.inv(ms.S)
        US = gnp.matmul(U, gnp.asarray(Sinv))         # n x q on the library GEMM
        beta = alpha - gnp.matmul(US, gnp.asarray(ms.b))   # Qinv z
        Fm = gnp.hstack((US, beta.reshape(-1, 1)))
        Gm = gnp.hstack((U, beta.reshape(-1, 1)))
        Kinv = F.inverse_lower()
        return _grad_trace(self.model.covariance, Kinv, xi, covparam, Fm, Gm)
