"""Leave-one-out by virtual cross-validation -- counterpart of gpmp/core/loo.py."""
import torch

from .. import num as gnp
from .utils import mean_values as _mean_values
from .linalg import covariance_factor


def loo(model, xi, zi):
    """gpmp/core/loo.py:21-59 -> (zloo, sigma2loo, eloo)."""
    if model.meantype == "zero":
        return _loo_with_zero_mean(model, model.covparam, xi, zi)
    if model.meantype == "parameterized":
        return _loo_with_parameterized_mean(model, model.meanparam, model.covparam, xi, zi)
    if model.meantype == "linear_predictor":
        return _loo_with_linear_predictor_mean_cpd(model, model.meanparam, model.covparam, xi, zi)
    raise ValueError(f"Unknown mean type: {model.meantype}")


def _kinv_diag_and_solve(model, covparam, xi, Y):
    """diag(K^-1) (trtri + column sums of squares, linalg.py:17-46) and K^-1 Y."""
    F = covariance_factor(model, xi, covparam)
    T = F.inverse_factor()
    d = gnp.coldots(T, None)[0]
    return d, F.solve(Y)


def _loo_with_zero_mean(model, covparam, xi, zi):
    """gpmp/core/loo.py:65-83."""
    d, Kinv_zi = _kinv_diag_and_solve(model, covparam, xi, zi.reshape(-1))
    eloo = Kinv_zi.reshape(-1) / d
    sigma2loo = 1.0 / d
    return zi - eloo, sigma2loo, eloo


def _loo_with_parameterized_mean(model, meanparam, covparam, xi, zi):
    """gpmp/core/loo.py:89-97."""
    zi_prior_mean = _mean_values(model, xi, meanparam).reshape(-1)
    zloo_c, sigma2loo, eloo_c = _loo_with_zero_mean(model, covparam, xi, zi - zi_prior_mean)
    return zloo_c + zi_prior_mean, sigma2loo, eloo_c


def _loo_with_linear_predictor_mean_cpd(model, meanparam, covparam, xi, zi):
    """gpmp/core/loo.py:103-130 with Qinv = K^-1 - U S^-1 U^T, U = K^-1 P (no QR, no n^3 GEMMs)."""
    P = _mean_values(model, xi, meanparam)
    Y = gnp.hstack((zi.reshape(-1, 1), P))
    d, X = _kinv_diag_and_solve(model, covparam, xi, Y)
    Kinv_z, U = X[:, 0], X[:, 1:]
    G = gnp.coldots(U, Y)[:-1]                   # (1 + q) x q: rows z^T U, P^T U  (one pass over U)
    S = G[1:]                                    # q x q = P^T K^-1 P
    S = 0.5 * (S + S.T)
    US = gnp.matmul(U, gnp.small_spd_inverse(S, "P^T K^-1 P"))      # n x q on the library GEMM
    Qinv_z = Kinv_z - gnp.matmul(US, G[0])
    Qinv_diag = d - torch.sum(US * U, dim=1)
    eloo = Qinv_z / Qinv_diag
    return zi - eloo, 1.0 / Qinv_diag, eloo
