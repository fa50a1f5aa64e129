"""Fisher information of the covariance parameters -- counterpart of gpmp/core/fisher.py.

I_ij = 1/2 tr(K^-1 dK_i K^-1 dK_j)  (and the contrast-space form with Qinv = W (W^T K W)^-1 W^T).
The reference differentiates the covariance by 5-point finite differences (fisher.py:40-50); for a declared
``MaternCovariance`` the derivative matrices are analytic (gpmp_matern_gram_deriv), otherwise the same finite
differences are taken.  K^-1 dK_i costs two triangular solves with n right-hand sides on the MFMA GEMM.
"""
import numpy as np
import torch

from .. import _lib
from .. import num as gnp
from .utils import mean_values as _mean_values
from ..kernel.matern import MaternCovariance


def _derivative_matrices(model, xi, theta, epsilon):
    xi = gnp._points(xi)
    n, d = xi.shape
    p = len(theta)
    cov = model.covariance
    out = []
    if isinstance(cov, MaternCovariance):
        lib = _lib.load()
        hv = _lib.host_vec(theta)
        for j in range(p):
            D = gnp.alloc_matrix(n, n)
            _lib.check(lib.gpmp_matern_gram_deriv(gnp._ptr(xi), n, d, cov.p, hv, 1 if cov.noise else 0, j, gnp._ptr(D), gnp._ld(D),
                                                  gnp._stream()), "gpmp_matern_gram_deriv")
            out.append(D)
        return out
    for j in range(p):  # generic callable: 5-point central differences, as the reference
        def f(v, j=j):
            t = np.array(theta, dtype=np.float64)
            t[j] = v
            return gnp.asarray(cov(xi, xi, t))
        out.append(gnp.as_matrix(gnp.derivative_finite_diff(f, float(theta[j]), epsilon)))
    return out


def _pairwise_half_traces(B):
    """I_ij = 1/2 tr(B_i B_j) = 1/2 sum_ab B_i[a,b] B_j[b,a]."""
    p = len(B)
    out = np.empty((p, p))
    for i in range(p):
        Bit = B[i].T
        for j in range(i, p):
            out[i, j] = out[j, i] = 0.5 * float(torch.sum(Bit * B[j]).item())
    return out


def fisher_information(model, xi, covparam=None, epsilon: float = 1e-3):
    """gpmp/core/fisher.py:18-78."""
    theta = np.asarray(gnp.to_np(model.covparam if covparam is None else covparam), dtype=np.float64).reshape(-1)
    xi = gnp.asarray(xi)
    try:
        F = gnp.cholesky_factor(gnp.asarray(model.covariance(xi, xi, theta)), overwrite=True)
    except Exception as exc:
        raise RuntimeError("Covariance matrix not invertible; adjust hyperparameters or add jitter.") from exc
    B = [F.solve(D) for D in _derivative_matrices(model, xi, theta, epsilon)]   # K^-1 dK_i
    return _pairwise_half_traces(B)


def fisher_information_cpd(model, xi, covparam=None, epsilon: float = 1e-3):
    """gpmp/core/fisher.py:81-147: contrast-space Fisher information for a linear-predictor mean,
    with Qinv dK_i = K^-1 dK_i - U S^-1 (U^T dK_i), U = K^-1 P (no complete QR)."""
    if model.meantype != "linear_predictor":
        return fisher_information(model, xi, covparam=covparam, epsilon=epsilon)
    theta = np.asarray(gnp.to_np(model.covparam if covparam is None else covparam), dtype=np.float64).reshape(-1)
    xi = gnp.asarray(xi)
    F = gnp.cholesky_factor(gnp.asarray(model.covariance(xi, xi, theta)), overwrite=True)
    P = _mean_values(model, xi, model.meanparam)
    U = F.solve(P)                                   # n x q
    S = gnp.matmul(P, U, ta=True)                    # q x q = P^T K^-1 P (library GEMM)
    US = gnp.matmul(U, gnp.small_spd_inverse(S, "P^T K^-1 P"))   # n x q
    B = []
    for D in _derivative_matrices(model, xi, theta, epsilon):
        KD = F.solve(D)
        B.append(KD - gnp.matmul(US, gnp.matmul(U, D, ta=True)))
    return _pairwise_half_traces(B)


def fisher_information_torch(model, xi, covparam):
    """gpmp/core/fisher.py:158-191: 0.5 * Hessian of log|K(theta)| through ``gnp.SecondOrderDifferentiableFunction``
    (second-order autograd in the reference's torch backend; central finite differences of the HIP log-det here)."""
    xi = gnp.asarray(xi)

    def log_det_cov(params):
        return gnp.cholesky_factor(gnp.asarray(model.covariance(xi, xi, params)), overwrite=True).logdet()

    sodf = gnp.SecondOrderDifferentiableFunction(log_det_cov)
    sodf.evaluate(np.asarray(gnp.to_np(covparam), dtype=np.float64))
    sodf.gradient()
    return 0.5 * sodf.hessian()
