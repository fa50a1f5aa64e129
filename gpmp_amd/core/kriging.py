"""Kriging predictors -- counterpart of gpmp/core/kriging.py.

Algorithmic restatement (exact algebra): with K = L L^T,  V = L^-1 K(xi, xt) is the only O(n^2 m)
solve that the posterior mean and variance need,

    zero mean:         mean = V^T (L^-1 z),   var = k(xt,xt) - colsumsq(V)
    linear predictor:  mu = S^-1 (Wp^T V - Pt^T),  S = Wp^T Wp,  Wp = L^-1 P
                       mean = V^T w - mu^T (Wp^T w),  var = k_tt - colsumsq(V) + sum_a mu_a (S mu)_a

instead of the reference's two trsm (kriging.py:62) / the dense (n+q) sysv solve (kriging.py:98-109).
The kriging weights lambda_t = L^-T (V - Wp mu) cost a second solve and are only formed on request.
xt is processed in column chunks so that the n x m_chunk cross-covariance fits the configured budget.
"""
import numpy
import torch

from .. import num as gnp
from .utils import mean_values as _mean_values
from ..config import get_config
from .linalg import covariance_factor


def _chunk_cols(n, m):
    budget = get_config().predict_chunk_bytes
    per_col = 8 * max(n, 1)
    mc = max(256, budget // per_col)
    return int(min(m, mc)) if m > 0 else 0


class _Predictor:
    """State shared by every chunk of one prediction: factor of K, W = L^-1 [z, P], S^-1."""

    def __init__(self, model, xi, zi_centered, use_mean, xt_first=None):
        self.model, self.xi = model, xi
        self.V_first = None
        if xt_first is not None and xt_first.shape[0] > 4:
            # K(xi, xt) of the first chunk is built before the factorisation so that its solve can overlap the factorisation
            Kit = gnp.as_matrix(gnp.asarray(model.covariance(xi, xt_first, model.covparam)))
            self.F, self.V_first = covariance_factor(model, xi, model.covparam, solve_along=Kit)
        else:
            self.F = covariance_factor(model, xi, model.covparam)
        cols = [zi_centered.reshape(-1, 1)]
        self.q = 0
        if use_mean:
            P = _mean_values(model, xi, model.meanparam)
            self.q = P.shape[1]
            cols.append(P)
        self.W = self.F.solve_lower(gnp.hstack(cols) if len(cols) > 1 else cols[0])
        if self.q:
            g = gnp.coldots(self.W, self.W)[:-1]            # (1+q) x (1+q) Gram of W
            S = g[1:, 1:]
            self.S = 0.5 * (S + S.T)
            self.b = g[1:, 0]                               # Wp^T w
            self.Sinv = torch.linalg.inv(self.S)            # q x q, plumbing-sized

    def chunk(self, xt, want_lambda, want_var=True):
        model = self.model
        if self.V_first is not None:                        # first chunk: solved together with the factorisation
            V, self.V_first = self.V_first, None
        else:
            Kit = gnp.as_matrix(gnp.asarray(model.covariance(self.xi, xt, model.covparam)))
            V = self.F.solve_lower(Kit, overwrite=True)     # V = L^-1 Kit, in place
        D = gnp.coldots(V, self.W)                          # rows: V^T w, V^T Wp (q rows), colsumsq(V)
        mean = D[0].clone()
        reduction = D[-1].clone()                           # lambda^T Kit (+ mu^T Pt^T)
        mu = None
        if self.q:
            Pt = _mean_values(model, xt, model.meanparam)   # m x q
            R = D[1:-1] - Pt.T                                  # S mu
            mu = self.Sinv @ R                                  # q x m
            mean = mean - self.b @ mu
            reduction = reduction - torch.sum(mu * R, dim=0)
        lam = None
        if want_lambda:
            if self.q:
                # V - Wp mu on the MFMA GEMM, then lambda = L^-T (.)
                lib = gnp._lib.load()
                Wp = self.W[:, 1:]
                mum = gnp.as_matrix(mu)
                gnp._lib.check(
                    lib.gpmp_dgemm(0, 0, V.shape[0], V.shape[1], self.q, -1.0, gnp._ptr(Wp), gnp._ld(self.W), gnp._ptr(mum),
                                   gnp._ld(mum), 1.0, gnp._ptr(V), gnp._ld(V), 0, gnp._stream()),
                    "gpmp_dgemm",
                )
            lam = self.F.solve_lower(V, trans=True, overwrite=True)
        return mean, reduction, lam, mu


def _run(model, xi, zi_centered, xt, use_mean, want_lambda):
    n, m = xi.shape[0], xt.shape[0]
    mc = _chunk_cols(n, m)
    pred = _Predictor(model, xi, zi_centered, use_mean, xt_first=xt[: max(mc, 1)] if m > 0 else None)
    means, reds, lams, mus = [], [], [], []
    for j0 in range(0, m, max(mc, 1)):
        xtc = xt[j0 : j0 + mc]
        mean, red, lam, mu = pred.chunk(xtc, want_lambda)
        means.append(mean)
        reds.append(red)
        if want_lambda:
            lams.append(lam)
            if mu is not None:
                mus.append(mu)
    cat = lambda parts, dim=0: parts[0] if len(parts) == 1 else torch.cat(parts, dim=dim)  # noqa: E731
    mean = cat(means) if means else gnp.zeros((0,))
    red = cat(reds) if reds else gnp.zeros((0,))
    lam = (cat(lams, 1) if lams else gnp.zeros((n, 0))) if want_lambda else None
    mu = cat(mus, 1) if mus else None
    return mean, red, lam, mu


def _prior_variance(model, xt):
    return gnp.asarray(model.covariance(xt, None, model.covparam, pairwise=True)).reshape(-1)


def kriging_predictor_with_zero_mean(model, xi, xt, return_type=0):
    """gpmp/core/kriging.py:35-67 -> (lambda_t, posterior variance | covariance | None)."""
    xi, xt = gnp.asarray(xi), gnp.asarray(xt)
    zero = gnp.zeros((xi.shape[0],))
    _, red, lam, _ = _run(model, xi, zero, xt, use_mean=False, want_lambda=True)
    return lam, _posterior_variance(model, xi, xt, lam, None, red, return_type)


def kriging_predictor(model, xi, xt, return_type=0):
    """gpmp/core/kriging.py:70-116 (universal kriging) through the Schur complement of the block system."""
    xi, xt = gnp.asarray(xi), gnp.asarray(xt)
    zero = gnp.zeros((xi.shape[0],))
    _, red, lam, mu = _run(model, xi, zero, xt, use_mean=True, want_lambda=True)
    return lam, _posterior_variance(model, xi, xt, lam, mu, red, return_type)


def _posterior_variance(model, xi, xt, lam, mu, red, return_type):
    """gpmp/core/kriging.py:170-199."""
    if return_type == -1:
        return None
    if return_type == 0:
        return _prior_variance(model, xt) - red
    if return_type == 1:
        Ktt = gnp.asarray(model.covariance(xt, None, model.covparam, pairwise=False))
        Kit = gnp.asarray(model.covariance(xi, xt, model.covparam))
        cov = Ktt - gnp.matmul(lam.T.contiguous(), Kit)
        if mu is not None:
            Pt = _mean_values(model, xt, model.meanparam)
            cov = cov - mu.T @ Pt.T
        return cov
    raise ValueError("return_type must be in {-1, 0, 1}")


def select_predictor(model, xi, zi, xt, return_lambdas=True):
    """gpmp/core/kriging.py:119-164.

    Returns (zi_centered, zt_prior_mean, lambda_t, zt_posterior_variance, zt_kriging_mean); lambda_t is
    None unless ``return_lambdas`` -- the posterior mean lambda_t^T zi_centered is returned directly.
    """
    zt_prior_mean = 0.0
    zi_centered = zi
    if model.meantype == "zero":
        use_mean = False
    elif model.meantype == "linear_predictor":
        use_mean = True
    elif model.meantype == "parameterized":
        if model.meanparam is None:
            raise ValueError("For meantype 'parameterized', meanparam should not be None.")
        use_mean = False
        zi_centered = zi - _mean_values(model, xi, model.meanparam).reshape(-1)
        zt_prior_mean = _mean_values(model, xt, model.meanparam).reshape(-1)
    else:
        raise ValueError(
            f"Invalid meantype {model.meantype}. Supported types are 'zero', 'parameterized', and 'linear_predictor'."
        )
    mean, red, lam, _ = _run(model, xi, zi_centered, xt, use_mean, return_lambdas)
    var = _prior_variance(model, xt) - red
    return zi_centered, zt_prior_mean, lam, var, mean
