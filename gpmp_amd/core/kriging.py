"""Kriging predictors -- counterpart of gpmp/core/kriging.py.

Algorithmic restatement (exact algebra): with K = L L^T,  V = L^-1 K(xi, xt) is the only O(n^2 m)
solve that the posterior mean and variance need,

    zero mean:         mean = V^T (L^-1 z),   var = k(xt,xt) - colsumsq(V)
    linear predictor:  mu = S^-1 (Wp^T V - Pt^T),  S = Wp^T Wp,  Wp = L^-1 P
                       mean = V^T w - mu^T (Wp^T w),  var = k_tt - colsumsq(V) + sum_a mu_a (S mu)_a

instead of the reference's two trsm (kriging.py:62) / the dense (n+q) sysv solve (kriging.py:98-109).
The kriging weights lambda_t = L^-T (V - Wp mu) cost a second solve and are only formed on request.
xt is processed in column chunks so that the n x m_chunk cross-covariance fits the configured budget.

When K has no Cholesky factor (a conditionally positive definite kernel such as a variogram, where the
reference's sysv block solve still answers) or P^T K^-1 P cannot be inverted, universal kriging falls back
to the contrast space (the role of gpmp/core/kriging.py:116,202-257): ``_ContrastPredictor`` below.
"""
import numpy
import torch

from .. import num as gnp
from .utils import mean_values as _mean_values
from ..config import get_config
from .linalg import covariance_factor
from ..num.householder import HouseholderQR as _Reflectors


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
            self.Sinv = gnp.small_spd_inverse(self.S, "P^T K^-1 P")     # q x q, on the host (rank check included)

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
            mu = gnp.matmul(self.Sinv, R)                       # q x m
            mean = mean - gnp.matmul(self.b, mu)
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


class _ContrastPredictor:
    """Universal kriging in the contrast space Null(P^T), for covariances that are only conditionally positive
    definite.  With lambda = Q [beta; a] (first q coordinates fixed by the unbiasedness constraint R^T beta = Pt^T):

        G a = (Q^T Kit)_2 - (Q^T K Q)_21 beta,      G = (Q^T K Q)_22   (= W^T K W, positive definite)
        var = k_tt - 2 lambda^T Kit + lambda^T K lambda

    which is the solution of the reference's block system (kriging.py:98-109).  ``reference_formula=True`` reproduces
    the reference's own contrast route (kriging.py:202-257) instead: it solves G a = (Q^T Kit)_2 without the coupling
    term, takes beta from diag(R) alone (see ``chunk``) and reports k_tt - [lambda; beta]^T [Kit; Pt^T] -- it differs
    from the block solve by O(1) whenever beta != 0 (tests/golden/ref_cpd.npz pins both)."""

    def __init__(self, model, xi, zi_centered, reference_formula=False):
        self.model, self.xi, self.ref = model, xi, reference_formula
        P = _mean_values(model, xi, model.meanparam)
        self.hq = _Reflectors(P)
        q = self.q = self.hq.q
        Kq = gnp.as_matrix(gnp.asarray(model.covariance(xi, xi, model.covparam)), copy=True)
        self.hq.congruence(Kq)
        self.K11 = Kq[:q, :q].clone()
        self.K21 = gnp.as_matrix(Kq[q:, :q], copy=True)                    # (n - q) x q
        self.F = gnp.cholesky_factor(gnp.as_matrix(Kq[q:, q:], copy=True), overwrite=True)
        del Kq
        zq = self.hq.apply(gnp.as_matrix(zi_centered.reshape(-1, 1), copy=True), transpose=True).reshape(-1)
        self.z1, self.z2 = zq[:q].clone(), zq[q:].clone()

    def chunk(self, xt, want_lambda, want_var=True):
        model, q = self.model, self.q
        Kit = gnp.as_matrix(gnp.asarray(model.covariance(self.xi, xt, model.covparam)), copy=True)
        self.hq.apply(Kit, transpose=True)                                  # Q^T Kit
        Pt = _mean_values(model, xt, model.meanparam)                       # m x q
        if self.ref:
            # kriging.py:238 calls solve(Rq.T, Pt.T, assume_a="sym") on a TRIANGULAR matrix: SciPy's symmetric solver
            # reads the upper triangle of Rq.T only, i.e. diag(Rq) -- exact for q = 1, not beyond.  Reproduced as is.
            beta = Pt.T / torch.diagonal(self.hq.R).reshape(-1, 1)
        else:
            beta = gnp.solve_triangular(self.hq.R.T.contiguous(), Pt.T.contiguous(), lower=True)   # q x m
        C1 = Kit[:q].clone()
        rhs = gnp.as_matrix(Kit[q:], copy=True)
        if not self.ref:
            rhs = rhs - gnp.matmul(self.K21, beta)
        a = self.F.solve(rhs)                                               # (n - q) x m
        mean = gnp.matmul(self.z1, beta) + gnp.matmul(self.z2, a)
        # lambda^T Kit and lambda^T K lambda in the rotated coordinates (Q is orthogonal)
        lk = torch.sum(beta * C1, dim=0) + torch.sum(a * Kit[q:], dim=0)
        if self.ref:
            reduction = lk + torch.sum(beta * Pt.T, dim=0)
        else:
            Ga = rhs                                                        # G a = rhs
            lKl = torch.sum(beta * gnp.matmul(self.K11, beta), dim=0) + 2.0 * torch.sum(a * gnp.matmul(self.K21, beta), dim=0) \
                + torch.sum(a * Ga, dim=0)
            reduction = 2.0 * lk - lKl
        lam = None
        if want_lambda:
            Z = gnp.alloc_matrix(self.hq.n, xt.shape[0])
            Z[:q] = beta
            Z[q:] = a
            lam = self.hq.apply(Z, transpose=False)
        return mean, reduction, lam, beta


def _run(model, xi, zi_centered, xt, use_mean, want_lambda, contrast=None):
    n, m = xi.shape[0], xt.shape[0]
    mc = _chunk_cols(n, m)
    if contrast is not None:
        pred = _ContrastPredictor(model, xi, zi_centered, reference_formula=(contrast == "reference"))
    else:
        try:
            pred = _Predictor(model, xi, zi_centered, use_mean, xt_first=xt[: max(mc, 1)] if m > 0 else None)
        except numpy.linalg.LinAlgError:
            # K has no Cholesky factor / P^T K^-1 P is singular: with a mean design the contrast space may still be
            # positive definite (conditionally positive definite kernels); without one the reference raises as well
            if not use_mean:
                raise
            pred = _ContrastPredictor(model, xi, zi_centered)
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
    return mean, red, lam, mu, type(pred) is _ContrastPredictor and not pred.ref


def _prior_variance(model, xt):
    return gnp.asarray(model.covariance(xt, None, model.covparam, pairwise=True)).reshape(-1)


def kriging_predictor_with_zero_mean(model, xi, xt, return_type=0):
    """gpmp/core/kriging.py:35-67 -> (lambda_t, posterior variance | covariance | None)."""
    xi, xt = gnp.asarray(xi), gnp.asarray(xt)
    zero = gnp.zeros((xi.shape[0],))
    _, red, lam, _, _ = _run(model, xi, zero, xt, use_mean=False, want_lambda=True)
    return lam, _posterior_variance(model, xi, xt, lam, None, red, return_type)


def kriging_predictor(model, xi, xt, return_type=0):
    """gpmp/core/kriging.py:70-116 (universal kriging) through the Schur complement of the block system."""
    xi, xt = gnp.asarray(xi), gnp.asarray(xt)
    zero = gnp.zeros((xi.shape[0],))
    _, red, lam, mu, general = _run(model, xi, zero, xt, use_mean=True, want_lambda=True)
    return lam, _posterior_variance(model, xi, xt, lam, mu, red, return_type, general=general)


def _kriging_predictor_nullspace(model, xi, xt, return_type=0):
    """gpmp/core/kriging.py:202-257, formula for formula (see ``_ContrastPredictor``: ``reference_formula``)."""
    xi, xt = gnp.asarray(xi), gnp.asarray(xt)
    zero = gnp.zeros((xi.shape[0],))
    _, red, lam, beta, _ = _run(model, xi, zero, xt, use_mean=True, want_lambda=True, contrast="reference")
    return lam, _posterior_variance(model, xi, xt, lam, beta, red, return_type)


def _posterior_variance(model, xi, xt, lam, mu, red, return_type, general=False):
    """gpmp/core/kriging.py:170-199.  ``general``: lambda came from the contrast-space fallback, where no Lagrange
    multipliers exist; the full covariance is then Ktt - lambda^T Kit - Kit^T lambda + lambda^T K lambda."""
    if return_type == -1:
        return None
    if return_type == 0:
        return _prior_variance(model, xt) - red
    if return_type == 1:
        Ktt = gnp.asarray(model.covariance(xt, None, model.covparam, pairwise=False))
        Kit = gnp.asarray(model.covariance(xi, xt, model.covparam))
        LtK = gnp.matmul(lam, Kit, ta=True)
        if general:
            Kii = gnp.asarray(model.covariance(xi, xi, model.covparam))
            return Ktt - LtK - LtK.T + gnp.matmul(lam, gnp.matmul(Kii, lam), ta=True)
        cov = Ktt - LtK
        if mu is not None:
            Pt = _mean_values(model, xt, model.meanparam)
            cov = cov - gnp.matmul(mu, Pt, ta=True, tb=True)
        return cov
    raise ValueError("return_type must be in {-1, 0, 1}")


def fused_prediction(model, xi, zi, xt):
    """Posterior mean and (unclamped) variance through ONE library call -- gpmp_predict_zero_mean / gpmp_predict_mean
    (include/gpmp_hip.h) -- when the model is a declared Matern covariance with a zero, parameterized or linear-predictor mean
    and the prediction fits one chunk.  Returns ``(kriging mean, variance, prior mean)`` or None: then, and whenever the call
    reports a failed factorisation or a rank-deficient mean design, the general route below runs (it raises the reference's
    errors and knows the contrast-space fallback).  The dozen small launches and host round trips of the general route are
    most of a small prediction (n = 128, m = 500, constant mean: 0.55 ms against 0.3)."""
    import os

    from ..kernel.matern import MaternCovariance

    cov = model.covariance
    if os.environ.get("GPMP_PREDICT_FUSED", "1") == "0" or not isinstance(cov, MaternCovariance):
        return None
    if model.meantype not in ("zero", "parameterized", "linear_predictor") or model.covparam is None:
        return None
    if not (isinstance(xi, torch.Tensor) and isinstance(xt, torch.Tensor)) or xt is xi:     # (xt is xi: the reference's identity
        return None                                                                        #  dispatch puts the nugget on K(xi, xt))
    n, m = int(xi.shape[0]), int(xt.shape[0])
    if n < 1 or m < 1 or xi.dim() != 2 or xt.dim() != 2 or _chunk_cols(n, m) < m:
        return None
    d = int(xi.shape[1])
    lib = gnp._lib.load()
    theta = gnp._host_params(model.covparam)
    if d > 64 or cov.p > 16 or len(theta) != 1 + (1 if cov.noise else 0) + d:
        return None
    zt_prior_mean = 0.0
    zc = gnp.asarray(zi).reshape(-1)
    Pi = Pt = None
    if model.meantype == "parameterized":
        if model.meanparam is None:
            return None
        zc = zc - _mean_values(model, xi, model.meanparam).reshape(-1)
        zt_prior_mean = _mean_values(model, xt, model.meanparam).reshape(-1)
    elif model.meantype == "linear_predictor":
        Pi = gnp.as_matrix(_mean_values(model, xi, model.meanparam))
        Pt = gnp.as_matrix(_mean_values(model, xt, model.meanparam))
        q = int(Pi.shape[1])
        if q < 1 or q > 71 or n <= q or int(Pt.shape[1]) != q:
            return None
    X, Xt, zc = gnp._points(xi), gnp._points(xt), zc.contiguous()
    dev = X.device
    zpm = torch.empty(m, dtype=torch.float64, device=dev)
    zpv = torch.empty(m, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    hv = gnp._lib.host_vec(theta)
    noise = 1 if cov.noise else 0
    if Pi is None:
        ws = torch.empty(int(lib.gpmp_predict_ws_elems(n, m)), dtype=torch.float64, device=dev)
        gnp._lib.check(lib.gpmp_predict_zero_mean(gnp._ptr(X), gnp._ptr(zc), gnp._ptr(Xt), n, m, d, cov.p, hv, noise, 0, gnp._ptr(ws),
                                                  gnp._ptr(zpm), gnp._ptr(zpv), gnp._ptr(info), gnp._stream()), "gpmp_predict_zero_mean")
    else:
        ws = torch.empty(int(lib.gpmp_predict_mean_ws_elems(n, m, q)), dtype=torch.float64, device=dev)
        gnp._lib.check(lib.gpmp_predict_mean(gnp._ptr(X), gnp._ptr(zc), gnp._ptr(Pi), gnp._ld(Pi), gnp._ptr(Xt), gnp._ptr(Pt), gnp._ld(Pt),
                                             n, m, d, q, cov.p, hv, noise, 0, gnp._ptr(ws), gnp._ptr(zpm), gnp._ptr(zpv), gnp._ptr(info),
                                             gnp._stream()), "gpmp_predict_mean")
    if int(info.item()) != 0:
        return None
    return zpm, zpv, zt_prior_mean


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
    mean, red, lam, _, _ = _run(model, xi, zi_centered, xt, use_mean, return_lambdas)
    var = _prior_variance(model, xt) - red
    return zi_centered, zt_prior_mean, lam, var, mean
