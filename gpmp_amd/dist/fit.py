"""Covariance-parameter selection on the block-cyclic factor: ML / REML fits at n beyond one GPU's HBM.

Counterpart of the single-GPU flow ``make_selection_criterion_with_gradient`` + ``autoselect_parameters``
(gpmp/kernel/parameter_selection.py:35-124, 253-260: SciPy drives a criterion and its gradient) with the criterion
evaluated by ``BlockCyclicCholesky.value_and_grad``: per evaluation one local Gram build (no communication), one
distributed factorisation, one value + analytic gradient.  Every rank runs the SAME SciPy iteration: values and gradients
come out of all-reduces, so they are bit-identical on all ranks and the optimisers stay in lockstep without a broadcast.
"""
from __future__ import annotations

import math

import numpy as np

from .cholesky import BlockCyclicCholesky


def distributed_criterion(grid, cov, x, z, P=None, nb=1024, ops=None, transport=None, p=None, noise=None, factor_class=None):
    """-> f(covparam) = (value, gradient) on the grid.  ``cov``: a covariance callable (``MaternCovariance``: p and the noise
    flag are read from it; any other callable needs ``p`` / ``noise``), x: (n, d) and z: (n,) replicated host arrays, P: the
    (n, q) mean design for REML or None for the zero-mean likelihood.  A failed factorisation gives (+inf, 0): the reference's
    ``evaluate_pre_grad`` convention (gpmp/num/numpy_backend.py:344-350)."""
    x = np.asarray(x, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64).reshape(-1)
    n = x.shape[0]
    p = int(getattr(cov, "p", p))
    noise = bool(getattr(cov, "noise", noise))
    eps = float(np.finfo(np.float64).eps)
    Factor = BlockCyclicCholesky if factor_class is None else factor_class

    def f(covparam):
        th = np.asarray(covparam, dtype=np.float64)
        diag = math.exp(th[1]) if noise else 10.0 * math.exp(th[0]) * eps          # matern.py:90 / the noise variance
        ch = Factor(grid, n, nb=nb, ops=ops, transport=transport)
        ch.build_local_gram(cov, x, th, diag)
        if ch.factor() != 0:
            return math.inf, np.zeros_like(th)
        return ch.value_and_grad(x, z, th, p, noise=noise, P=P)

    return f


def fit_covparam(grid, cov, x, z, covparam0, P=None, bounds=None, method="L-BFGS-B", options=None, **kw):
    """Minimise the ML / REML criterion over the covariance parameters with SciPy (same defaults as the single-GPU
    ``autoselect_parameters``: ftol 1e-6, maxiter 15000, gpmp/kernel/parameter_selection.py:253-260).  Returns
    (covparam, info) with info = SciPy's OptimizeResult fields + the evaluation history; identical on every rank."""
    from scipy.optimize import minimize

    f = distributed_criterion(grid, cov, x, z, P=P, **kw)
    history = []

    def fun(th):
        v, g = f(th)
        history.append((np.array(th, copy=True), float(v)))
        if not math.isfinite(v):
            return 1e300, np.zeros_like(th)        # SciPy's line search needs a number: back off from the failed point
        return float(v), np.asarray(g, dtype=np.float64)

    opts = {"ftol": 1e-6, "maxiter": 15000}
    opts.update(options or {})
    res = minimize(fun, np.asarray(covparam0, dtype=np.float64), jac=True, method=method, bounds=bounds, options=opts)
    info = {"fun": float(res.fun), "nfev": int(res.nfev), "nit": int(res.nit), "success": bool(res.success), "message": str(res.message),
            "history": history}
    return np.asarray(res.x), info
