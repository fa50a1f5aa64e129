"""``DistributedModel``: the reference's ``Model`` surface (gpmp/core/model.py:136-343) on the 2-D block-cyclic factor.

Same constructor and the same method names, argument meaning and error behaviour as ``gpmp_amd.Model`` /
``gpmp.core.Model`` for ``predict``, ``loo`` and the three likelihood criteria, for problems whose K(xi, xi) does not fit
one GPU's HBM: every rank of the process grid makes the SAME call with the SAME (replicated) host arrays and gets the
same full-length NumPy results back.  What differs from the single-GPU model: a covariance callable is evaluated on
blocks (cross-covariance calls on subsets of the points), the diagonal term -- the nugget 10 sigma^2 eps of
gpmp/kernel/matern.py:90, or the noise variance of a ``MaternCovariance(noise=True)`` -- is added by the model, and
``return_lambdas`` (round 4: the second solve of cholesky_solve on the block-cyclic factor, ``solve_upper_many``) comes either as
the full n x m host matrix on every rank or, ``return_lambdas="local"``, as this rank's block.
"""
from __future__ import annotations

import math
import warnings

import numpy as np
import torch.distributed as dist

from .cholesky import BlockCyclicCholesky
from .fit import fit_covparam


class DistributedModel:
    def __init__(self, grid, mean, covariance, meanparam=None, covparam=None, meantype="linear_predictor", nb=1024, ops=None,
                 transport=None, factor_class=None):
        if meantype not in ("zero", "parameterized", "linear_predictor"):
            raise ValueError("meantype must be one of 'zero', 'parameterized', 'linear_predictor'")       # core/utils.py:84-118
        if (meantype == "zero") != (mean is None):
            raise ValueError("mean must be None exactly when meantype is 'zero'")
        self.grid, self.mean, self.covariance = grid, mean, covariance
        self.meanparam, self.covparam, self.meantype = meanparam, covparam, meantype
        self.nb, self.ops, self.transport = nb, ops, transport
        # the factorisation class: BlockCyclicCholesky, or a subclass with another transport under the same schedule (tests:
        # tools/thread_ranks.py runs the ranks as threads of one process)
        self.factor_class = BlockCyclicCholesky if factor_class is None else factor_class
        self._cache = None                    # (key, factor): predict after a criterion at the same parameters re-uses nothing
                                              # implicitly -- the cache holds the factor of the LAST (xi, covparam) only

    # ---- pieces
    def _diag(self, covparam):
        th = np.asarray(covparam, dtype=np.float64)
        if bool(getattr(self.covariance, "noise", False)):
            return math.exp(th[1])
        return 10.0 * math.exp(th[0]) * float(np.finfo(np.float64).eps)

    def _factor(self, xi, covparam):
        xi = np.ascontiguousarray(np.asarray(xi, dtype=np.float64))
        th = np.asarray(covparam, dtype=np.float64)
        key = (xi.shape, hash(xi.tobytes()), th.tobytes())
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        ch = self.factor_class(self.grid, xi.shape[0], nb=self.nb, ops=self.ops, transport=self.transport)
        ch.build_local_gram(self.covariance, xi, th, self._diag(th))
        ch.factor()
        self._cache = (key, ch)
        return ch

    def _design(self, x, meanparam=None):
        P = np.asarray(self.mean(x, self.meanparam if meanparam is None else meanparam), dtype=np.float64)
        return P.reshape(len(x), -1)

    def _gather(self, idx, *shards):
        """shards of length len(idx) held by every process column -> full-length arrays on every rank"""
        g = self.grid
        parts = [None] * g.pc
        dist.all_gather_object(parts, (np.asarray(idx), [np.asarray(s) for s in shards]), group=g.row_group)
        n = int(sum(len(p[0]) for p in parts))
        out = [np.empty(n) for _ in shards]
        for ix, vals in parts:
            for o, v in zip(out, vals):
                o[ix] = v
        return out

    # ---- the Model surface
    def predict(self, xi, zi, xt, return_lambdas=False, zero_neg_variances=True):
        """Posterior mean and variance at xt (gpmp/core/model.py:227-307): zero mean, parameterized mean (centred
        observations + prior mean at xt) or linear predictor (universal kriging).  ``return_lambdas`` (model.py:305-306):
        True appends the kriging weights lambda_t as the FULL n x m NumPy matrix on every rank (an all-gather of n x m
        doubles: meant for the sizes a host array holds); "local" appends this rank's block instead, as
        (device tensor, global row indices, (j0, j1)) -- the form that scales."""
        xi, xt = np.asarray(xi, dtype=np.float64), np.asarray(xt, dtype=np.float64)
        zi = np.asarray(zi, dtype=np.float64).reshape(-1)
        ch = self._factor(xi, self.covparam)
        prior = 0.0
        P = Pt = None
        if self.meantype == "parameterized":
            zi = zi - self._design(xi).reshape(-1)
            prior = self._design(xt).reshape(-1)
        elif self.meantype == "linear_predictor":
            P, Pt = self._design(xi), self._design(xt)
        out = ch.predict(self.covariance, xi, zi, xt, self.covparam, P=P, Pt=Pt, return_lambdas=bool(return_lambdas))
        mean, var, (j0, j1) = out[:3]
        zpm, zpv = self._gather(np.arange(j0, j1), mean, var)
        if np.any(zpv < 0.0):
            warnings.warn("Negative variances detected. Consider using jitter.", RuntimeWarning)       # model.py:290-296
        if zero_neg_variances:
            zpv = np.maximum(zpv, 0.0)
        if return_lambdas:
            lam, ri = out[3], ch.global_row_index()
            if return_lambdas == "local":
                return zpm + prior, zpv, (lam, ri, (j0, j1))
            parts = [None] * self.grid.world
            dist.all_gather_object(parts, (np.asarray(ri), (j0, j1), lam.cpu().numpy()), group=self.grid.world_group)
            full = np.zeros((xi.shape[0], xt.shape[0]))
            for rows, (a, b), blk in parts:
                if len(rows) and b > a:
                    full[np.ix_(rows, np.arange(a, b))] = blk
            return zpm + prior, zpv, full
        return zpm + prior, zpv

    def loo(self, xi, zi):
        """(zloo, sigma2loo, eloo) -- gpmp/core/loo.py:21-130."""
        xi = np.asarray(xi, dtype=np.float64)
        zi = np.asarray(zi, dtype=np.float64).reshape(-1)
        ch = self._factor(xi, self.covparam)
        prior = 0.0
        P = None
        if self.meantype == "parameterized":
            prior = self._design(xi).reshape(-1)
            zi = zi - prior
        elif self.meantype == "linear_predictor":
            P = self._design(xi)
        zloo, s2, eloo, idx = ch.loo(zi, P)
        zloo, s2, eloo = self._gather(idx, zloo, s2, eloo)
        return zloo + prior, s2, eloo

    def negative_log_likelihood_zero_mean(self, covparam, xi, zi):
        """gpmp/core/likelihood.py:18-52; +inf when K has no Cholesky factor (likelihood.py:47-48)."""
        return self._factor(xi, covparam).negative_log_likelihood(np.asarray(zi, dtype=np.float64).reshape(-1))

    def negative_log_likelihood(self, meanparam, covparam, xi, zi):
        """gpmp/core/likelihood.py:55-89 (parameterized mean: the centred observations through the zero-mean criterion)."""
        zc = np.asarray(zi, dtype=np.float64).reshape(-1) - self._design(np.asarray(xi, dtype=np.float64), meanparam).reshape(-1)
        return self._factor(xi, covparam).negative_log_likelihood(zc)

    def negative_log_restricted_likelihood(self, covparam, xi, zi):
        """gpmp/core/likelihood.py:92-129."""
        xi = np.asarray(xi, dtype=np.float64)
        return self._factor(xi, covparam).negative_log_restricted_likelihood(np.asarray(zi, dtype=np.float64).reshape(-1), self._design(xi))

    def select_parameters(self, xi, zi, covparam0=None, criterion=None, bounds=None, options=None):
        """ML (zero / parameterized mean) or REML (linear predictor) selection of the covariance parameters with SciPy over the
        distributed value + analytic gradient (gpmp/kernel/parameter_selection.py:253-260); sets and returns ``covparam``."""
        xi = np.asarray(xi, dtype=np.float64)
        zi = np.asarray(zi, dtype=np.float64).reshape(-1)
        P = None
        if self.meantype == "parameterized":
            zi = zi - self._design(xi).reshape(-1)
        elif self.meantype == "linear_predictor" and criterion != "ml":
            P = self._design(xi)
        th0 = self.covparam if covparam0 is None else covparam0
        self.covparam, info = fit_covparam(self.grid, self.covariance, xi, zi, th0, P=P, bounds=bounds, options=options, nb=self.nb,
                                           ops=self.ops, transport=self.transport, factor_class=self.factor_class)
        self._cache = None
        return self.covparam, info
