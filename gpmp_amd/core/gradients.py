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
import os

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


# ---- many small problems in one call ------------------------------------------------------------------
BATCH_MAX_N = 4096      # GPMP_BATCH_MAX_N (include/gpmp_hip.h)
BATCH_MAX_Q = 16        # GPMP_BATCH_MAX_Q


def batch_qualifies(model, use_mean=False):
    """Model-level part of the batched route's conditions, decidable BEFORE any batch is moved to the device: a declared
    Matern covariance (the fused Gram / gradient kernels)."""
    return isinstance(model.covariance, MaternCovariance)


def batch_piece_limit(nmax, d, q, want_grad, device=None):
    """Largest number of problems of up to ``nmax`` points whose workspace fits the budget of ONE library call (a quarter
    of the free device memory, at most 32 GiB; GPMP_BATCH_WS_BUDGET_MB overrides): two nmax x nmax matrices per problem
    with gradients, 70 MB at 2048 points, 270 MB at 4096.  0: the shape is outside the batched kernel's limits."""
    lib = _lib.load()
    per_problem = 8 * int(lib.gpmp_batch_ws_elems(int(nmax), int(d), int(q), 1, 1 if want_grad else 0))
    if per_problem == 0:
        return 0
    budget = 1 << 30
    if device is not None and torch.device(device).type == "cuda":
        budget = min(int(torch.cuda.mem_get_info(device)[0]) // 4, 32 << 30)
    if os.environ.get("GPMP_BATCH_WS_BUDGET_MB"):            # (tests: force the piecewise route)
        budget = int(float(os.environ["GPMP_BATCH_WS_BUDGET_MB"]) * (1 << 20))
    return max(1, budget // per_problem)


def batch_values_and_gradients(model, covparams, batches, want_grad=True, use_mean=False, mean_offset=None):
    """Criterion values (and gradients) of B small problems through ONE library call (gpmp_nll_grad_batch: every
    kernel batched over the problems) -- the throughput path behind ``gnp.BatchDifferentiableSelectionCriterion``
    (gpmp/num/torch_backend.py:607-718) and multi-parameter log_prob evaluations (gpmp/mcmc/param_posterior.py:229-278).

    ``batches``: list of (x_b, z_b) device arrays; ``covparams``: one parameter vector (shared) or a (B, ntheta) array.
    ``use_mean``: REML with ``model.mean`` as the linear predictor (q <= 16 columns), else the zero-mean NLL.
    Returns ``(values, grads)`` as NumPy arrays ((B,), (B, ntheta) or None), or ``None`` when the batch does not
    qualify (not a declared Matern covariance, a batch above 4096 points, more than 16 mean columns): the caller then
    evaluates the batches one after the other.  A failed factorisation raises ``HipLinAlgError`` like the array path."""
    cov = model.covariance
    if not isinstance(cov, MaternCovariance) or len(batches) == 0:
        return None
    lib = _lib.load()
    # ONE data set under many parameter vectors (the chains of a sampler: every entry of ``batches`` is the same pair of arrays):
    # the data are passed once with stride 0 -- no replication, no per-problem packing
    one_data = len(batches) > 1 and all(b[0] is batches[0][0] and b[1] is batches[0][1] for b in batches)
    src = batches[:1] if one_data else batches
    xs = [gnp._points(xb) for xb, _ in src]
    zs = [gnp.asarray(zb).reshape(-1) for _, zb in src]
    if mean_offset is not None:
        zs = [z - mean_offset(x) for x, z in zip(xs, zs)]
    B = len(batches)
    ns = [int(x.shape[0]) for x in xs] * (B if one_data else 1)
    d = int(xs[0].shape[1])
    nmax = max(ns)
    if nmax > BATCH_MAX_N or any(int(x.shape[1]) != d for x in xs):
        return None
    Ps, q = None, 0
    if use_mean:
        Ps = [gnp.asarray(_mean_values(model, x, model.meanparam)) for x in xs]
        q = int(Ps[0].shape[1])
        if q > BATCH_MAX_Q or any(int(P.shape[1]) != q for P in Ps):
            return None
    if min(ns) <= q:
        return None
    dev = xs[0].device
    # the workspace grows with B (two nmax x nmax matrices per problem with gradients: 70 MB at 2048 points): a call is cut into
    # pieces whose workspace fits a quarter of the free device memory (at most 32 GiB)
    b_piece = max(1, batch_piece_limit(nmax, d, q, want_grad, dev))
    if B > b_piece:
        th_all = numpy.asarray(covparams, dtype=numpy.float64)
        vals, grads = [], []
        for b0 in range(0, B, b_piece):
            sl = slice(b0, min(B, b0 + b_piece))
            out = batch_values_and_gradients(model, th_all if th_all.ndim == 1 else th_all[sl], batches[sl], want_grad, use_mean, mean_offset)
            if out is None:
                return None
            vals.append(out[0])
            if want_grad:
                grads.append(out[1])
        return numpy.concatenate(vals), (numpy.concatenate(grads, axis=0) if want_grad else None)
    # pack the problems into padded (B, nmax, .) arrays with a constant number of device operations: one stack when the sizes are
    # equal (mini-batches of a loader), one concatenation + one scatter when they are ragged -- a slice assignment per problem cost
    # 2-3 small copy kernels each: 85 % of the GPU time of a call at n = 128, B = 256 (profiles/r5/batch_n128_B256_kernel_stats.csv)
    if one_data or len(set(ns)) == 1:
        X, Z = torch.stack(xs).contiguous(), torch.stack(zs).contiguous()
        Pm = torch.stack(Ps).contiguous() if q else None
    else:
        dest = torch.as_tensor(numpy.concatenate([numpy.arange(nb, dtype=numpy.int64) + b * nmax for b, nb in enumerate(ns)]), device=dev)
        X = torch.zeros((B * nmax, d), dtype=torch.float64, device=dev).index_copy_(0, dest, torch.cat(xs))
        Z = torch.zeros(B * nmax, dtype=torch.float64, device=dev).index_copy_(0, dest, torch.cat(zs))
        Pm = torch.zeros((B * nmax, q), dtype=torch.float64, device=dev).index_copy_(0, dest, torch.cat(Ps)) if q else None
    sx, sz, sp = (0, 0, 0) if one_data else (nmax * d, nmax, nmax * max(q, 1))
    th = numpy.ascontiguousarray(numpy.asarray(covparams, dtype=numpy.float64))
    shared = th.ndim == 1
    ntheta = th.shape[-1]
    if not shared and th.shape[0] != B:
        raise ValueError("covparams must be one vector or one row per problem")
    noise = 1 if cov.noise else 0
    if ntheta != 1 + noise + d:
        raise ValueError("covparam length does not match 1 + noise + d")
    hv = _lib.host_vec(th.reshape(-1))
    import ctypes

    n_host = (ctypes.c_int * B)(*ns)
    ws = torch.empty(int(lib.gpmp_batch_ws_elems(nmax, d, q, B, 1 if want_grad else 0)), dtype=torch.float64, device=dev)
    values = torch.empty(B, dtype=torch.float64, device=dev)
    grads = torch.empty((B, ntheta), dtype=torch.float64, device=dev) if want_grad else None
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    _lib.check(
        lib.gpmp_nll_grad_batch(gnp._ptr(X), sx, gnp._ptr(Z), sz, gnp._ptr(Pm) if q else None, max(q, 1), sp, q,
                                n_host, nmax, d, B, cov.p, hv, 0 if shared else ntheta, noise, gnp._ptr(ws), gnp._ptr(values),
                                gnp._ptr(grads), gnp._ptr(info), gnp._stream()),
        "gpmp_nll_grad_batch",
    )
    info_h = gnp.to_np(info)
    bad = numpy.nonzero(info_h)[0]
    if bad.size:
        k = int(info_h[bad[0]])
        raise gnp.HipLinAlgError(
            f"Matrix is not positive definite: Cholesky factorization failed in batched problem {int(bad[0])} (info={k})")
    return gnp.to_np(values), (gnp.to_np(grads) if want_grad else None)


def _ml_batch(self, covparam, batches, want_grad=True):
    return batch_values_and_gradients(self.model, covparam, batches, want_grad, use_mean=False, mean_offset=self.mean_offset)


def _reml_batch(self, covparam, batches, want_grad=True):
    return batch_values_and_gradients(self.model, covparam, batches, want_grad, use_mean=True)


MLZeroMeanAnalytic.batch_values_and_gradients = _ml_batch
REMLAnalytic.batch_values_and_gradients = _reml_batch
MLZeroMeanAnalytic.batch_qualifies = lambda self: batch_qualifies(self.model)
REMLAnalytic.batch_qualifies = lambda self: batch_qualifies(self.model, use_mean=True)
MLZeroMeanAnalytic.batch_max_points = REMLAnalytic.batch_max_points = BATCH_MAX_N
MLZeroMeanAnalytic.batch_piece_limit = REMLAnalytic.batch_piece_limit = staticmethod(batch_piece_limit)


def _many(self, P, xi, zi, want_grad=False):
    """the criterion at every row of ``P`` on the same data: one batched call with per-problem parameters (the data are
    replicated per row: at most 4096 x d doubles each).  None when the batched driver does not apply."""
    P = numpy.atleast_2d(numpy.asarray(P, dtype=numpy.float64))
    xi, zi = gnp.asarray(xi), gnp.asarray(zi)
    return self.batch_values_and_gradients(P, [(xi, zi)] * P.shape[0], want_grad)


MLZeroMeanAnalytic.many_values_and_gradients = _many
REMLAnalytic.many_values_and_gradients = _many
