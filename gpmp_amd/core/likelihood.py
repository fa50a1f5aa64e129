"""Negative (restricted) log-likelihoods -- counterpart of gpmp/core/likelihood.py."""
import math

import numpy

from .. import num as gnp
from .utils import mean_values as _mean_values
from .linalg import MeanSpace, covariance_factor


def _scalar(v):
    return numpy.asarray(float(v), dtype=numpy.float64).reshape(())


def negative_log_likelihood_zero_mean(model, covparam, xi, zi):
    """gpmp/core/likelihood.py:18-52: 1/2 (n ln 2pi + 2 sum ln L_ii + z^T K^-1 z)."""
    xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
    n = xi.shape[0]
    # a non-PD covariance raises HipLinAlgError (a numpy LinAlgError), as numpy.linalg.cholesky does under the reference's
    # NumPy backend; the criterion wrappers (gnp.DifferentiableSelectionCriterion) turn it into +inf.  Library / HIP
    # failures (GpmpHipError, out of memory) propagate: they are not numerical events.
    F = covariance_factor(model, xi, covparam)
    w = F.solve_lower(zi)
    norm2 = float(gnp.sum(w * w).item())
    return _scalar(0.5 * (n * math.log(2.0 * math.pi) + F.logdet() + norm2))


def negative_log_likelihood(model, meanparam, covparam, xi, zi):
    """gpmp/core/likelihood.py:55-89."""
    xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
    centered = zi - _mean_values(model, xi, meanparam).reshape(-1)
    return negative_log_likelihood_zero_mean(model, covparam, xi, centered)


def negative_log_restricted_likelihood(model, covparam, xi, zi):
    """gpmp/core/likelihood.py:92-129 without the n x n Q / W^T K W (identities in core/linalg.py)."""
    xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
    # a non-PD covariance raises HipLinAlgError (a numpy LinAlgError), as numpy.linalg.cholesky does under the reference's
    # NumPy backend; the criterion wrappers (gnp.DifferentiableSelectionCriterion) turn it into +inf.  Library / HIP
    # failures (GpmpHipError, out of memory) propagate: they are not numerical events.
    F = covariance_factor(model, xi, covparam)
    P = _mean_values(model, xi, model.meanparam)
    n, q = P.shape
    ms = MeanSpace(F, zi, P)
    return _scalar(0.5 * ((n - q) * math.log(2.0 * math.pi) + ms.logdet_contrast() + ms.quad()))
