"""Data-driven parameter bounds -- counterpart of gpmp/kernel/bounds.py."""
import numpy as np

from .. import num as gnp
from .prior_helpers import _minimum_nonzero_gap_distance_1d


def empirical_bounds_factory(xi, zi, *, mean_paramlength=0, var_lower_factor=2.0, var_upper_factor=10.0, length_lower_factor=2.0):
    """bounds.py:22-46: rows (lower, upper) for [mean..., log sigma^2, -log rho_0 .. -log rho_{d-1}]: the variance within
    [var_lower_factor, var_upper_factor] x the empirical variance of zi, each length-scale at least length_lower_factor x the
    smallest non-zero gap of that coordinate."""
    x = np.asarray(gnp.to_np(xi), dtype=np.float64)
    z = np.asarray(gnp.to_np(zi), dtype=np.float64).reshape(-1)
    rows = [(-np.inf, np.inf)] * int(mean_paramlength)
    v = float(np.var(z))
    rows.append((np.log(var_lower_factor * v), np.log(var_upper_factor * v)))
    for j in range(x.shape[1]):
        gap = float(_minimum_nonzero_gap_distance_1d(x[:, j]))
        rows.append((-np.inf, -np.log(length_lower_factor * gap)) if np.isfinite(gap) else (-np.inf, np.inf))
    return gnp.asarray(np.array(rows, dtype=np.float64))
