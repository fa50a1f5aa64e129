"""Helpers of the prior-informed selection procedures -- counterpart of gpmp/kernel/prior_helpers.py."""
import numpy as np

from .. import num as gnp
from .init import anisotropic_parameters_initial_guess
from .prior_defaults import get_default_prior_hyperparameters, resolve_prior_defaults_for_selection


def _minimum_nonzero_gap_distance_1d(xj):
    """prior_helpers.py:20-28: smallest positive spacing among points in 1-D (inf if none)."""
    xj = np.asarray(xj, dtype=np.float64).reshape(-1)
    if xj.shape[0] < 2:
        return np.inf
    diffs = np.diff(np.sort(xj))
    diffs = diffs[diffs > 0.0]
    return np.min(diffs) if diffs.shape[0] > 0 else np.inf


def compute_logrho_min_from_xi(xi, prior_rho_min_range_factor=None):
    """prior_helpers.py:61-95: max(log(min nonzero gap), log(range * factor)) per component."""
    if prior_rho_min_range_factor is None:
        prior_rho_min_range_factor = get_default_prior_hyperparameters(xi)["rho_min_range_factor"]
    if prior_rho_min_range_factor <= 0:
        raise ValueError("prior_rho_min_range_factor must be strictly positive.")
    x = np.asarray(gnp.to_np(xi), dtype=np.float64)
    d = x.shape[1]
    gaps = np.array([_minimum_nonzero_gap_distance_1d(x[:, j]) for j in range(d)])
    with np.errstate(divide="ignore"):
        logrho_min_gap = np.where(np.isfinite(gaps), np.log(gaps), -np.inf)
    x_range = x.max(axis=0) - x.min(axis=0)
    min_rho = x_range * float(prior_rho_min_range_factor)
    pos = min_rho > 0.0
    logrho_min_range = np.where(pos, np.log(np.where(pos, min_rho, 1.0)), -np.inf)
    return np.maximum(logrho_min_gap, logrho_min_range)


def resolve_covparam0_prior_and_init(model, xi=None, zi=None, dataloader=None, *, covparam0=None, covparam0_prior=None,
                                     covparam0_init=None):
    """prior_helpers.py:98-150: prior anchor and optimiser start (one initial guess shared if both missing)."""
    guess = None
    if covparam0_init is None:
        if covparam0 is not None:
            covparam0_init = covparam0
        else:
            guess = anisotropic_parameters_initial_guess(model, xi, zi, dataloader)
            covparam0_init = guess
    if covparam0_prior is None:
        if covparam0 is not None:
            covparam0_prior = covparam0
        elif guess is not None:
            covparam0_prior = guess
        else:
            covparam0_prior = anisotropic_parameters_initial_guess(model, xi, zi, dataloader)
    tonp = lambda v: np.asarray(gnp.to_np(v), dtype=np.float64).reshape(-1)  # noqa: E731
    return tonp(covparam0_prior), tonp(covparam0_init)


def resolve_covparam0_roles_for_update(model, xi=None, zi=None, dataloader=None, *, covparam0=None, covparam0_prior=None,
                                       covparam0_init=None, warn_covparam0_prior=True):
    """prior_helpers.py:152-218: roles in an update procedure.  Optimiser start: covparam0_init, else covparam0, else
    model.covparam, else an initial guess.  Prior anchor: covparam0_prior, else covparam0 or model.covparam (with a
    warning about the coupling), else the initial guess."""
    import warnings

    guess = None
    if covparam0_init is None:
        if covparam0 is not None:
            covparam0_init = covparam0
        elif model.covparam is not None:
            covparam0_init = model.covparam
        else:
            guess = anisotropic_parameters_initial_guess(model, xi, zi, dataloader)
            covparam0_init = guess
    if covparam0_prior is None:
        if covparam0 is not None:
            if warn_covparam0_prior:
                warnings.warn("covparam0 provided without covparam0_prior in update procedure; using covparam0 as "
                              "covparam0_prior. Pass covparam0_prior explicitly to avoid this coupling.", stacklevel=2)
            covparam0_prior = covparam0
        elif model.covparam is not None:
            if warn_covparam0_prior:
                warnings.warn("covparam0 and covparam0_prior not provided in update procedure; using model.covparam as "
                              "covparam0_prior. Pass covparam0_prior explicitly to avoid this coupling.", stacklevel=2)
            covparam0_prior = model.covparam
        elif guess is not None:
            covparam0_prior = guess
        else:
            covparam0_prior = anisotropic_parameters_initial_guess(model, xi, zi, dataloader)
    tonp = lambda v: np.asarray(gnp.to_np(v), dtype=np.float64).reshape(-1)  # noqa: E731
    return tonp(covparam0_prior), tonp(covparam0_init)


def resolve_logsigma2_logrho_prior_args(*, covparam0_prior, xi=None, dataloader=None, prior_gamma=None,
                                        prior_sigma2_coverage=None, prior_alpha=None, prior_rho_min_range_factor=None,
                                        prior_log_sigma2_0=None, prior_logrho_0=None, prior_logrho_min=None):
    """prior_helpers.py:225-292."""
    prior_gamma, prior_sigma2_coverage, prior_alpha, prior_rho_min_range_factor = resolve_prior_defaults_for_selection(
        xi=xi, dataloader=dataloader, gamma=prior_gamma, sigma2_coverage=prior_sigma2_coverage, alpha=prior_alpha,
        rho_min_range_factor=prior_rho_min_range_factor)
    covparam0_prior = np.asarray(gnp.to_np(covparam0_prior), dtype=np.float64).reshape(-1)
    if prior_log_sigma2_0 is None:
        prior_log_sigma2_0 = covparam0_prior[0]
    prior_logrho_0 = -covparam0_prior[1:] if prior_logrho_0 is None else np.asarray(gnp.to_np(prior_logrho_0), dtype=np.float64)
    if prior_logrho_min is None:
        if xi is None and dataloader is not None and hasattr(dataloader, "dataset"):   # prior_helpers.py:263-273
            ds = dataloader.dataset
            if not hasattr(ds, "x_list"):
                raise ValueError("dataloader.dataset must provide x_list when prior_logrho_min is None.")
            xi = gnp.concatenate(ds.x_list, 0) if isinstance(ds.x_list, list) else ds.x_list
        if xi is None:
            raise ValueError("xi or dataloader.dataset.x_list must be provided when prior_logrho_min is None.")
        prior_logrho_min = compute_logrho_min_from_xi(xi, prior_rho_min_range_factor=prior_rho_min_range_factor)
    prior_logrho_min = np.asarray(gnp.to_np(prior_logrho_min), dtype=np.float64)
    return (prior_gamma, prior_sigma2_coverage, prior_alpha, prior_rho_min_range_factor, float(prior_log_sigma2_0),
            prior_logrho_0, prior_logrho_min)
