"""Default hyper-parameters of the prior models -- counterpart of gpmp/kernel/prior_defaults.py."""
from dataclasses import dataclass


@dataclass
class _PriorDefaults:
    gamma: float = 1.5
    sigma2_coverage: float = 0.95
    alpha: float = 1.0
    rho_min_range_factor: float = 1 / 20.0


_PRIOR_DEFAULTS = _PriorDefaults()


def get_default_prior_hyperparameters(xi=None):
    """prior_defaults.py:36-60 (dataset-agnostic defaults; xi only shape-checked)."""
    if xi is not None and hasattr(xi, "shape") and len(tuple(xi.shape)) != 2:
        raise ValueError("xi must have shape (n, d).")
    d = _PRIOR_DEFAULTS
    return {"gamma": d.gamma, "sigma2_coverage": d.sigma2_coverage, "alpha": d.alpha,
            "rho_min_range_factor": d.rho_min_range_factor}


def set_default_prior_hyperparameters(*, gamma=None, sigma2_coverage=None, alpha=None, rho_min_range_factor=None):
    """prior_defaults.py:63-107 (same validation rules)."""
    if gamma is not None:
        if float(gamma) <= 1.0:
            raise ValueError("gamma must be > 1.")
        _PRIOR_DEFAULTS.gamma = float(gamma)
    if sigma2_coverage is not None:
        if not (0.0 < float(sigma2_coverage) < 1.0):
            raise ValueError("sigma2_coverage must be in (0, 1).")
        _PRIOR_DEFAULTS.sigma2_coverage = float(sigma2_coverage)
    if alpha is not None:
        if float(alpha) <= 0.0:
            raise ValueError("alpha must be > 0.")
        _PRIOR_DEFAULTS.alpha = float(alpha)
    if rho_min_range_factor is not None:
        if float(rho_min_range_factor) <= 0.0:
            raise ValueError("rho_min_range_factor must be > 0.")
        _PRIOR_DEFAULTS.rho_min_range_factor = float(rho_min_range_factor)


def set_default_prior_hyperparameters_from_kwargs(kwargs):
    """prior_defaults.py:115-135: pop the ``prior_*`` convenience keys of a kwargs dict into the defaults."""
    for key, name in (("prior_logsigma2_gamma", "gamma"), ("prior_logsigma2_coverage", "sigma2_coverage"),
                      ("prior_logrho_alpha", "alpha"), ("prior_logrho_min_range_factor", "rho_min_range_factor")):
        if key in kwargs:
            set_default_prior_hyperparameters(**{name: kwargs.pop(key)})


def resolve_prior_defaults_for_selection(xi=None, dataloader=None, gamma=None, sigma2_coverage=None, alpha=None,
                                         rho_min_range_factor=None):
    """prior_defaults.py:137-175."""
    d = get_default_prior_hyperparameters(xi)
    return (d["gamma"] if gamma is None else gamma,
            d["sigma2_coverage"] if sigma2_coverage is None else sigma2_coverage,
            d["alpha"] if alpha is None else alpha,
            d["rho_min_range_factor"] if rho_min_range_factor is None else rho_min_range_factor)
