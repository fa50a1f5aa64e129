"""Priors added to the REML criterion (REMAP) -- counterpart of gpmp/kernel/priors.py.

These are O(d) scalar terms on the host covparam vector; the O(n^3) REML value and gradient they are added
to come from the HIP path.  Every prior here also returns its gradient with respect to covparam, so the
REMAP criteria keep an analytic gradient (the reference differentiates them by autograd).
"""
import math
from statistics import NormalDist

import numpy as np

from .. import num as gnp
from .prior_defaults import get_default_prior_hyperparameters


def _resolve_prior_defaults(gamma=None, sigma2_coverage=None, alpha=None, xi=None):
    """priors.py:166-174."""
    d = get_default_prior_hyperparameters(xi)
    return (d["gamma"] if gamma is None else gamma,
            d["sigma2_coverage"] if sigma2_coverage is None else sigma2_coverage,
            d["alpha"] if alpha is None else alpha)


def _logsigma2_prior_std(gamma, sigma2_coverage):
    """priors.py:177-188: std in log-space such that P(s2_0/gamma <= s2 <= s2_0 gamma) = coverage."""
    if gamma <= 1.0:
        raise ValueError("gamma must be > 1.")
    if not (0.0 < sigma2_coverage < 1.0):
        raise ValueError("sigma2_coverage must be in (0, 1).")
    zq = NormalDist().inv_cdf(0.5 * (1.0 + sigma2_coverage))
    if zq <= 0.0:
        raise ValueError("Invalid sigma2_coverage: non-positive Gaussian quantile.")
    return math.log(gamma) / zq


def _vec(covparam):
    from .. import num as gnp

    return np.asarray(gnp.to_np(covparam), dtype=np.float64).reshape(-1)


def log_prior_jeffreys_variance(covparam, lambda_var=1.0):
    """priors.py:43-70."""
    return -lambda_var * _vec(covparam)[0]


def log_prior_power_law(covparam, lambda_var=1.0, cut_logvariance_high=9.21, lambda_lengthscales=0.0,
                        cut_loginvrho_low=-9.21, cut_loginvrho_high=9.21, penalty_factor=100):
    """priors.py:73-128 (verbatim arithmetic, including the sign of the variance cutoff term)."""
    th = _vec(covparam)
    log_sigma2, p = th[0], th[1:]
    log_prior_sigma2 = -lambda_var * log_sigma2
    extra_sigma2 = penalty_factor * max(log_sigma2 - cut_logvariance_high, 0.0)
    extra_low = penalty_factor * np.maximum(cut_loginvrho_low - p, 0)
    extra_high = penalty_factor * np.maximum(p - cut_loginvrho_high, 0)
    log_prior_lengths = -lambda_lengthscales * np.sum(p) - np.sum(extra_low) - np.sum(extra_high)
    return log_prior_sigma2 + extra_sigma2 + log_prior_lengths


def log_prior_reference(model, covparam, xi):
    """priors.py:131-166: reference prior, 1/2 log det of the Fisher information of the covariance parameters."""
    info = np.asarray(gnp.to_np(model.fisher_information(xi, covparam)), dtype=np.float64)
    sign, logabs = np.linalg.slogdet(info)
    if sign <= 0:
        raise np.linalg.LinAlgError("Fisher information is not positive definite")
    return 0.5 * logabs


def log_prior_gaussian_logsigma2(covparam, log_sigma2_0, gamma=None, sigma2_coverage=None):
    """priors.py:191-231: -1/2 ((log s2 - log s2_0) / std)^2."""
    gamma, sigma2_coverage, _ = _resolve_prior_defaults(gamma=gamma, sigma2_coverage=sigma2_coverage)
    std = _logsigma2_prior_std(gamma, sigma2_coverage)
    z = (_vec(covparam)[0] - float(log_sigma2_0)) / std
    return -0.5 * z * z


def neglog_f_logrho(logrho, logrho_min, logrho_0, alpha=None):
    """priors.py:234-271: barrier + linear tail, +inf where logrho <= logrho_min, minimum at logrho_0."""
    _, _, alpha = _resolve_prior_defaults(alpha=alpha)
    if alpha <= 0:
        raise ValueError("alpha must be > 0.")
    logrho, logrho_min, logrho_0 = (np.asarray(v, dtype=np.float64) for v in (logrho, logrho_min, logrho_0))
    if np.any(logrho_0 <= logrho_min):
        raise ValueError("logrho_0 must be > logrho_min (componentwise).")
    beta = alpha
    alpha_eff = beta * (logrho_0 - logrho_min)
    shifted = logrho - logrho_min
    mask = shifted > 0.0
    safe = np.where(mask, shifted, 1.0)
    return np.where(mask, -alpha_eff * np.log(safe) + beta * safe, np.inf)


def log_prior_logrho_barrier_linear(covparam, logrho_min, logrho_0, alpha=None):
    """priors.py:274-302: logrho = -covparam[1:]."""
    return -np.sum(neglog_f_logrho(-_vec(covparam)[1:], logrho_min, logrho_0, alpha=alpha))


# ---- gradients of the NEGATIVE log-priors with respect to covparam (added to the REML gradient) ----
def grad_neg_log_prior_gaussian_logsigma2(covparam, log_sigma2_0, gamma=None, sigma2_coverage=None):
    gamma, sigma2_coverage, _ = _resolve_prior_defaults(gamma=gamma, sigma2_coverage=sigma2_coverage)
    std = _logsigma2_prior_std(gamma, sigma2_coverage)
    th = _vec(covparam)
    g = np.zeros_like(th)
    g[0] = (th[0] - float(log_sigma2_0)) / (std * std)
    return g


def grad_neg_log_prior_logrho_barrier_linear(covparam, logrho_min, logrho_0, alpha=None):
    _, _, alpha = _resolve_prior_defaults(alpha=alpha)
    th = _vec(covparam)
    logrho_min, logrho_0 = np.asarray(logrho_min, dtype=np.float64), np.asarray(logrho_0, dtype=np.float64)
    shifted = -th[1:] - logrho_min
    alpha_eff = alpha * (logrho_0 - logrho_min)
    g = np.zeros_like(th)
    with np.errstate(divide="ignore", invalid="ignore"):
        dn_dlogrho = np.where(shifted > 0.0, -alpha_eff / shifted + alpha, 0.0)
    g[1:] = -dn_dlogrho            # d logrho / d covparam = -1
    return g


# ---- posterior objective wrappers (priors.py:305-558) ----
def neg_log_restricted_posterior_with_jeffreys_prior(model, covparam, xi, zi, lambda_var=1.0):
    return model.negative_log_restricted_likelihood(covparam, xi, zi) - log_prior_jeffreys_variance(covparam, lambda_var)


def neg_log_restricted_posterior_power_laws_prior(model, covparam, xi, zi):
    return model.negative_log_restricted_likelihood(covparam, xi, zi) - log_prior_power_law(covparam)


def neg_log_restricted_posterior_logsigma2_prior(model, covparam, xi, zi, log_sigma2_0, gamma=None, sigma2_coverage=None):
    return model.negative_log_restricted_likelihood(covparam, xi, zi) - log_prior_gaussian_logsigma2(
        covparam, log_sigma2_0, gamma=gamma, sigma2_coverage=sigma2_coverage)


def neg_log_restricted_posterior_with_logrho_prior(model, covparam, xi, zi, logrho_min, logrho_0, alpha=None):
    return model.negative_log_restricted_likelihood(covparam, xi, zi) - log_prior_logrho_barrier_linear(
        covparam, logrho_min=logrho_min, logrho_0=logrho_0, alpha=alpha)


def neg_log_restricted_posterior_logsigma2_and_logrho_prior(model, covparam, xi, zi, log_sigma2_0, gamma=None,
                                                            sigma2_coverage=None, logrho_min=None, logrho_0=None,
                                                            alpha=None):
    """priors.py:467-558: REML - log p(log s2) - log p(logrho)."""
    if logrho_min is None or logrho_0 is None:
        raise ValueError("logrho_min and logrho_0 must be provided.")
    gamma, sigma2_coverage, alpha = _resolve_prior_defaults(gamma=gamma, sigma2_coverage=sigma2_coverage, alpha=alpha, xi=xi)
    nlrl = model.negative_log_restricted_likelihood(covparam, xi, zi)
    return (nlrl - log_prior_gaussian_logsigma2(covparam, log_sigma2_0, gamma=gamma, sigma2_coverage=sigma2_coverage)
            - log_prior_logrho_barrier_linear(covparam, logrho_min=logrho_min, logrho_0=logrho_0, alpha=alpha))
