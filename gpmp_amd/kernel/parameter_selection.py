"""Selection criteria and the SciPy driver -- counterpart of the parts of
gpmp/kernel/parameter_selection.py that configs 1 and 4 exercise:
``make_selection_criterion_with_gradient`` (:35-124), ``autoselect_parameters`` (:128-276),
``select_parameters_with_criterion`` (:280-437) and ``select_parameters_with_reml`` (:747-808).
The optimiser stays SciPy on the host; each evaluation is one Gram build + Cholesky (+ gradient) on
the GPU.  The REMAP family is out of scope (host-side priors, section 8f).
"""
import time

import numpy as np
from scipy.optimize import minimize

from .. import num as gnp
from .init import anisotropic_parameters_initial_guess, anisotropic_parameters_initial_guess_constant_mean
from .matern import MaternCovariance
from .prior_defaults import resolve_prior_defaults_for_selection
from .prior_helpers import (resolve_covparam0_prior_and_init, resolve_covparam0_roles_for_update,
                            resolve_logsigma2_logrho_prior_args)
from . import priors as _priors


# criteria with the reference's call signatures (parameter_selection.py re-exports of core.likelihood)
def negative_log_likelihood_zero_mean(model, covparam, xi, zi):
    return model.negative_log_likelihood_zero_mean(covparam, xi, zi)


def negative_log_likelihood(model, meanparam, covparam, xi, zi):
    return model.negative_log_likelihood(meanparam, covparam, xi, zi)


def negative_log_restricted_likelihood(model, covparam, xi, zi):
    return model.negative_log_restricted_likelihood(covparam, xi, zi)


def _analytic_for(model, selection_criterion, parameterized_mean):
    if parameterized_mean or not isinstance(model.covariance, MaternCovariance):
        return None
    from ..core import gradients as _gradients  # deferred: core imports kernel.matern

    factory = getattr(selection_criterion, "_gpmp_analytic_factory", None)
    if factory is not None:          # REML + host-side priors (REMAP): analytic REML gradient + prior gradient
        return factory(model)

    if selection_criterion is negative_log_likelihood_zero_mean:
        return _gradients.MLZeroMeanAnalytic(model)
    if selection_criterion is negative_log_restricted_likelihood:
        return _gradients.REMLAnalytic(model)
    return None


def make_selection_criterion_with_gradient(model, selection_criterion, xi=None, zi=None, dataloader=None,
                                           batches_per_eval=0, parameterized_mean=False, meanparam_len=1):
    """gpmp/kernel/parameter_selection.py:35-124 -> (evaluate, evaluate_pre_grad, evaluate_no_grad, gradient).

    ``gradient`` is analytic for the library's ML / REML criteria on a ``MaternCovariance`` and None
    otherwise (SciPy then differentiates numerically, as with the reference's NumPy backend).
    """
    arrays = xi is not None and zi is not None
    if arrays and dataloader is not None:
        raise ValueError("Provide either (xi, zi) or dataloader, not both.")   # kernel/utils.py:12-20
    if not arrays and dataloader is None:
        raise ValueError("Provide either (xi, zi) or dataloader.")
    if parameterized_mean:

        def crit_(param, xi, zi):
            return selection_criterion(model, param[:meanparam_len], param[meanparam_len:], xi, zi)

    else:

        def crit_(covparam, xi, zi):
            return selection_criterion(model, covparam, xi, zi)

    analytic = _analytic_for(model, selection_criterion, parameterized_mean)
    if arrays:
        xi_, zi_ = gnp.asarray(xi), gnp.asarray(zi)
        crit = gnp.DifferentiableSelectionCriterion(crit_, xi_, zi_, analytic=analytic)
    else:   # mini-batches: sized iterable of (x_batch, z_batch)
        crit = gnp.BatchDifferentiableSelectionCriterion(crit_, dataloader, batches_per_eval=batches_per_eval, analytic=analytic)
    return crit.evaluate, crit.evaluate_pre_grad, crit.evaluate_no_grad, crit.gradient


def autoselect_parameters(p0, criterion, gradient, bounds=None, bounds_auto=True, bounds_delta=10.0, silent=True,
                          info=False, method="SLSQP", method_options=None):
    """gpmp/kernel/parameter_selection.py:128-276 (same options, bounds, history and best-seen rule)."""
    if method_options is None:
        method_options = {}
    tic = time.time()
    p0 = np.asarray(gnp.to_np(p0), dtype=np.float64).reshape(-1)
    safe_lower, safe_upper = -500, 500
    if bounds is None and bounds_auto:
        bounds = [(max(p - bounds_delta, safe_lower), min(p + bounds_delta, safe_upper)) for p in p0]

    history_params, history_criterion = [], []
    best = {"p": None, "J": float("inf")}

    def criterion_with_history(p):
        try:
            J = float(criterion(p))
        except Exception as exc:  # linear-algebra failure -> +inf (parameter_selection.py:222-231)
            if gnp._is_linalg_exception(exc):
                J = np.inf
            else:
                raise
        history_params.append(p.copy())
        history_criterion.append(J)
        if J < best["J"]:
            best["J"], best["p"] = J, p.copy()
        return J

    options = {"disp": not silent}
    if method == "L-BFGS-B":
        options.update(dict(maxcor=20, ftol=1e-6, gtol=1e-5, eps=1e-8, maxfun=15000, maxiter=15000, maxls=40))
    elif method == "SLSQP":
        options.update(dict(ftol=1e-6, eps=1e-8, maxiter=15000))
    else:
        raise ValueError("Optimization method not implemented.")
    options.update(method_options)
    if method == "L-BFGS-B":
        options.pop("disp", None)

    r = minimize(criterion_with_history, p0, method=method, jac=gradient, bounds=bounds, options=options)
    if best["p"] is not None and r.fun > best["J"]:
        r.x, r.fun, r.best_value_returned = best["p"], best["J"], False
    else:
        r.best_value_returned = True
    r.history_params, r.history_criterion = history_params, history_criterion
    r.initial_params, r.final_params, r.bounds = p0, r.x, bounds
    r.selection_criterion = criterion
    r.total_time = time.time() - tic
    return (r.x, r) if info else (r.x, None)


def select_parameters_with_criterion(model, criterion, xi=None, zi=None, dataloader=None, meanparam0=None,
                                     covparam0=None, parameterized_mean=False, meanparam_len=1, info=False,
                                     verbosity=0, *, bounds=None, bounds_auto=True, bounds_delta=10.0,
                                     batches_per_eval=0, method="SLSQP", method_options=None):
    """gpmp/kernel/parameter_selection.py:280-437."""
    tic = time.time()
    if covparam0 is None:
        covparam0 = anisotropic_parameters_initial_guess(model, xi, zi, dataloader)
    covparam0 = np.asarray(gnp.to_np(covparam0), dtype=np.float64)
    if parameterized_mean:
        if meanparam0 is None:
            raise ValueError("meanparam0 must be provided when parameterized_mean=True.")
        param0 = np.concatenate([np.asarray(gnp.to_np(meanparam0), dtype=np.float64), covparam0])
    else:
        param0 = covparam0
    crit, crit_pre_grad, crit_no_grad, crit_grad = make_selection_criterion_with_gradient(
        model, criterion, xi, zi, dataloader, batches_per_eval=batches_per_eval,
        parameterized_mean=parameterized_mean, meanparam_len=meanparam_len)
    if verbosity == 1:
        print("Parameter selection using custom criterion...")
    param_opt, info_ret = autoselect_parameters(param0, crit_pre_grad, crit_grad, bounds=bounds, bounds_auto=bounds_auto,
                                                bounds_delta=bounds_delta, silent=not (verbosity == 2), info=True,
                                                method=method, method_options=method_options)
    if verbosity == 1:
        print("done.")
    if parameterized_mean:
        meanparam_opt, covparam_opt = param_opt[:meanparam_len], param_opt[meanparam_len:]
        model.meanparam = np.asarray(meanparam_opt)
    else:
        meanparam_opt, covparam_opt = None, param_opt
    model.covparam = np.asarray(covparam_opt)
    if info:
        info_ret["meanparam0"] = meanparam0 if parameterized_mean else None
        info_ret["covparam0"] = covparam0
        info_ret["meanparam"] = meanparam_opt
        info_ret["covparam"] = covparam_opt
        info_ret["selection_criterion"] = crit
        info_ret["selection_criterion_nograd"] = crit_no_grad
        info_ret["time"] = time.time() - tic
        return model, info_ret
    return model, None


def select_parameters_with_reml(model, xi=None, zi=None, dataloader=None, covparam0=None, info=False, verbosity=0, *,
                                bounds=None, bounds_auto=True, bounds_delta=10.0, method="SLSQP", method_options=None):
    """gpmp/kernel/parameter_selection.py:747-808."""
    return select_parameters_with_criterion(model, negative_log_restricted_likelihood, xi=xi, zi=zi, dataloader=dataloader,
                                            covparam0=covparam0, info=info, verbosity=verbosity, bounds=bounds,
                                            bounds_auto=bounds_auto, bounds_delta=bounds_delta, method=method,
                                            method_options=method_options)


# ------------------------------------------------------------------------------------------------
# REMAP: REML regularised by priors (gpmp/kernel/parameter_selection.py:867-1483).  The priors are O(d)
# scalar terms on the host; value and gradient of the REML part come from the HIP path.
# ------------------------------------------------------------------------------------------------
class _RemapAnalytic:
    """REML analytic value/gradient + (value, gradient) of a host-side negative log-prior."""

    def __init__(self, model, neg_log_prior, grad_neg_log_prior):
        from ..core import gradients as _gradients

        self.reml = _gradients.REMLAnalytic(model)
        self.nlp, self.gnlp = neg_log_prior, grad_neg_log_prior

    def value_and_state(self, covparam, xi, zi):
        prior = float(self.nlp(covparam))
        if not np.isfinite(prior):
            raise np.linalg.LinAlgError("prior support violated (treated like a singular matrix: criterion = +inf)")
        value, state = self.reml.value_and_state(covparam, xi, zi)
        return value + prior, (state, np.array(covparam, dtype=np.float64))

    def gradient_from_state(self, state):
        reml_state, covparam = state
        return self.reml.gradient_from_state(reml_state) + self.gnlp(covparam)

    # (qualification and piece size of the batched route are the REML part's)
    def batch_qualifies(self):
        return self.reml.batch_qualifies()

    @property
    def batch_max_points(self):
        return self.reml.batch_max_points

    def batch_piece_limit(self, *a, **kw):
        return self.reml.batch_piece_limit(*a, **kw)

    def batch_values_and_gradients(self, covparam, batches, want_grad=True):
        """every batch's REML through the batched library call; the prior (O(d), host) is added to each batch as
        the one-at-a-time route does"""
        prior = float(self.nlp(covparam))
        if not np.isfinite(prior):
            raise np.linalg.LinAlgError("prior support violated (treated like a singular matrix: criterion = +inf)")
        out = self.reml.batch_values_and_gradients(covparam, batches, want_grad)
        if out is None:
            return None
        values, grads = out
        return values + prior, (grads + self.gnlp(np.asarray(covparam, dtype=np.float64)) if want_grad else None)

    def many_values_and_gradients(self, P, xi, zi, want_grad=False):
        """REMAP at every row of ``P`` on the same data (sampler pattern): the REMLs in one batched call, each row's prior
        (O(d), host) added; a row outside the prior's support gets +inf and a zero gradient"""
        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        priors = np.array([float(self.nlp(p)) for p in P])
        ok = np.isfinite(priors)
        values = np.full(P.shape[0], np.inf)
        grads = np.zeros_like(P) if want_grad else None
        if ok.any():
            out = self.reml.many_values_and_gradients(P[ok], xi, zi, want_grad)
            if out is None:
                return None
            v, g = out
            values[ok] = v + priors[ok]
            if want_grad:
                grads[ok] = g + np.array([self.gnlp(p) for p in P[ok]])
        return values, grads


def select_parameters_with_remap_gaussian_logsigma2_and_logrho_prior(
        model, xi=None, zi=None, dataloader=None, covparam0=None, info=False, verbosity=0, *, covparam0_prior=None,
        prior_gamma=None, prior_sigma2_coverage=None, prior_rho_min_range_factor=None, prior_logrho_min=None,
        prior_log_sigma2_0=None, prior_logrho_0=None, prior_alpha=None, covparam0_init=None, bounds=None,
        bounds_auto=True, bounds_delta=10.0, method="SLSQP", method_options=None):
    """gpmp/kernel/parameter_selection.py:1301-1483: REML - log p(log sigma^2) - log p(log rho)."""
    covparam0_prior, covparam0_init = resolve_covparam0_prior_and_init(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_prior=covparam0_prior,
        covparam0_init=covparam0_init)
    (prior_gamma, prior_sigma2_coverage, prior_alpha, prior_rho_min_range_factor, prior_log_sigma2_0, prior_logrho_0,
     prior_logrho_min) = resolve_logsigma2_logrho_prior_args(
        covparam0_prior=covparam0_prior, xi=xi, dataloader=dataloader, prior_gamma=prior_gamma,
        prior_sigma2_coverage=prior_sigma2_coverage, prior_alpha=prior_alpha,
        prior_rho_min_range_factor=prior_rho_min_range_factor, prior_log_sigma2_0=prior_log_sigma2_0,
        prior_logrho_0=prior_logrho_0, prior_logrho_min=prior_logrho_min)

    def criterion(m, covparam, x, z):
        return _priors.neg_log_restricted_posterior_logsigma2_and_logrho_prior(
            m, covparam, x, z, log_sigma2_0=prior_log_sigma2_0, gamma=prior_gamma, sigma2_coverage=prior_sigma2_coverage,
            logrho_min=prior_logrho_min, logrho_0=prior_logrho_0, alpha=prior_alpha)

    def neg_log_prior(covparam):
        return -(_priors.log_prior_gaussian_logsigma2(covparam, prior_log_sigma2_0, gamma=prior_gamma,
                                                      sigma2_coverage=prior_sigma2_coverage)
                 + _priors.log_prior_logrho_barrier_linear(covparam, logrho_min=prior_logrho_min, logrho_0=prior_logrho_0,
                                                           alpha=prior_alpha))

    def grad_neg_log_prior(covparam):
        return (_priors.grad_neg_log_prior_gaussian_logsigma2(covparam, prior_log_sigma2_0, gamma=prior_gamma,
                                                              sigma2_coverage=prior_sigma2_coverage)
                + _priors.grad_neg_log_prior_logrho_barrier_linear(covparam, prior_logrho_min, prior_logrho_0,
                                                                   alpha=prior_alpha))

    criterion._gpmp_analytic_factory = lambda m: _RemapAnalytic(m, neg_log_prior, grad_neg_log_prior)
    return select_parameters_with_criterion(model, criterion, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0_init,
                                            info=info, verbosity=verbosity, bounds=bounds, bounds_auto=bounds_auto,
                                            bounds_delta=bounds_delta, method=method, method_options=method_options)


def select_parameters_with_remap_gaussian_logsigma2(
        model, xi=None, zi=None, dataloader=None, covparam0=None, info=False, verbosity=0, *, covparam0_prior=None,
        prior_gamma=None, prior_sigma2_coverage=None, covparam0_init=None, bounds=None, bounds_auto=True, bounds_delta=10.0,
        method="SLSQP", method_options=None):
    """gpmp/kernel/parameter_selection.py:1089-1201: REML - log p(log sigma^2), Gaussian prior centred at covparam0_prior[0]."""
    covparam0_prior, covparam0_init = resolve_covparam0_prior_and_init(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_prior=covparam0_prior,
        covparam0_init=covparam0_init)
    prior_gamma, prior_sigma2_coverage, _, _ = resolve_prior_defaults_for_selection(
        xi=xi, dataloader=dataloader, gamma=prior_gamma, sigma2_coverage=prior_sigma2_coverage)
    log_sigma2_0 = float(covparam0_prior[0])

    def criterion(m, covparam, x, z):
        return _priors.neg_log_restricted_posterior_logsigma2_prior(m, covparam, x, z, log_sigma2_0=log_sigma2_0, gamma=prior_gamma,
                                                                    sigma2_coverage=prior_sigma2_coverage)

    criterion._gpmp_analytic_factory = lambda m: _RemapAnalytic(
        m, lambda c: -_priors.log_prior_gaussian_logsigma2(c, log_sigma2_0, gamma=prior_gamma, sigma2_coverage=prior_sigma2_coverage),
        lambda c: _priors.grad_neg_log_prior_gaussian_logsigma2(c, log_sigma2_0, gamma=prior_gamma, sigma2_coverage=prior_sigma2_coverage))
    return select_parameters_with_criterion(model, criterion, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0_init,
                                            info=info, verbosity=verbosity, bounds=bounds, bounds_auto=bounds_auto,
                                            bounds_delta=bounds_delta, method=method, method_options=method_options)


def select_parameters_with_remap(model, xi=None, zi=None, dataloader=None, covparam0=None, covparam0_init=None, info=False,
                                 verbosity=0, **kwargs):
    """gpmp/kernel/parameter_selection.py:867-965: alias of the Gaussian-log-sigma2 + log-rho procedure."""
    return select_parameters_with_remap_gaussian_logsigma2_and_logrho_prior(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_init=covparam0_init, info=info,
        verbosity=verbosity, **kwargs)


def select_parameters_with_remap_with_power_laws_prior(model, xi=None, zi=None, dataloader=None, covparam0=None, info=False,
                                                       verbosity=0, *, bounds=None, bounds_auto=True, bounds_delta=10.0,
                                                       method="SLSQP", method_options=None):
    """gpmp/kernel/parameter_selection.py:969-1030 (finite-difference jacobian, as with the NumPy backend)."""
    return select_parameters_with_criterion(model, _priors.neg_log_restricted_posterior_power_laws_prior, xi=xi, zi=zi,
                                            dataloader=dataloader, covparam0=covparam0, info=info, verbosity=verbosity,
                                            bounds=bounds, bounds_auto=bounds_auto, bounds_delta=bounds_delta, method=method,
                                            method_options=method_options)


# ------------------------------------------------------------------------------------------------
# update_* procedures (start from the model's current parameters) and ML with a constant mean
# ------------------------------------------------------------------------------------------------
def update_parameters_with_criterion(model, criterion, xi=None, zi=None, dataloader=None, parameterized_mean=False,
                                     meanparam_len=1, info=False, verbosity=0, **kw):
    """gpmp/kernel/parameter_selection.py:440-507: re-optimise starting at model.meanparam / model.covparam."""
    if model.covparam is None:
        raise ValueError("model.covparam must be set before an update; use select_parameters_with_criterion first.")
    return select_parameters_with_criterion(model, criterion, xi=xi, zi=zi, dataloader=dataloader,
                                            meanparam0=model.meanparam if parameterized_mean else None,
                                            covparam0=model.covparam, parameterized_mean=parameterized_mean,
                                            meanparam_len=meanparam_len, info=info, verbosity=verbosity, **kw)


def update_parameters_with_reml(model, xi=None, zi=None, dataloader=None, info=False, verbosity=0, **kw):
    """gpmp/kernel/parameter_selection.py:811-864."""
    return update_parameters_with_criterion(model, negative_log_restricted_likelihood, xi=xi, zi=zi, dataloader=dataloader,
                                            info=info, verbosity=verbosity, **kw)


def update_parameters_with_remap_gaussian_logsigma2_and_logrho_prior(
        model, xi=None, zi=None, dataloader=None, info=False, verbosity=0, *, covparam0=None, covparam0_prior=None,
        covparam0_init=None, **kw):
    """gpmp/kernel/parameter_selection.py:1486-1577: prior anchor / optimiser start resolved by
    resolve_covparam0_roles_for_update (model.covparam when nothing is given, with the reference's warning)."""
    covparam0_prior, covparam0_init = resolve_covparam0_roles_for_update(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_prior=covparam0_prior,
        covparam0_init=covparam0_init)
    return select_parameters_with_remap_gaussian_logsigma2_and_logrho_prior(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_prior=covparam0_prior,
        covparam0_init=covparam0_init, info=info, verbosity=verbosity, **kw)


def update_parameters_with_remap(model, xi=None, zi=None, dataloader=None, covparam0=None, covparam0_prior=None,
                                 covparam0_init=None, info=False, verbosity=0, **kw):
    """gpmp/kernel/parameter_selection.py:922-965: alias of the Gaussian-log-sigma2 + log-rho update."""
    return update_parameters_with_remap_gaussian_logsigma2_and_logrho_prior(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_prior=covparam0_prior,
        covparam0_init=covparam0_init, info=info, verbosity=verbosity, **kw)


def update_parameters_with_remap_gaussian_logsigma2(
        model, xi=None, zi=None, dataloader=None, info=False, verbosity=0, *, covparam0=None, covparam0_prior=None,
        covparam0_init=None, **kw):
    """gpmp/kernel/parameter_selection.py:1204-1297."""
    covparam0_prior, covparam0_init = resolve_covparam0_roles_for_update(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_prior=covparam0_prior,
        covparam0_init=covparam0_init)
    return select_parameters_with_remap_gaussian_logsigma2(
        model, xi=xi, zi=zi, dataloader=dataloader, covparam0=covparam0, covparam0_prior=covparam0_prior,
        covparam0_init=covparam0_init, info=info, verbosity=verbosity, **kw)


def update_parameters_with_remap_with_power_laws_prior(model, xi=None, zi=None, dataloader=None, info=False, verbosity=0, **kw):
    """gpmp/kernel/parameter_selection.py:1033-1086: re-optimise from model.covparam with the power-laws prior."""
    return update_parameters_with_criterion(model, _priors.neg_log_restricted_posterior_power_laws_prior, xi=xi, zi=zi,
                                            dataloader=dataloader, info=info, verbosity=verbosity, **kw)


def select_parameters_with_ml_constant_mean(model, xi=None, zi=None, dataloader=None, meanparam0=None, covparam0=None,
                                            info=False, verbosity=0, *, bounds=None, bounds_auto=True, bounds_delta=10.0,
                                            method="SLSQP", method_options=None):
    """gpmp/kernel/parameter_selection.py:583-680: ML over [constant mean, covparam] (meantype 'parameterized')."""
    if meanparam0 is None or covparam0 is None:
        m0, c0 = anisotropic_parameters_initial_guess_constant_mean(model, xi, zi, dataloader)
        meanparam0 = m0 if meanparam0 is None else meanparam0
        covparam0 = c0 if covparam0 is None else covparam0
    return select_parameters_with_criterion(model, negative_log_likelihood, xi=xi, zi=zi, dataloader=dataloader,
                                            meanparam0=meanparam0, covparam0=covparam0, parameterized_mean=True, meanparam_len=1,
                                            info=info, verbosity=verbosity, bounds=bounds, bounds_auto=bounds_auto,
                                            bounds_delta=bounds_delta, method=method, method_options=method_options)


def update_parameters_with_ml_constant_mean(model, xi=None, zi=None, dataloader=None, info=False, verbosity=0, **kw):
    """gpmp/kernel/parameter_selection.py:683-744."""
    return select_parameters_with_ml_constant_mean(model, xi=xi, zi=zi, dataloader=dataloader, meanparam0=model.meanparam,
                                                   covparam0=model.covparam, info=info, verbosity=verbosity, **kw)
