"""Gaussian-process model facade -- counterpart of gpmp/core/model.py:22-696 for the hot path.

Same constructor, attributes and method signatures as the reference ``Model`` for
``predict`` / ``loo`` / ``negative_log_likelihood*`` / ``negative_log_restricted_likelihood`` /
``norm_k_sqrd*`` / ``k_inverses`` and the kriging predictors; the arithmetic runs in libgpmp_hip.so.
Also carried: Fisher information (analytic covariance derivatives) and sample paths by
the Cholesky route; ``fisher_information_torch`` differentiates the HIP log-det by finite differences.
"""
import warnings

from .. import num as gnp
from . import fisher, kriging, likelihood, linalg, loo, sample_paths, utils


class Model:
    """GP model: mean / covariance callables + parameters (gpmp/core/model.py:136-166).

    ``covariance(x, y, covparam, pairwise=False)`` with ``y is None`` meaning ``y := x`` and
    ``pairwise=True`` returning an (n,) vector; ``mean(x, meanparam) -> (n, q)``
    (gpmp/core/model.py:30-66).  Passing a ``gpmp_amd.kernel.MaternCovariance`` as ``covariance``
    enables the fused lower-triangle Gram build and the analytic ML / REML gradients.
    """

    def __init__(self, mean, covariance, meanparam=None, covparam=None, meantype="linear_predictor"):
        utils.validate_model_mean(meantype, mean, meanparam)
        self.meantype = meantype
        self.mean = mean
        self.meanparam = meanparam
        self.covparam = covparam
        self.covariance = covariance

    def __repr__(self):
        return "<gpmp_amd.core.Model object> " + hex(id(self))

    def __str__(self):
        mean_desc = "Zero Mean" if self.meantype == "zero" else getattr(self.mean, "__name__", str(self.mean))
        cov_desc = getattr(self.covariance, "__name__", str(self.covariance))
        return (
            f"GP Model:\n  Mean Type: {self.meantype}\n  Mean Function: {mean_desc}\n"
            f"  Mean Parameters: {self.meanparam}\n  Covariance Function: {cov_desc}\n"
            f"  Covariance Parameters: {self.covparam}"
        )

    # ------------------------------------------------------------------ kriging predictors
    def kriging_predictor_with_zero_mean(self, xi, xt, return_type=0):
        return kriging.kriging_predictor_with_zero_mean(self, xi, xt, return_type)

    def kriging_predictor(self, xi, xt, return_type=0):
        return kriging.kriging_predictor(self, xi, xt, return_type)

    # ------------------------------------------------------------------ public API
    def predict(self, xi, zi, xt, return_lambdas=False, zero_neg_variances=True, convert_in=True, convert_out=True):
        """Posterior mean and variance at xt given (xi, zi) -- gpmp/core/model.py:227-307."""
        xi, zi, xt = utils.ensure_shapes_and_type(xi=xi, zi=zi, xt=xt, convert=convert_in)
        fused = None if return_lambdas else kriging.fused_prediction(self, xi, zi, xt)
        if fused is not None:
            lambda_t = None
            zt_kriging_mean, zt_posterior_variance, zt_prior_mean = fused
        else:
            zi_centered, zt_prior_mean, lambda_t, zt_posterior_variance, zt_kriging_mean = kriging.select_predictor(
                self, xi, zi, xt, return_lambdas=return_lambdas
            )
        if bool(gnp.any(zt_posterior_variance < 0.0)):
            warnings.warn("Negative variances detected. Consider using jitter.", RuntimeWarning)
        if zero_neg_variances:
            zt_posterior_variance = gnp.maximum(zt_posterior_variance, 0.0)
        zt_posterior_mean = zt_kriging_mean + zt_prior_mean   # = lambda_t^T zi_centered + prior mean
        if convert_out:
            zt_posterior_mean = gnp.to_np(zt_posterior_mean)
            zt_posterior_variance = gnp.to_np(zt_posterior_variance)
        if return_lambdas:
            return (zt_posterior_mean, zt_posterior_variance, lambda_t)
        return (zt_posterior_mean, zt_posterior_variance)

    def loo(self, xi, zi, convert_in=True, convert_out=False):
        """Leave-one-out predictions -- gpmp/core/model.py:309-343."""
        xi_, zi_, _ = utils.ensure_shapes_and_type(xi=xi, zi=zi, convert=convert_in)
        zloo, sigma2loo, eloo = loo.loo(self, xi_, zi_)
        if convert_out:
            zloo, sigma2loo, eloo = gnp.to_np(zloo), gnp.to_np(sigma2loo), gnp.to_np(eloo)
        return zloo, sigma2loo, eloo

    # ------------------------------------------------------------------ likelihoods and norms
    def negative_log_likelihood_zero_mean(self, covparam, xi, zi):
        return likelihood.negative_log_likelihood_zero_mean(self, covparam, xi, zi)

    def negative_log_likelihood(self, meanparam, covparam, xi, zi):
        return likelihood.negative_log_likelihood(self, meanparam, covparam, xi, zi)

    def negative_log_restricted_likelihood(self, covparam, xi, zi):
        return likelihood.negative_log_restricted_likelihood(self, covparam, xi, zi)

    def norm_k_sqrd_with_zero_mean(self, xi, zi, covparam):
        xi, zi, _ = utils.ensure_shapes_and_type(xi=xi, zi=zi)
        return linalg.norm_k_sqrd_with_zero_mean(self, xi, zi, covparam)

    def k_inverses(self, xi, zi, covparam):
        xi = gnp.asarray(xi)
        zi = gnp.asarray(zi)
        return linalg.k_inverses(self, xi, zi, covparam)

    def norm_k_sqrd(self, xi, zi, covparam):
        xi, zi, _ = utils.ensure_shapes_and_type(xi=xi, zi=zi)
        return linalg.norm_k_sqrd(self, xi, zi, covparam)

    # ------------------------------------------------------------------ Fisher information (gpmp/core/model.py:509-571)
    def fisher_information(self, xi, covparam=None, epsilon=1e-3):
        return fisher.fisher_information(self, xi, covparam=covparam, epsilon=epsilon)

    def fisher_information_cpd(self, xi, covparam=None, epsilon=1e-3):
        return fisher.fisher_information_cpd(self, xi, covparam=covparam, epsilon=epsilon)

    def fisher_information_torch(self, xi, covparam):
        """gpmp/core/model.py:569-571 (0.5 * Hessian of log|K|; finite differences on this backend)."""
        return fisher.fisher_information_torch(self, xi, covparam)

    # ------------------------------------------------------------------ sample paths (gpmp/core/model.py:576-696)
    def sample_paths(self, xt, nb_paths, method="chol", check_result=True):
        return sample_paths.sample_paths(self, xt, nb_paths, method=method, check_result=check_result)

    def conditional_sample_paths(self, ztsim, xi_ind, zi, xt_ind, lambda_t, convert_out=True):
        return sample_paths.conditional_sample_paths(self, ztsim, xi_ind, zi, xt_ind, lambda_t, convert_out=convert_out)

    def conditional_sample_paths_parameterized_mean(self, ztsim, xi, xi_ind, zi, xt, xt_ind, lambda_t, convert_out=True):
        return sample_paths.conditional_sample_paths_parameterized_mean(self, ztsim, xi, xi_ind, zi, xt, xt_ind, lambda_t,
                                                                        convert_out=convert_out)
