"""gpmp_amd.kernel -- Matern covariances and the REML selection driver (gpmp/kernel counterpart)."""
from .matern import (
    MaternCovariance,
    matern32_kernel,
    maternp_covariance,
    maternp_covariance_ii_or_tt,
    maternp_covariance_it,
    maternp_kernel,
)
from .init import anisotropic_parameters_initial_guess, anisotropic_parameters_initial_guess_zero_mean
from .parameter_selection import (
    autoselect_parameters,
    make_selection_criterion_with_gradient,
    negative_log_likelihood,
    negative_log_likelihood_zero_mean,
    negative_log_restricted_likelihood,
    select_parameters_with_criterion,
    select_parameters_with_reml,
)

__all__ = [
    "MaternCovariance", "matern32_kernel", "maternp_kernel", "maternp_covariance",
    "maternp_covariance_ii_or_tt", "maternp_covariance_it",
    "anisotropic_parameters_initial_guess", "anisotropic_parameters_initial_guess_zero_mean",
    "negative_log_likelihood_zero_mean", "negative_log_likelihood", "negative_log_restricted_likelihood",
    "make_selection_criterion_with_gradient", "autoselect_parameters",
    "select_parameters_with_criterion", "select_parameters_with_reml",
]
