"""Host-side pieces of the REMAP criteria (priors, data-driven bounds) against the reference's values.
Pure host arithmetic: runs without a GPU."""
import numpy as np

from gpmp_amd.kernel import prior_helpers, priors


def test_logrho_min_and_prior_values(golden):
    g = golden("remap")
    for tag in ("a", "b"):
        xi = g[f"remap_{tag}_xi"]
        np.testing.assert_allclose(prior_helpers.compute_logrho_min_from_xi(xi), g[f"remap_{tag}_logrho_min"], rtol=1e-14)
        gamma, cov, alpha, rfac, ls20 = g[f"remap_{tag}_prior_scalars"]
        assert (gamma, cov, alpha, rfac) == (1.5, 0.95, 1.0, 1 / 20.0)
        args = prior_helpers.resolve_logsigma2_logrho_prior_args(covparam0_prior=g[f"remap_{tag}_covparam0"], xi=xi)
        assert abs(args[4] - ls20) < 1e-15
        np.testing.assert_allclose(args[5], g[f"remap_{tag}_logrho_0"], rtol=1e-15)
        np.testing.assert_allclose(args[6], g[f"remap_{tag}_logrho_min_resolved"], rtol=1e-14)
        lr0, lrmin = g[f"remap_{tag}_logrho_0"], g[f"remap_{tag}_logrho_min_resolved"]
        for i, t in enumerate(g[f"remap_{tag}_thetas"]):
            assert abs(priors.log_prior_gaussian_logsigma2(t, ls20) - g[f"remap_{tag}_lp_sigma2"][i]) < 1e-13
            assert abs(priors.log_prior_logrho_barrier_linear(t, lrmin, lr0) - g[f"remap_{tag}_lp_logrho"][i]) < 1e-12
            assert abs(priors.log_prior_power_law(t) - g[f"remap_{tag}_lp_power"][i]) < 1e-12
        assert np.isinf(float(g[f"remap_{tag}_crit_barrier"]))
        assert np.isneginf(priors.log_prior_logrho_barrier_linear(g[f"remap_{tag}_theta_barrier"], lrmin, lr0))


def test_prior_gradients_by_finite_differences(golden):
    g = golden("remap")
    lr0, lrmin = g["remap_b_logrho_0"], g["remap_b_logrho_min_resolved"]
    ls20 = float(g["remap_b_prior_scalars"][4])
    t = g["remap_b_thetas"][0]

    def f(th):
        return -(priors.log_prior_gaussian_logsigma2(th, ls20) + priors.log_prior_logrho_barrier_linear(th, lrmin, lr0))

    ana = priors.grad_neg_log_prior_gaussian_logsigma2(t, ls20) + priors.grad_neg_log_prior_logrho_barrier_linear(t, lrmin, lr0)
    num = np.array([(f(t + 1e-6 * e) - f(t - 1e-6 * e)) / 2e-6 for e in np.eye(len(t))])
    np.testing.assert_allclose(ana, num, rtol=1e-6, atol=1e-8)
