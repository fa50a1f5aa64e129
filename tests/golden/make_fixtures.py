#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (gpmp v0.9.37).

Run in the build container only (the reference does not travel to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=numpy python3 /root/repo/tests/golden/make_fixtures.py numpy
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=torch python3 /root/repo/tests/golden/make_fixtures.py torch
    (single files: ... make_fixtures.py torch batch | torch gradients_p0 | numpy cpd | numpy remap_extra | numpy dataloader | numpy namespace)

The "numpy" pass pins values (Matern, covariance, predict, NLL, REML, LOO, init guess,
example02 flow) with the reference's NumPy backend -- the parity target named by
BASELINE.json.  The "torch" pass pins the ML / REML gradients with the reference's
torch-CPU autograd route (the NumPy backend has no gradient: numpy_backend.py:333).

Only inputs and outputs (plain arrays) are stored; no reference source is copied.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
backend = sys.argv[1] if len(sys.argv) > 1 else "numpy"
os.environ["GPMP_BACKEND"] = backend
os.environ.setdefault("GPMP_LOG_LEVEL", "WARNING")

import gpmp as gp  # noqa: E402  (the reference)
import gpmp.num as gnp  # noqa: E402

assert gnp._gpmp_backend_ == backend, (gnp._gpmp_backend_, backend)


# ---------------------------------------------------------------- helpers
def make_xz(n, d, seed, noise=0.0):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    z = np.sin(2 * np.pi * x[:, 0]) + x[:, 1:].sum(axis=1)
    if noise:
        z = z + noise * rng.standard_normal(n)
    return x, z


def theta_aniso(d, sigma2=1.0, scale=1.0):
    rho = scale * 0.5 * (1.0 + np.arange(d) / d)
    return np.concatenate(([np.log(sigma2)], -np.log(rho)))


def constant_mean(x, param):
    return gnp.ones((x.shape[0], 1))


def linear_mean(x, param):
    return gnp.hstack((gnp.ones((x.shape[0], 1)), gnp.asarray(x)))


def param_mean(x, param):
    return (param[0] + param[1] * x[:, 0]).reshape(-1, 1)


def make_kernel(p):
    def kernel(x, y, covparam, pairwise=False):
        return gp.kernel.maternp_covariance(x, y, p, covparam, pairwise)

    return kernel


def make_noisy_kernel(p):
    # same construction as examples/gpmp_example07_nd_regression.py:95-131
    def kernel(x, y, param, pairwise=False):
        sigma2 = gnp.exp(param[0])
        noise_variance = gnp.exp(param[1])
        loginvrho = param[2:]
        if y is x or y is None:
            if pairwise:
                return sigma2 * gnp.ones((x.shape[0],))
            K = gnp.scaled_distance(loginvrho, x, x)
            return sigma2 * gp.kernel.maternp_kernel(p, K) + noise_variance * gnp.eye(K.shape[0])
        if pairwise:
            K = gnp.scaled_distance_elementwise(loginvrho, x, y)
        else:
            K = gnp.scaled_distance(loginvrho, x, y)
        return sigma2 * gp.kernel.maternp_kernel(p, K)

    return kernel


def tonp(a):
    return np.asarray(gnp.to_np(a) if backend == "numpy" else a.detach().cpu().numpy(), dtype=np.float64)


# ---------------------------------------------------------------- numpy pass
def gen_matern(out):
    h = np.array([0.0, 1e-300, 1e-12, 1e-6, 0.01, 0.1, 0.5, 1.0, 2.0, 5.0, 10.0, 40.0, 100.0, 400.0, np.inf])
    out["matern_h"] = h
    for p in (0, 1, 2, 3, 6, 10):
        with np.errstate(over="ignore", invalid="ignore"):
            out[f"matern_k_p{p}"] = gp.kernel.maternp_kernel(p, h.copy())
    for tag, (n, m, d), p in (("a", (37, 53, 3), 2), ("b", (160, 64, 8), 2), ("c", (33, 17, 2), 3), ("d", (40, 40, 5), 0)):
        x, _ = make_xz(n, d, 11)
        y, _ = make_xz(m, d, 12)
        th = theta_aniso(d, sigma2=1.7)
        out[f"cov_{tag}_x"], out[f"cov_{tag}_y"], out[f"cov_{tag}_theta"], out[f"cov_{tag}_p"] = x, y, th, np.array(p)
        out[f"cov_{tag}_ii"] = gp.kernel.maternp_covariance(x, x, p, th)            # identity dispatch y is x
        out[f"cov_{tag}_ii_none_equal"] = np.array(np.array_equal(out[f"cov_{tag}_ii"], gp.kernel.maternp_covariance(x, None, p, th)))
        out[f"cov_{tag}_it"] = gp.kernel.maternp_covariance(x, y, p, th)
        out[f"cov_{tag}_ii_pw"] = gp.kernel.maternp_covariance(x, None, p, th, True)
        ym = y[: min(n, m)]
        xm = x[: min(n, m)]
        out[f"cov_{tag}_it_pw"] = gp.kernel.maternp_covariance(xm, ym, p, th, True)
        # equal-but-not-identical copy goes down the "it" path (no nugget)
        out[f"cov_{tag}_copy"] = gp.kernel.maternp_covariance(x, x.copy(), p, th)
        out[f"dist_{tag}"] = gnp.scaled_distance(th[1:], x, y)


def gen_predict(out):
    cases = []
    for tag, (n, m, d), p in (("s", (64, 10, 3), 2), ("m", (200, 120, 8), 2), ("p3", (50, 20, 2), 3)):
        xi, zi = make_xz(n, d, 21)
        xt, _ = make_xz(m, d, 22)
        th = theta_aniso(d, sigma2=0.8)
        out[f"pred_{tag}_xi"], out[f"pred_{tag}_zi"], out[f"pred_{tag}_xt"] = xi, zi, xt
        out[f"pred_{tag}_theta"], out[f"pred_{tag}_p"] = th, np.array(p)
        k = make_kernel(p)
        mp = np.array([0.3, -0.7])
        out[f"pred_{tag}_meanparam"] = mp
        models = {
            "zero": gp.core.Model(None, k, None, th, "zero"),
            "const": gp.core.Model(constant_mean, k, None, th, "linear_predictor"),
            "lin": gp.core.Model(linear_mean, k, None, th, "linear_predictor"),
            "param": gp.core.Model(param_mean, k, mp, th, "parameterized"),
        }
        for mt, model in models.items():
            zpm, zpv, lam = model.predict(xi, zi, xt, return_lambdas=True)
            out[f"pred_{tag}_{mt}_zpm"], out[f"pred_{tag}_{mt}_zpv"], out[f"pred_{tag}_{mt}_lambda"] = zpm, zpv, lam
            zl, s2, el = model.loo(xi, zi)
            out[f"loo_{tag}_{mt}_zloo"], out[f"loo_{tag}_{mt}_s2"], out[f"loo_{tag}_{mt}_eloo"] = zl, s2, el
        lam, cov = models["zero"].kriging_predictor_with_zero_mean(xi, xt, return_type=1)
        out[f"pred_{tag}_zero_fullcov"] = cov
        # zi given as a column
        zpm, zpv = models["zero"].predict(xi, zi.reshape(-1, 1), xt)
        out[f"pred_{tag}_zero_zpm_col"] = zpm
        cases.append(tag)
    # near-singular: duplicated observation points -> clamp + warning path (model.py:290-296)
    xi, zi = make_xz(40, 2, 23)
    xi = np.vstack((xi, xi[:5] + 1e-9))
    zi = np.concatenate((zi, zi[:5]))
    xt = np.vstack((xi[:7], make_xz(20, 2, 24)[0]))  # predicting AT observation points: var ~ 0 (+-)
    th = theta_aniso(2, sigma2=1.0, scale=4.0)
    model = gp.core.Model(None, make_kernel(2), None, th, "zero")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        zpm, zpv = model.predict(xi, zi, xt)
        zpm2, zpv_raw = model.predict(xi, zi, xt, zero_neg_variances=False)
    out["pred_dup_xi"], out["pred_dup_zi"], out["pred_dup_xt"], out["pred_dup_theta"] = xi, zi, xt, th
    out["pred_dup_zpm"], out["pred_dup_zpv"], out["pred_dup_zpv_raw"] = zpm, zpv, zpv_raw
    out["pred_dup_warned"] = np.array(any(issubclass(x.category, RuntimeWarning) for x in w))
    out["pred_cases"] = np.array(cases)


def gen_likelihood(out):
    rng = np.random.default_rng(31)
    for tag, (n, d), p in (("a", (64, 3), 2), ("b", (256, 8), 2), ("c", (512, 20), 2), ("d", (100, 4), 1)):
        xi, zi = make_xz(n, d, 32)
        out[f"lik_{tag}_xi"], out[f"lik_{tag}_zi"], out[f"lik_{tag}_p"] = xi, zi, np.array(p)
        k = make_kernel(p)
        mz = gp.core.Model(None, k, None, None, "zero")
        mc = gp.core.Model(constant_mean, k, None, None, "linear_predictor")
        ml = gp.core.Model(linear_mean, k, None, None, "linear_predictor")
        mpm = gp.core.Model(param_mean, k, np.array([0.2, 0.5]), None, "parameterized")
        thetas = np.stack([theta_aniso(d) + 0.3 * rng.standard_normal(d + 1) for _ in range(5)])
        out[f"lik_{tag}_thetas"] = thetas
        out[f"lik_{tag}_nll"] = np.array([mz.negative_log_likelihood_zero_mean(t, xi, zi) for t in thetas])
        out[f"lik_{tag}_nll_param"] = np.array([mpm.negative_log_likelihood(np.array([0.2, 0.5]), t, xi, zi) for t in thetas])
        out[f"lik_{tag}_reml_const"] = np.array([mc.negative_log_restricted_likelihood(t, xi, zi) for t in thetas])
        out[f"lik_{tag}_reml_lin"] = np.array([ml.negative_log_restricted_likelihood(t, xi, zi) for t in thetas])
        out[f"lik_{tag}_normk0"] = np.array([mz.norm_k_sqrd_with_zero_mean(xi, zi, t) for t in thetas])
        out[f"lik_{tag}_normk_const"] = np.array([mc.norm_k_sqrd(xi, zi, t) for t in thetas])
        a, b, c = mz.k_inverses(xi, zi, thetas[0])
        out[f"lik_{tag}_kinv_ztKz"], out[f"lik_{tag}_kinv_1"], out[f"lik_{tag}_kinv_z"] = np.array(a), b, c
        out[f"lik_{tag}_init_const"] = gp.kernel.anisotropic_parameters_initial_guess(mc, xi, zi)
        out[f"lik_{tag}_init_zero"] = gp.kernel.anisotropic_parameters_initial_guess_zero_mean(mz, xi, zi)
    # non positive-definite: enormous length-scales -> numerically singular K.
    xi, zi = make_xz(300, 2, 33)
    th_bad = np.array([0.0, -12.0, -12.0])
    mz = gp.core.Model(None, make_kernel(2), None, None, "zero")
    crit = gp.kernel.make_selection_criterion_with_gradient(mz, gp.kernel.negative_log_likelihood_zero_mean, xi, zi)
    try:
        mz.negative_log_likelihood_zero_mean(th_bad, xi, zi)
        raised = ""
    except Exception as exc:  # reference raises numpy.linalg.LinAlgError here
        raised = type(exc).__name__ + ": " + str(exc)
    out["lik_bad_xi"], out["lik_bad_zi"], out["lik_bad_theta"] = xi, zi, th_bad
    out["lik_bad_raised"] = np.array(raised)
    out["lik_bad_pre_grad_value"] = np.array(crit[1](th_bad))  # evaluate_pre_grad -> inf


def gen_example02(out):
    # config 1: examples/gpmp_example02_1d_interpolation.py (ldrandunif is unseeded -> capture xi)
    xt = gp.misc.designs.regulargrid(1, 200, [[-1], [1]])
    zt = gp.misc.testfunctions.twobumps(xt)
    np.random.seed(20261004)
    xi = gp.misc.designs.ldrandunif(1, 6, [[-1], [1]])
    zi = gp.misc.testfunctions.twobumps(xi)
    kernel = make_kernel(3)
    model = gp.core.Model(constant_mean, kernel)
    covparam0 = gp.kernel.anisotropic_parameters_initial_guess(model, xi, zi)
    reml0 = model.negative_log_restricted_likelihood(covparam0, xi, zi)
    model, info = gp.kernel.select_parameters_with_reml(model, xi, zi, info=True)
    zpm, zpv = model.predict(xi, zi, xt)
    out["ex02_xt"], out["ex02_zt"], out["ex02_xi"], out["ex02_zi"] = xt, zt, xi, zi
    out["ex02_covparam0"], out["ex02_reml0"] = covparam0, np.array(reml0)
    out["ex02_covparam"] = np.asarray(model.covparam)
    out["ex02_reml_opt"] = np.array(model.negative_log_restricted_likelihood(model.covparam, xi, zi))
    out["ex02_zpm"], out["ex02_zpv"] = zpm, zpv
    out["ex02_nevals"] = np.array(len(info["history_criterion"]))


def gen_remap(out):
    """REMAP (REML + priors): priors, data-driven bounds, criterion values, and one full selection run."""
    from gpmp.kernel import priors as rp
    from gpmp.kernel.prior_helpers import resolve_logsigma2_logrho_prior_args

    rng = np.random.default_rng(51)
    for tag, (n, d), p in (("a", (60, 2), 2), ("b", (150, 5), 2)):
        xi, zi = make_xz(n, d, 52)
        out[f"remap_{tag}_xi"], out[f"remap_{tag}_zi"], out[f"remap_{tag}_p"] = xi, zi, np.array(p)
        model = gp.core.Model(constant_mean, make_kernel(p))
        out[f"remap_{tag}_logrho_min"] = gp.kernel.compute_logrho_min_from_xi(xi)
        c0 = gp.kernel.anisotropic_parameters_initial_guess(model, xi, zi)
        out[f"remap_{tag}_covparam0"] = c0
        args = resolve_logsigma2_logrho_prior_args(covparam0_prior=c0, xi=xi)
        gamma, cov, alpha, rfac, ls20, lr0, lrmin = args
        out[f"remap_{tag}_prior_scalars"] = np.array([gamma, cov, alpha, rfac, float(ls20)])
        out[f"remap_{tag}_logrho_0"], out[f"remap_{tag}_logrho_min_resolved"] = np.asarray(lr0), np.asarray(lrmin)
        thetas = np.stack([c0 + 0.3 * rng.standard_normal(d + 1) for _ in range(4)])
        out[f"remap_{tag}_thetas"] = thetas
        out[f"remap_{tag}_lp_sigma2"] = np.array([rp.log_prior_gaussian_logsigma2(t, ls20) for t in thetas])
        out[f"remap_{tag}_lp_logrho"] = np.array([rp.log_prior_logrho_barrier_linear(t, lrmin, lr0) for t in thetas])
        out[f"remap_{tag}_lp_power"] = np.array([rp.log_prior_power_law(t) for t in thetas])
        out[f"remap_{tag}_crit"] = np.array([
            rp.neg_log_restricted_posterior_logsigma2_and_logrho_prior(model, t, xi, zi, log_sigma2_0=ls20, logrho_min=lrmin, logrho_0=lr0)
            for t in thetas])
        # barrier: a length-scale below its lower bound -> +inf
        tb = c0.copy()
        tb[1] = -(lrmin[0] - 0.5)
        out[f"remap_{tag}_theta_barrier"] = tb
        out[f"remap_{tag}_crit_barrier"] = np.array(
            rp.neg_log_restricted_posterior_logsigma2_and_logrho_prior(model, tb, xi, zi, log_sigma2_0=ls20, logrho_min=lrmin, logrho_0=lr0))
        model, info = gp.kernel.select_parameters_with_remap(model, xi, zi, info=True)
        out[f"remap_{tag}_covparam_opt"] = np.asarray(model.covparam)
        out[f"remap_{tag}_crit_opt"] = np.array(
            rp.neg_log_restricted_posterior_logsigma2_and_logrho_prior(model, model.covparam, xi, zi, log_sigma2_0=ls20, logrho_min=lrmin, logrho_0=lr0))


def gen_fisher_paths(out):
    """Fisher information (finite-difference covariance derivatives in the reference) and conditioning of sample paths."""
    for tag, (n, d), p in (("a", (40, 2), 2), ("b", (70, 3), 1)):
        xi, zi = make_xz(n, d, 61)
        th = theta_aniso(d, sigma2=0.7)
        out[f"fish_{tag}_xi"], out[f"fish_{tag}_theta"], out[f"fish_{tag}_p"] = xi, th, np.array(p)
        mz = gp.core.Model(None, make_kernel(p), None, th, "zero")
        mc = gp.core.Model(constant_mean, make_kernel(p), None, th, "linear_predictor")
        out[f"fish_{tag}_I"] = mz.fisher_information(xi)
        out[f"fish_{tag}_I_cpd"] = mc.fisher_information_cpd(xi)
    # conditional sample paths: deterministic given (ztsim, lambda_t)
    rng = np.random.default_rng(62)
    xi, zi = make_xz(30, 2, 63)
    xt, _ = make_xz(50, 2, 64)
    th = theta_aniso(2)
    model = gp.core.Model(constant_mean, make_kernel(2), None, th)
    _, _, lam = model.predict(xi, zi, xt, return_lambdas=True)
    xtsim = np.vstack((xi, xt))
    ztsim = rng.standard_normal((80, 5))
    xi_ind, xt_ind = np.arange(30), np.arange(30, 80)
    out["paths_xi"], out["paths_zi"], out["paths_xt"], out["paths_theta"] = xi, zi, xt, th
    out["paths_ztsim"], out["paths_lambda"] = ztsim, lam
    out["paths_cond"] = model.conditional_sample_paths(ztsim, xi_ind, zi, xt_ind, lam)
    mp = np.array([0.3, -0.7])
    mpm = gp.core.Model(param_mean, make_kernel(2), mp, th, "parameterized")
    _, _, lam2 = mpm.predict(xi, zi, xt, return_lambdas=True)
    out["paths_lambda_param"] = lam2
    out["paths_cond_param"] = mpm.conditional_sample_paths_parameterized_mean(ztsim, xi, xi_ind, zi, xt, xt_ind, lam2)


def gen_remap_extra(out):
    """procedures added after ref_remap.npz: Gaussian-log-sigma2-only REMAP (select + update), power-laws update,
    reference prior, empirical bounds"""
    from gpmp.kernel import priors as rp
    from gpmp.kernel.bounds import empirical_bounds_factory

    xi, zi = make_xz(90, 2, 91)
    p = 2
    out["rx_xi"], out["rx_zi"], out["rx_p"] = xi, zi, np.array(p)
    model = gp.core.Model(constant_mean, make_kernel(p))
    c0 = theta_aniso(2) + np.array([0.3, -0.2, 0.1])
    out["rx_covparam0"] = c0
    model, info = gp.kernel.select_parameters_with_remap_gaussian_logsigma2(model, xi, zi, covparam0=c0, info=True)
    out["rx_sel_covparam"], out["rx_sel_crit"] = tonp(model.covparam), np.array(float(info.selection_criterion(model.covparam)))
    out["rx_crit_at_c0"] = np.array(float(info.selection_criterion(gnp.asarray(c0))))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model2 = gp.core.Model(constant_mean, make_kernel(p), covparam=gnp.asarray(c0))
        model2, info2 = gp.kernel.update_parameters_with_remap_gaussian_logsigma2(model2, xi, zi, info=True)
        out["rx_upd_covparam"] = tonp(model2.covparam)
        model3 = gp.core.Model(constant_mean, make_kernel(p), covparam=gnp.asarray(c0))
        model3, info3 = gp.kernel.update_parameters_with_remap_with_power_laws_prior(model3, xi, zi, info=True)
        out["rx_upd_pl_covparam"], out["rx_upd_pl_crit"] = tonp(model3.covparam), np.array(float(info3.selection_criterion(model3.covparam)))
    mz = gp.core.Model(None, make_kernel(p), None, gnp.asarray(c0), "zero")
    out["rx_log_prior_reference"] = np.array(float(rp.log_prior_reference(mz, gnp.asarray(c0), gnp.asarray(xi))))
    out["rx_bounds"] = tonp(empirical_bounds_factory(xi, zi, mean_paramlength=1))
    W = gp.core.linalg.compute_contrast_matrix(gnp.asarray(linear_mean(xi, None)))
    out["rx_contrast_proj"] = tonp(gnp.matmul(W, W.T))          # the projector is unique, W is not



def gen_cpd(out):
    """Universal kriging with a covariance that is only CONDITIONALLY positive definite: k(x, y) = -||invrho (x - y)||
    (the linear variogram, CPD with respect to constants).  K itself is indefinite (zero diagonal), so a Cholesky of K
    does not exist; the reference's block system [[K, P], [P^T, 0]] is solved by LAPACK sysv (kriging.py:98-109) and the
    contrast-space route (kriging.py:202-257) answers as well.  Both are pinned."""
    from gpmp.core import kriging as rk

    def variogram(x, y, covparam, pairwise=False):
        if y is x or y is None:
            if pairwise:
                return gnp.zeros((x.shape[0],))
            return -gnp.exp(covparam[0]) * gnp.scaled_distance(covparam[1:], x, x)
        if pairwise:
            return -gnp.exp(covparam[0]) * gnp.scaled_distance_elementwise(covparam[1:], x, y)
        return -gnp.exp(covparam[0]) * gnp.scaled_distance(covparam[1:], x, y)

    for tag, (n, m, d), mean in (("c", (60, 25, 2), constant_mean), ("l", (150, 40, 3), constant_mean)):
        xi, zi = make_xz(n, d, 31)
        xt, _ = make_xz(m, d, 32)
        th = theta_aniso(d, sigma2=1.3)
        model = gp.core.Model(mean, variogram, None, th, "linear_predictor")
        zpm, zpv, lam = model.predict(xi, zi, xt, return_lambdas=True)
        lam_ns, var_ns = rk._kriging_predictor_nullspace(model, xi, xt, 0)
        _, cov_ns = rk._kriging_predictor_nullspace(model, xi, xt, 1)
        out[f"cpd_{tag}_xi"], out[f"cpd_{tag}_zi"], out[f"cpd_{tag}_xt"], out[f"cpd_{tag}_theta"] = xi, zi, xt, th
        out[f"cpd_{tag}_zpm"], out[f"cpd_{tag}_zpv"], out[f"cpd_{tag}_lambda"] = zpm, zpv, lam
        out[f"cpd_{tag}_ns_lambda"], out[f"cpd_{tag}_ns_var"], out[f"cpd_{tag}_ns_cov"] = tonp(lam_ns), tonp(var_ns), tonp(cov_ns)
    # a positive definite case through the contrast-space route with a q = d + 1 linear mean: must agree with the block solve
    xi, zi = make_xz(90, 3, 33)
    xt, _ = make_xz(30, 3, 34)
    th = theta_aniso(3, sigma2=0.7)
    model = gp.core.Model(linear_mean, make_kernel(2), None, th, "linear_predictor")
    lam_ns, var_ns = rk._kriging_predictor_nullspace(model, xi, xt, 0)
    out["cpd_pd_xi"], out["cpd_pd_zi"], out["cpd_pd_xt"], out["cpd_pd_theta"] = xi, zi, xt, th
    out["cpd_pd_ns_lambda"], out["cpd_pd_ns_var"] = tonp(lam_ns), tonp(var_ns)


def numpy_pass():
    for name, fn in (("matern", gen_matern), ("predict", gen_predict), ("likelihood", gen_likelihood), ("example02", gen_example02), ("remap", gen_remap), ("fisher_paths", gen_fisher_paths)):
        out = {}
        fn(out)
        path = os.path.join(HERE, f"ref_{name}.npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
        print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


# ---------------------------------------------------------------- torch pass (gradients)
def torch_pass():
    import torch

    out = {}
    rng = np.random.default_rng(41)
    cases = (("a", (64, 3), 2), ("b", (256, 8), 2), ("c", (512, 20), 2), ("d", (100, 4), 1), ("e", (80, 2), 3))
    for tag, (n, d), p in cases:
        xi, zi = make_xz(n, d, 42)
        out[f"grad_{tag}_xi"], out[f"grad_{tag}_zi"], out[f"grad_{tag}_p"] = xi, zi, np.array(p)
        k = make_kernel(p)
        mz = gp.core.Model(None, k, None, None, "zero")
        mc = gp.core.Model(constant_mean, k, None, None, "linear_predictor")
        ml = gp.core.Model(linear_mean, k, None, None, "linear_predictor")
        thetas = np.stack([theta_aniso(d) + 0.2 * rng.standard_normal(d + 1) for _ in range(3)])
        out[f"grad_{tag}_thetas"] = thetas
        for name, model, crit_fn in (
            ("nll", mz, gp.kernel.negative_log_likelihood_zero_mean),
            ("reml_const", mc, gp.kernel.negative_log_restricted_likelihood),
            ("reml_lin", ml, gp.kernel.negative_log_restricted_likelihood),
        ):
            _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit_fn, xi, zi)
            vals, grads = [], []
            for t in thetas:
                tt = torch.as_tensor(t, dtype=torch.float64)
                vals.append(float(pre(tt)))
                grads.append(tonp(grad(tt)))
            out[f"grad_{tag}_{name}_val"], out[f"grad_{tag}_{name}_grad"] = np.array(vals), np.stack(grads)
    # noisy kernel theta = [log s2, log s2_noise, log 1/rho ...] (example07 construction)
    for tag, (n, d), p in (("na", (120, 3), 2), ("nb", (300, 6), 2)):
        xi, zi = make_xz(n, d, 43, noise=0.05)
        out[f"grad_{tag}_xi"], out[f"grad_{tag}_zi"], out[f"grad_{tag}_p"] = xi, zi, np.array(p)
        k = make_noisy_kernel(p)
        mz = gp.core.Model(None, k, None, None, "zero")
        mc = gp.core.Model(constant_mean, k, None, None, "linear_predictor")
        base = np.concatenate(([0.0, np.log(0.05 ** 2)], theta_aniso(d)[1:]))
        thetas = np.stack([base + 0.2 * rng.standard_normal(d + 2) for _ in range(3)])
        out[f"grad_{tag}_thetas"] = thetas
        for name, model, crit_fn in (
            ("nll", mz, gp.kernel.negative_log_likelihood_zero_mean),
            ("reml_const", mc, gp.kernel.negative_log_restricted_likelihood),
        ):
            _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit_fn, xi, zi)
            vals, grads = [], []
            for t in thetas:
                tt = torch.as_tensor(t, dtype=torch.float64)
                vals.append(float(pre(tt)))
                grads.append(tonp(grad(tt)))
            out[f"grad_{tag}_{name}_val"], out[f"grad_{tag}_{name}_grad"] = np.array(vals), np.stack(grads)
    # REMAP criterion gradient (autograd through REML + priors) at the thetas of ref_remap.npz
    from gpmp.kernel import priors as rp

    g = np.load(os.path.join(HERE, "ref_remap.npz"))
    for tag in ("a", "b"):
        xi, zi, p = g[f"remap_{tag}_xi"], g[f"remap_{tag}_zi"], int(g[f"remap_{tag}_p"])
        model = gp.core.Model(constant_mean, make_kernel(p))
        ls20 = float(g[f"remap_{tag}_prior_scalars"][4])
        lrmin = torch.as_tensor(g[f"remap_{tag}_logrho_min_resolved"])
        lr0 = torch.as_tensor(g[f"remap_{tag}_logrho_0"])

        def crit(m, covparam, x, z):
            return rp.neg_log_restricted_posterior_logsigma2_and_logrho_prior(m, covparam, x, z, log_sigma2_0=ls20, logrho_min=lrmin, logrho_0=lr0)

        _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit, xi, zi)
        vals, grads = [], []
        for t in g[f"remap_{tag}_thetas"]:
            tt = torch.as_tensor(t, dtype=torch.float64)
            vals.append(float(pre(tt)))
            grads.append(tonp(grad(tt)))
        out[f"grad_remap_{tag}_val"], out[f"grad_remap_{tag}_grad"] = np.array(vals), np.stack(grads)
    path = os.path.join(HERE, "ref_gradients.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
    gen_batch_criterion()


def gen_batch_criterion():
    """BatchDifferentiableSelectionCriterion (gpmp/num/torch_backend.py:607-718) on a list-of-batches loader:
    full-epoch evaluation and the cycling batches_per_eval mode, ML and REML, value + autograd gradient."""
    import torch

    out = {}
    rng = np.random.default_rng(77)
    n, d, p = 330, 3, 2
    xi, zi = make_xz(n, d, 78)
    sizes = [100, 100, 80, 50]                      # ragged last batches
    bounds = np.concatenate(([0], np.cumsum(sizes)))
    out["batch_xi"], out["batch_zi"], out["batch_bounds"], out["batch_p"] = xi, zi, bounds, np.array(p)
    loader = [(torch.as_tensor(xi[a:b]), torch.as_tensor(zi[a:b])) for a, b in zip(bounds[:-1], bounds[1:])]
    k = make_kernel(p)
    thetas = np.stack([theta_aniso(d) + 0.2 * rng.standard_normal(d + 1) for _ in range(3)])
    out["batch_thetas"] = thetas
    for name, model, crit_fn in (
        ("nll", gp.core.Model(None, k, None, None, "zero"), gp.kernel.negative_log_likelihood_zero_mean),
        ("reml", gp.core.Model(constant_mean, k, None, None, "linear_predictor"), gp.kernel.negative_log_restricted_likelihood),
    ):
        ev, pre, nograd, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit_fn, dataloader=loader)
        vals, grads, ng = [], [], []
        for t in thetas:
            tt = torch.as_tensor(t, dtype=torch.float64)
            vals.append(float(pre(tt)))
            grads.append(tonp(grad(tt)))
            ng.append(float(nograd(tt)))
        out[f"batch_{name}_val"], out[f"batch_{name}_grad"], out[f"batch_{name}_nograd"] = np.array(vals), np.stack(grads), np.array(ng)
        # cycling mode: 3 batches per call over a 4-batch loader -> calls see batches (0,1,2), (3,0,1), (2,3,0)
        ev, pre, nograd, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit_fn, dataloader=loader, batches_per_eval=3)
        tt = torch.as_tensor(thetas[0], dtype=torch.float64)
        cyc_v, cyc_g = [], []
        for _ in range(3):
            cyc_v.append(float(pre(tt)))
            cyc_g.append(tonp(grad(tt)))
        out[f"batch_{name}_cycle_val"], out[f"batch_{name}_cycle_grad"] = np.array(cyc_v), np.stack(cyc_g)
    # second-order autograd Fisher (gpmp/core/fisher.py:158-191) on a small problem
    xs, _ = make_xz(60, 2, 79)
    ks = make_kernel(2)
    ms = gp.core.Model(None, ks, None, None, "zero")
    th = theta_aniso(2) + 0.1
    out["hess_xi"], out["hess_theta"] = xs, th
    out["hess_fisher_torch"] = tonp(ms.fisher_information_torch(torch.as_tensor(xs), torch.as_tensor(th, dtype=torch.float64)))
    path = os.path.join(HERE, "ref_batch.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def gen_gradients_p0():
    """The exponential kernel (p = 0) is not differentiable at coincident points; what the reference's autograd route returns there
    (torch.cdist backward: subgradient 0 on the diagonal) is part of the behaviour to match: ML / REML values + autograd gradients at
    p = 0, plain and noisy kernel.  Own file (ref_gradients_p0.npz), own seed: the other gradient fixtures are unchanged."""
    import torch

    out = {}
    rng = np.random.default_rng(44)
    for tag, (n, d), noisy in (("p0a", (90, 3), False), ("p0b", (200, 5), False), ("p0n", (120, 2), True)):
        xi, zi = make_xz(n, d, 45, noise=0.05 if noisy else 0.0)
        out[f"grad_{tag}_xi"], out[f"grad_{tag}_zi"], out[f"grad_{tag}_p"] = xi, zi, np.array(0)
        k = make_noisy_kernel(0) if noisy else make_kernel(0)
        base = np.concatenate(([0.0, np.log(0.05 ** 2)], theta_aniso(d)[1:])) if noisy else theta_aniso(d)
        thetas = np.stack([base + 0.2 * rng.standard_normal(base.size) for _ in range(3)])
        out[f"grad_{tag}_thetas"] = thetas
        mz = gp.core.Model(None, k, None, None, "zero")
        mc = gp.core.Model(constant_mean, k, None, None, "linear_predictor")
        ml = gp.core.Model(linear_mean, k, None, None, "linear_predictor")
        for name, model, crit_fn in (("nll", mz, gp.kernel.negative_log_likelihood_zero_mean),
                                     ("reml_const", mc, gp.kernel.negative_log_restricted_likelihood),
                                     ("reml_lin", ml, gp.kernel.negative_log_restricted_likelihood)):
            _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit_fn, xi, zi)
            vals, grads = [], []
            for t in thetas:
                tt = torch.as_tensor(t, dtype=torch.float64)
                vals.append(float(pre(tt)))
                grads.append(tonp(grad(tt)))
            out[f"grad_{tag}_{name}_val"], out[f"grad_{tag}_{name}_grad"] = np.array(vals), np.stack(grads)
    path = os.path.join(HERE, "ref_gradients_p0.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def gen_dataloader(out):
    """examples/gpmp_example30_dataloader.py flow at small size: Dataset / DataLoader (no shuffle), loader-based initial
    guess, REMAP selection through the batch criterion (NumPy backend: finite-difference jacobian)."""
    from gpmp.dataloader import Dataset, DataLoader

    xi, zi = make_xz(240, 3, 93)
    out["dl_xi"], out["dl_zi"], out["dl_p"], out["dl_batch"] = xi, zi, np.array(2), np.array(80)
    loader = DataLoader(Dataset(xi, zi), batch_size=80, shuffle=False)
    model = gp.core.Model(constant_mean, make_kernel(2))
    out["dl_len"] = np.array(len(loader))
    out["dl_guess"] = tonp(gp.kernel.anisotropic_parameters_initial_guess(model, dataloader=loader))
    out["dl_guess_zero_mean"] = tonp(gp.kernel.anisotropic_parameters_initial_guess_zero_mean(
        gp.core.Model(None, make_kernel(2), None, None, "zero"), dataloader=loader))
    m0, c0 = gp.kernel.anisotropic_parameters_initial_guess_constant_mean(
        gp.core.Model(param_mean, make_kernel(2), None, None, "parameterized"), dataloader=loader)
    out["dl_guess_cm_mean"], out["dl_guess_cm_cov"] = tonp(m0), tonp(c0)
    model, info = gp.kernel.select_parameters_with_remap(model, dataloader=loader, info=True)
    out["dl_covparam"] = tonp(model.covparam)
    out["dl_crit_opt"] = np.array(float(info.selection_criterion_nograd(model.covparam)))
    out["dl_crit_at_guess"] = np.array(float(info.selection_criterion_nograd(gnp.asarray(out["dl_guess"]))))


def gen_namespace():
    """public names of the reference's NumPy backend namespace (the backend contract), one per line"""
    names = [n for n in dir(gnp) if not n.startswith("_")]
    path = os.path.join(HERE, "ref_gnp_names.txt")
    with open(path, "w") as f:
        f.write("\n".join(names) + "\n")
    print("wrote", path, len(names), "names")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "namespace":
        gen_namespace()
    elif len(sys.argv) > 2 and sys.argv[2] == "dataloader":
        o = {}
        gen_dataloader(o)
        path = os.path.join(HERE, "ref_dataloader.npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in o.items()})
        print("wrote", path, os.path.getsize(path), "bytes,", len(o), "arrays")
    elif len(sys.argv) > 2 and sys.argv[2] == "cpd":
        o = {}
        gen_cpd(o)
        path = os.path.join(HERE, "ref_cpd.npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in o.items()})
        print("wrote", path, os.path.getsize(path), "bytes,", len(o), "arrays")
    elif len(sys.argv) > 2 and sys.argv[2] == "remap_extra":
        o = {}
        gen_remap_extra(o)
        path = os.path.join(HERE, "ref_remap_extra.npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in o.items()})
        print("wrote", path, os.path.getsize(path), "bytes,", len(o), "arrays")
    elif backend == "numpy":
        numpy_pass()
    elif len(sys.argv) > 2 and sys.argv[2] == "batch":
        gen_batch_criterion()
    elif len(sys.argv) > 2 and sys.argv[2] == "gradients_p0":        # (torch backend)
        gen_gradients_p0()
    else:
        torch_pass()
