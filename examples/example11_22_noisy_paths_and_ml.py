#!/usr/bin/env python3
"""Two more reference flows on the HIP path, written as a GPmp user writes them:

* examples/gpmp_example11_sample_paths_noisy_obs.py -- heteroscedastic noise carried as an extra input column, user kernel
  from primitives (``gnp.diag(noise_variance)``), sample paths by the SVD route on the union of observation and
  prediction points, conditioning by kriging;
* examples/gpmp_example22_1d_interpolation_variation_ml.py -- ML over [constant mean, covparam] with a parameterized mean
  (``param * gnp.ones``), initial guess by GLS, ``make_selection_criterion_with_gradient(parameterized_mean=True)`` +
  ``autoselect_parameters``, prediction and leave-one-out.

    python examples/example11_22_noisy_paths_and_ml.py            # needs a MI355X
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpmp_amd as gp          # noqa: E402
import gpmp_amd.num as gnp     # noqa: E402


def twobumps(x):
    x = np.asarray(x)
    return (-(0.7 * x + np.sin(5 * x + 1) + 0.1 * np.sin(10 * x))).reshape(-1)


# ------------------------------------------------------------------ example 11
def kernel11(x, y, param, pairwise=False):
    p, sigma2, loginvrho = 2, gnp.exp(param[0]), param[1]
    if y is x or y is None:
        noise_variance = gnp.asarray(x[:, -1])
        if pairwise:
            return sigma2 * gnp.ones((x.shape[0],)) + x[:, -1]
        K = gnp.scaled_distance(loginvrho, x[:, :-1], x[:, :-1])
        return sigma2 * gp.kernel.maternp_kernel(p, K) + gnp.diag(noise_variance)
    if pairwise:
        K = gnp.scaled_distance_elementwise(loginvrho, x[:, :-1], y[:, :-1])
    else:
        K = gnp.scaled_distance(loginvrho, x[:, :-1], y[:, :-1])
    return sigma2 * gp.kernel.maternp_kernel(p, K)


def ones_mean(x, param):
    return gnp.ones((x.shape[0], 1))


def example11():
    rng = np.random.default_rng(11)
    nt = 200
    xt1 = np.linspace(-1, 1, nt).reshape(-1, 1)
    zt = twobumps(xt1)
    xt = np.hstack((xt1, np.zeros((nt, 1))))                     # prediction of the noise-free function
    xi1 = np.sort(rng.uniform(-1, 1, size=(30, 1)), axis=0)
    noise_std = 0.05 + 0.15 * (xi1 + 1) / 2                      # grows from left to right
    xi = np.hstack((xi1, noise_std ** 2))
    zi = twobumps(xi1).reshape(-1, 1) + noise_std * rng.standard_normal((30, 1))

    covparam = gnp.array([math.log(0.5 ** 2), math.log(1 / 0.7)])
    model = gp.core.Model(ones_mean, kernel11, None, covparam)
    zpm, zpv, lambda_t = model.predict(xi, zi, xt, return_lambdas=True)
    ni = xi.shape[0]
    xixt = np.vstack((xi, xt))
    xi_ind, xt_ind = np.arange(ni), np.arange(nt) + ni
    zsim = model.sample_paths(xixt, 3, method="svd")
    zpsim = gnp.to_np(model.conditional_sample_paths(zsim, xi_ind, zi, xt_ind, lambda_t))
    print("example11: RMSE of the posterior mean %.3f; 3 conditional paths %s, their mean departs from the posterior mean by "
          "%.3f (posterior sd up to %.3f)" % (float(np.sqrt(np.mean((zpm - zt) ** 2))), zpsim.shape,
                                             float(np.max(np.abs(zpsim.mean(axis=1) - zpm))), float(np.sqrt(zpv.max()))))


# ------------------------------------------------------------------ example 22
def constant_mean(x, param):
    return param * gnp.ones((x.shape[0], 1))


def kernel22(x, y, covparam, pairwise=False):
    return gp.kernel.maternp_covariance(x, y, 3, covparam, pairwise)


def example22():
    rng = np.random.default_rng(22)
    xt = np.linspace(-1, 1, 200).reshape(-1, 1)
    c = 3.0
    zt = twobumps(xt) + c
    xi = np.sort(rng.uniform(-1, 1, size=(8, 1)), axis=0)
    zi = twobumps(xi) + c

    model = gp.core.Model(constant_mean, kernel22, None, None, meantype="parameterized")
    meanparam0, covparam0 = gp.kernel.anisotropic_parameters_initial_guess_constant_mean(model, xi, zi)
    param0 = gnp.concatenate((meanparam0, covparam0))
    nll, nll_pregrad, nll_nograd, dnll = gp.kernel.make_selection_criterion_with_gradient(
        model, gp.kernel.negative_log_likelihood, xi, zi, parameterized_mean=True, meanparam_len=1)
    param_ml, info = gp.kernel.autoselect_parameters(param0, nll_pregrad, dnll, silent=True, info=True)
    model.meanparam = gnp.asarray(param_ml[0])
    model.covparam = gnp.asarray(param_ml[1:])
    zpm, zpv = model.predict(xi, zi, xt)
    zloom, zloov, eloo = model.loo(xi, zi)
    print("example22: ML mean %.3f (offset %.1f), covparam %s, NLL %.4f; max |mean - truth| %.3f; LOO RMSE %.3f" % (
        float(param_ml[0]), c, np.round(np.asarray(param_ml[1:]), 3), float(nll_nograd(param_ml)),
        float(np.max(np.abs(zpm - zt))), float(np.sqrt(np.mean(gnp.to_np(eloo) ** 2)))))


def main():
    example11()
    example22()


if __name__ == "__main__":
    main()
