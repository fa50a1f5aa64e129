#!/usr/bin/env python3
"""Two more reference flows on the HIP path:

* examples/gpmp_example03_2d.py -- a 2-D test function observed at 40 low-discrepancy-like points, parameters selected
  with ``select_parameters_with_remap`` (REML + priors on log sigma^2 and log rho), prediction on an 80 x 80 grid, LOO;
* examples/gpmp_example06_1d_regression.py -- noisy observations handled by SIDE INFORMATION: the last input column flags
  observed (1) / latent (0) points and the user kernel adds the noise variance on flagged points only; covparam given as a
  plain NumPy vector, ``meantype="linear_predictor"`` passed explicitly.

    python examples/example03_06_remap_2d_and_side_information.py            # needs a MI355X
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpmp_amd as gp          # noqa: E402
import gpmp_amd.num as gnp     # noqa: E402


def constant_mean(x, _):
    return gnp.ones((x.shape[0], 1))


# ------------------------------------------------------------------ example 03
def branin_like(x):
    """A smooth 2-D function on [-1, 1]^2 (the reference uses gp.misc.testfunctions)."""
    x1, x2 = 7.5 * x[:, 0] + 2.5, 7.5 * x[:, 1] + 7.5
    return ((x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2 + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10) / 50.0


def kernel03(x, y, covparam, pairwise=False):
    return gp.kernel.maternp_covariance(x, y, 4, covparam, pairwise)


def example03():
    rng = np.random.default_rng(3)
    g = np.linspace(-1, 1, 80)
    xt = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    zt = branin_like(xt)
    xi = rng.uniform(-1, 1, size=(40, 2))
    zi = branin_like(xi)
    model = gp.Model(constant_mean, kernel03)
    model, info = gp.kernel.select_parameters_with_remap(model, xi, zi, info=True)
    zpm, zpv = model.predict(xi, zi, xt)
    zloom, zloov, eloo = model.loo(xi, zi)
    print("example03: covparam %s after %d evaluations; RMSE on the 80 x 80 grid %.4f (function range %.2f); LOO RMSE %.4f" % (
        np.round(np.asarray(gnp.to_np(model.covparam)), 3), len(info["history_criterion"]), float(np.sqrt(np.mean((zpm - zt) ** 2))),
        float(zt.max() - zt.min()), float(np.sqrt(np.mean(gnp.to_np(eloo) ** 2)))))


# ------------------------------------------------------------------ example 06
def _split(x):
    return x[:, :-1], x[:, -1].reshape(-1)


def kernel06(x, y, param, pairwise=False):
    p, sigma2, loginvrho = 2, math.exp(param[0]), param[2:]
    if y is x or y is None:
        noise_var = math.exp(param[1])
        x_coord, flag = _split(x)
        if pairwise:
            return sigma2 * gnp.ones((x_coord.shape[0],)) + noise_var * flag
        D = gnp.scaled_distance(loginvrho, x_coord, x_coord)
        return sigma2 * gp.kernel.maternp_kernel(p, D) + gnp.diag(noise_var * flag)
    (x_coord, _), (y_coord, _) = _split(x), _split(y)
    D = gnp.scaled_distance_elementwise(loginvrho, x_coord, y_coord) if pairwise else gnp.scaled_distance(loginvrho, x_coord, y_coord)
    return sigma2 * gp.kernel.maternp_kernel(p, D)


def example06():
    rng = np.random.default_rng(6)
    noise_std = 1e-1
    xt = np.linspace(-1, 1, 200).reshape(-1, 1)
    zt = (-(0.7 * xt + np.sin(5 * xt + 1) + 0.1 * np.sin(10 * xt))).reshape(-1)
    xi = np.sort(rng.uniform(-1, 1, size=(30, 1)), axis=0)
    zi = (-(0.7 * xi + np.sin(5 * xi + 1) + 0.1 * np.sin(10 * xi))).reshape(-1) + noise_std * rng.standard_normal(30)
    xi_side = np.hstack((xi, np.ones((30, 1))))
    xt_side = np.hstack((xt, np.zeros((200, 1))))
    covparam = np.array([math.log(0.5 ** 2), 2.0 * math.log(noise_std), math.log(1 / 0.7)])
    model = gp.core.Model(constant_mean, kernel06, None, covparam, meantype="linear_predictor")
    zpm, zpv = model.predict(xi_side, zi, xt_side)
    print("example06: RMSE of the latent-function prediction %.4f (noise sd %.2f); posterior sd between %.3f and %.3f" % (
        float(np.sqrt(np.mean((zpm - zt) ** 2))), noise_std, float(np.sqrt(zpv.min())), float(np.sqrt(zpv.max()))))


def main():
    example03()
    example06()


if __name__ == "__main__":
    main()
