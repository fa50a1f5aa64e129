#!/usr/bin/env python3
"""The flows of the reference's examples/gpmp_example05_1d_custom_kernel.py (user-written Matern kernel with its own
nugget, model parameters given as a ``gnp.array``) and examples/gpmp_example10_sample_paths.py (unconditional sample
paths on a grid, conditioned on 5 observations by kriging) on the HIP path, written as a GPmp user writes them.

    python examples/example10_sample_paths.py            # needs a MI355X
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


def constant_mean(x, param):
    return gnp.ones((x.shape[0], 1))


def kernel(x, y, covparam, pairwise=False):
    return gp.kernel.maternp_covariance(x, y, 2, covparam, pairwise)


def custom_kernel(x, y, param, pairwise=False):
    """examples/gpmp_example05_1d_custom_kernel.py:60-111: Matern p = 2 with a 100 eps nugget, from primitives."""
    p, sigma2, loginvrho = 2, gnp.exp(param[0]), param[1]
    if y is x or y is None:
        if pairwise:
            return sigma2 * gnp.ones((x.shape[0],))
        D = gnp.scaled_distance(loginvrho, x, x)
        return sigma2 * gp.kernel.maternp_kernel(p, D) + 100 * gnp.eps * gnp.eye(D.shape[0])
    D = gnp.scaled_distance_elementwise(loginvrho, x, y) if pairwise else gnp.scaled_distance(loginvrho, x, y)
    return sigma2 * gp.kernel.maternp_kernel(p, D)


def main():
    nt = 200
    xt = np.linspace(-1, 1, nt).reshape(-1, 1)
    zt = twobumps(xt)
    ind = [10, 45, 100, 130, 155]
    xi, zi = xt[ind], zt[ind]

    covparam = gnp.array([math.log(0.5 ** 2), math.log(1 / 0.7)])

    # example05: prediction with the user-written kernel
    model5 = gp.core.Model(constant_mean, custom_kernel, None, covparam)
    zpm5, zpv5 = model5.predict(xi, zi, xt)
    print("example05: max |mean - truth| %.4f, interpolation error at the observations %.2e" % (
        float(np.max(np.abs(zpm5 - zt))), float(np.max(np.abs(zpm5[ind] - zi)))))

    # example10: unconditional and conditional sample paths
    model = gp.core.Model(constant_mean, kernel, None, covparam)
    n_samplepaths = 6
    zsim = model.sample_paths(xt, n_samplepaths, method="chol")
    zpm, zpv, lambda_t = model.predict(xi, zi, xt, return_lambdas=True)
    zpsim = model.conditional_sample_paths(zsim, ind, zi, gnp.arange(xt.shape[0]), lambda_t)
    zpsim = gnp.to_np(zpsim)
    print("example10: %d conditional paths of length %d; they pass through the observations to %.2e; spread at the widest "
          "point %.3f (posterior sd %.3f)" % (zpsim.shape[1], zpsim.shape[0], float(np.max(np.abs(zpsim[ind] - zi[:, None]))),
                                              float(zpsim.std(axis=1).max()), float(np.sqrt(zpv.max()))))


if __name__ == "__main__":
    main()
