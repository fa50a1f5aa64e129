#!/usr/bin/env python3
"""Config 1 of BASELINE.json on the HIP path: the flow of the reference's
examples/gpmp_example02_1d_interpolation.py (n = 6 observations of the two-bumps function, Matern p = 3,
constant mean, parameters selected by REML, prediction on a 200-point grid) -- written exactly as a GPmp
user writes it, with ``gpmp_amd`` in place of ``gpmp``.

    python examples/example02_1d_interpolation.py            # needs a MI355X
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpmp_amd as gp          # noqa: E402
import gpmp_amd.num as gnp     # noqa: E402


def twobumps(x):
    """The reference's test function (gpmp/misc/testfunctions.py:15), restated for the example."""
    x = np.asarray(x)
    return (-(0.7 * x + np.sin(5 * x + 1) + 0.1 * np.sin(10 * x))).reshape(-1)


def constant_mean(x, param):
    return gnp.ones((x.shape[0], 1))


def kernel(x, y, covparam, pairwise=False):
    p = 3
    return gp.kernel.maternp_covariance(x, y, p, covparam, pairwise)


def main():
    xt = np.linspace(-1.0, 1.0, 200).reshape(-1, 1)
    zt = twobumps(xt)
    rng = np.random.default_rng(0)
    xi = np.sort(rng.uniform(-1.0, 1.0, size=(6, 1)), axis=0)
    zi = twobumps(xi)

    model = gp.Model(constant_mean, kernel)                                  # generic callable: SciPy FD jacobian
    model, info = gp.kernel.select_parameters_with_reml(model, xi, zi, info=True)
    zpm, zpv = model.predict(xi, zi, xt)

    fast = gp.Model(constant_mean, gp.kernel.MaternCovariance(3))            # declared Matern: analytic jacobian
    fast, info2 = gp.kernel.select_parameters_with_reml(fast, xi, zi, info=True)

    print("covparam (FD jacobian)      :", np.asarray(model.covparam), "evals", len(info["history_criterion"]))
    print("covparam (analytic jacobian):", np.asarray(fast.covparam), "evals", len(info2["history_criterion"]))
    print("REML at optimum             :", float(model.negative_log_restricted_likelihood(model.covparam, xi, zi)))
    print("max |posterior mean - truth|:", float(np.max(np.abs(zpm - zt))), " max posterior sd:", float(np.sqrt(zpv.max())))
    zloo, s2loo, eloo = model.loo(xi, zi, convert_out=True)
    print("LOO errors                  :", eloo)


if __name__ == "__main__":
    main()
