#!/usr/bin/env python3
"""The flow of the reference's examples/gpmp_example07_nd_regression.py on the HIP path: noisy observations of a
d-dimensional function, Matern-5/2 kernel with a homoscedastic noise term written by the USER from gnp primitives
(covparam = [log s2, log s2_noise, log 1/rho_1..d]), constant mean, REML through
``make_selection_criterion_with_gradient`` + ``autoselect_parameters``, prediction at held-out points.

The user-written kernel runs on the generic path (Gram built by the callable's gnp calls, SciPy finite-difference
jacobian, exactly as with the reference's NumPy backend); declaring the same kernel as ``MaternCovariance(2, noise=True)``
switches to the fused Gram kernel and the analytic REML gradient.  Both are run and compared.

    python examples/example07_nd_noisy_regression.py            # needs a MI355X
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpmp_amd as gp          # noqa: E402
import gpmp_amd.num as gnp     # noqa: E402


def constant_mean(x, _):
    return gnp.ones((x.shape[0], 1))


def kernel(x, y, param, pairwise=False):
    """User kernel, same construction as examples/gpmp_example07_nd_regression.py:95-131."""
    p = 2
    sigma2, noise_variance, loginvrho = gnp.exp(param[0]), gnp.exp(param[1]), param[2:]
    if y is x or y is None:
        if pairwise:
            return sigma2 * gnp.ones((x.shape[0],))
        D = gnp.scaled_distance(loginvrho, x, x)
        return sigma2 * gp.kernel.maternp_kernel(p, D) + noise_variance * gnp.eye(D.shape[0])
    D = gnp.scaled_distance_elementwise(loginvrho, x, y) if pairwise else gnp.scaled_distance(loginvrho, x, y)
    return sigma2 * gp.kernel.maternp_kernel(p, D)


def f(x):
    """A smooth anisotropic test function on [0, 1]^d."""
    return np.sin(2 * np.pi * x[:, 0]) + 0.5 * np.cos(3 * x[:, 1]) + x[:, 2:].sum(axis=1) * 0.3


def main():
    rng = np.random.default_rng(7)
    d, ni, nt, noise_std = 4, 400, 2000, 0.1
    xi, xt = rng.random((ni, d)), rng.random((nt, d))
    zi, zt = f(xi) + noise_std * rng.standard_normal(ni), f(xt)

    covparam0 = np.concatenate(([np.log(np.var(zi))], [2 * np.log(0.1) + np.log(np.var(zi))], -np.log(np.std(xi, axis=0))))
    results = {}
    for name, cov in (("user callable (FD jacobian)", kernel), ("MaternCovariance(2, noise=True) (analytic jacobian)",
                                                              gp.kernel.MaternCovariance(2, noise=True))):
        model = gp.Model(constant_mean, cov, None, covparam0)
        crit = gp.kernel.negative_log_restricted_likelihood
        _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit, xi, zi)
        t0 = time.perf_counter()
        covparam, info = gp.kernel.autoselect_parameters(covparam0, pre, grad, silent=True, info=True)
        dt = time.perf_counter() - t0
        model.covparam = gnp.asarray(covparam)
        zpm, zpv = model.predict(xi, zi, xt)
        rmse = float(np.sqrt(np.mean((zpm - zt) ** 2)))
        results[name] = (np.asarray(covparam), rmse)
        print(f"{name}:\n  covparam {np.round(np.asarray(covparam), 4)}\n  noise sd {np.exp(0.5 * covparam[1]):.4f} (truth {noise_std})"
              f"  REML {float(pre(covparam)):.4f}  evals {len(info['history_criterion'])}  {dt:.2f} s  RMSE on {nt} points {rmse:.4f}")
    (c1, r1), (c2, r2) = results.values()
    print("max |covparam difference| between the two routes:", float(np.max(np.abs(c1 - c2))))


if __name__ == "__main__":
    main()
