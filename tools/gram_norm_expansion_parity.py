#!/usr/bin/env python3
"""Parity argument for a Gram pass whose distances come from the norm expansion  |x|^2 + |y|^2 - 2 x.y  (the form that would
put the 2d multiply-adds per entry of the direct differences on the MFMA pipe) -- evaluated on the CPU in fp64, against the
reference's own vectors (tests/golden/ref_matern.npz: the covariances the reference computed with scipy cdist, i.e. with
direct differences, gpmp/num/numpy_backend.py:432-436).  The Matern part is the oracle's, so every difference below is the
distance formula's.  The bar is the one the HIP kernel is held to (tests/test_hip_parity.py): rel 1e-14 on K.

    python tools/gram_norm_expansion_parity.py            (no GPU)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp_oracle as orc  # noqa: E402


def cov_expansion(x, y, p, theta, centre):
    """sigma^2 Matern_p(h) with h^2 = |xs|^2 + |ys|^2 - 2 xs.ys on scaled inputs; `centre` subtracts the common mean first
    (the cheapest way to shrink the cancelling terms)"""
    inv = np.exp(theta[1:])
    xs, ys = x * inv, y * inv
    if centre:
        c = 0.5 * (xs.mean(axis=0) + ys.mean(axis=0))
        xs, ys = xs - c, ys - c
    h2 = (xs * xs).sum(1)[:, None] + (ys * ys).sum(1)[None, :] - 2.0 * (xs @ ys.T)
    h = np.sqrt(np.maximum(h2, 0.0))
    return np.exp(theta[0]) * orc.maternp_kernel(p, h)


def main():
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ref_matern.npz"))
    worst = 0.0
    for tag in "abc":
        x, y, th, p = g[f"cov_{tag}_x"], g[f"cov_{tag}_y"], g[f"cov_{tag}_theta"], int(g[f"cov_{tag}_p"])
        ref = g[f"cov_{tag}_it"]
        direct = orc.maternp_covariance_it(x, y, p, th)
        for centre in (False, True):
            k = cov_expansion(x, y, p, th, centre)
            rel = np.max(np.abs(k - ref) / np.abs(ref))
            worst = max(worst, rel)
            print(f"case {tag}: n x m = {x.shape[0]} x {y.shape[0]}, d = {x.shape[1]}, p = {p}: norm expansion{' (centred)' if centre else ''}: "
                  f"max rel error {rel:.2e}   (direct differences: {np.max(np.abs(direct - ref) / np.abs(ref)):.2e})")
    # the case that decides it: near points (h -> 0), where K -> sigma^2 and the expansion's absolute error in h^2,
    # eps (|xs|^2 + |ys|^2), is not small against h^2
    rng = np.random.default_rng(0)
    d = 20
    th = np.concatenate(([0.0], -np.log(0.5 + np.arange(d) / (d - 1.0))))        # config 4's length scales
    x = rng.random((400, d))
    y = x + 1e-5 * rng.standard_normal(x.shape)
    ref = orc.maternp_covariance_it(x, y, 2, th)
    for centre in (False, True):
        k = cov_expansion(x, y, 2, th, centre)
        near = np.abs(np.diag(k) - np.diag(ref)) / np.diag(ref)
        print(f"d = 20, pairs 1e-5 apart{' (centred)' if centre else ''}: max rel error on K {near.max():.2e}; on 1 - K/sigma^2 "
              f"{np.max(np.abs((1 - np.diag(k)) - (1 - np.diag(ref))) / (1 - np.diag(ref))):.2e}")
        worst = max(worst, near.max())
    print(f"worst relative error on K: {worst:.2e}  -> {'within' if worst <= 1e-14 else 'OUTSIDE'} the 1e-14 bar of the Gram parity tests")


if __name__ == "__main__":
    main()
