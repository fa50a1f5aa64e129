"""Shared helpers for the parity tests (data generators identical to tests/golden/make_fixtures.py)."""
import numpy as np


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
    return np.ones((x.shape[0], 1))


def linear_mean(x, param):
    return np.hstack((np.ones((x.shape[0], 1)), np.asarray(x)))


def param_mean(x, param):
    return (param[0] + param[1] * x[:, 0]).reshape(-1, 1)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b)) if b.size else 1.0
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0)) if a.size else 0.0
