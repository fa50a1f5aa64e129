#!/usr/bin/env python3
"""Full-size golden vectors for BASELINE config 4, made by RUNNING THE REFERENCE (gpmp v0.9.37) at the stated size.

Build container only (8 vCPU, 62 GB; the reference does not travel to the GPU box).  Run one pass at a time:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=torch python3 /root/repo/tests/golden/make_fullsize_fixtures.py config4      # ~15 min, ~31 GB
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=numpy python3 /root/repo/tests/golden/make_fullsize_fixtures.py config4np    # after config4: ~10 min, ~15 GB

config4 -> ref_config4_n16384.npz : d = 20, n = 16384, rho_j in [0.5, 1.5]; the reference's torch-CPU backend: ML (zero mean)
           and REML (constant mean) criterion values + autograd gradients (num/torch_backend.py:574-604 through
           kernel/parameter_selection.py:35-124) at theta and at one perturbed parameter vector; cond(K) estimate.
config4np -> adds to ref_config4_n16384.npz the criterion VALUES of the NumPy backend at the same parameter vectors (the
           parity target BASELINE.json names; its `cdist` takes direct differences, the torch backend's expands the norms):
           ml_val_numpy, reml_val_numpy (the two backends agree to 6e-16 relative here).

Config 3 (n = 32768) is NOT made here: the reference's NumPy-backend `predict` peaks at ~6 n x n arrays (3.25 GB at n = 8192,
measured: ~52 GB at 32768) and -- what ends the attempt before memory does -- LAPACK's dpotrf does not survive n = 32768 in this
container: numpy.linalg.cholesky segfaults and SciPy's reports "16545-th leading minor not positive definite" (16384 * 32768 * 8
bytes = 2^32: an offset overflow with 8 threads; n = 20000 factors fine on both).  Its vector comes from the PINNED ORACLE run on
the GPU box's host, where the same call works: tests/golden/make_oracle_config3.py -> oracle_config3_n32768.npz.

Only inputs' seeds and outputs (plain arrays) are stored; no reference source is copied.
"""
import os
import resource
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
mode = sys.argv[1]
backend = {"config4": "torch", "config4np": "numpy"}[mode]
os.environ["GPMP_BACKEND"] = backend
os.environ.setdefault("GPMP_LOG_LEVEL", "WARNING")

import gpmp as gp  # noqa: E402  (the reference)
import gpmp.num as gnp  # noqa: E402

assert gnp._gpmp_backend_ == backend, (gnp._gpmp_backend_, backend)


def rss_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def kernel(x, y, covparam, pairwise=False):
    return gp.kernel.maternp_covariance(x, y, 2, covparam, pairwise)


def cond_estimate(K):
    """lambda_max by power iteration on K, lambda_min by inverse iteration through LAPACK's factor (K is overwritten)."""
    import scipy.linalg as sl

    n = K.shape[0]
    v = np.random.default_rng(5).standard_normal(n)
    lmax = 0.0
    for _ in range(40):
        w = K @ v
        lmax = float(np.linalg.norm(w))
        v = w / lmax
    L = sl.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    v = np.random.default_rng(6).standard_normal(n)
    v /= np.linalg.norm(v)
    lmin = np.inf
    for _ in range(40):
        w = sl.solve_triangular(L, v, lower=True, check_finite=False)
        w = sl.solve_triangular(L, w, lower=True, trans=1, check_finite=False)
        lmin = 1.0 / float(np.linalg.norm(w))
        v = w * lmin
    return lmax, lmin


def config4():
    import torch

    n, d = 16384, 20
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    th = np.concatenate(([0.0], -np.log(0.5 + np.arange(d) / (d - 1.0))))
    thetas = np.stack([th, th + 0.1 * np.random.default_rng(1234).standard_normal(d + 1)])

    def cm(x, param):
        return gnp.ones((x.shape[0], 1))

    out = {"n": np.array(n), "d": np.array(d), "thetas": thetas, "xi_sum": np.array(xi.sum()), "zi_sum": np.array(zi.sum()),
           "generator": np.array("reference gpmp 0.9.37, torch-CPU backend (autograd)")}
    for name, model, fn in (
        ("ml", gp.core.Model(None, kernel, None, None, "zero"), gp.kernel.negative_log_likelihood_zero_mean),
        ("reml", gp.core.Model(cm, kernel, None, None, "linear_predictor"), gp.kernel.negative_log_restricted_likelihood),
    ):
        _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(model, fn, xi, zi)
        vals, grads = [], []
        for t in thetas:
            t0 = time.time()
            tt = torch.as_tensor(t, dtype=torch.float64)
            vals.append(float(pre(tt)))
            g = grad(tt)
            grads.append(np.asarray(g.detach() if hasattr(g, "detach") else g, dtype=np.float64))
            print("%s value %.15g  |g| %.6g  %.0f s, maxrss %.1f GB" % (name, vals[-1], np.linalg.norm(grads[-1]), time.time() - t0, rss_gb()),
                  flush=True)
        out[f"{name}_val"], out[f"{name}_grad"] = np.array(vals), np.stack(grads)
    conds = []
    for t in thetas:
        with torch.no_grad():
            K = gnp.to_np(kernel(gnp.asarray(xi), None, gnp.asarray(t))).copy()
        lmax, lmin = cond_estimate(K)
        del K
        conds.append((lmax, lmin))
        print("lmax %.6g lmin %.6g cond %.4g" % (lmax, lmin, lmax / lmin), flush=True)
    out["lambda_max_min"] = np.array(conds)
    path = os.path.join(HERE, "ref_config4_n16384.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def config4np():
    path = os.path.join(HERE, "ref_config4_n16384.npz")
    g = dict(np.load(path))
    n, d = int(g["n"]), int(g["d"])
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    assert xi.sum() == float(g["xi_sum"]) and zi.sum() == float(g["zi_sum"])

    def cm(x, param):
        return gnp.ones((x.shape[0], 1))

    mz = gp.core.Model(None, kernel, None, None, "zero")
    mc = gp.core.Model(cm, kernel, None, None, "linear_predictor")
    ml, reml = [], []
    for t in g["thetas"]:
        t0 = time.time()
        ml.append(float(mz.negative_log_likelihood_zero_mean(t, xi, zi)))
        reml.append(float(mc.negative_log_restricted_likelihood(t, xi, zi)))
        print("numpy backend: ml %.15g reml %.15g  %.0f s, maxrss %.1f GB" % (ml[-1], reml[-1], time.time() - t0, rss_gb()), flush=True)
    g["ml_val_numpy"], g["reml_val_numpy"] = np.array(ml), np.array(reml)
    print("torch - numpy backend: ml", g["ml_val"] - g["ml_val_numpy"], "reml", g["reml_val"] - g["reml_val_numpy"])
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes")


{"config4": config4, "config4np": config4np}[mode]()
