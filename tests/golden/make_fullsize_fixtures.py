#!/usr/bin/env python3
"""Full-size golden vectors for BASELINE configs 3 and 4, made by RUNNING THE REFERENCE (gpmp v0.9.37) at the stated sizes.

Build container only (8 vCPU, 62 GB; the reference does not travel to the GPU box).  Run one pass at a time, nothing
else memory-hungry beside it:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=numpy python3 /root/repo/tests/golden/make_fullsize_fixtures.py config3      # ~25 min, ~50 GB
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=torch python3 /root/repo/tests/golden/make_fullsize_fixtures.py config4      # ~15 min, ~31 GB
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=numpy python3 /root/repo/tests/golden/make_fullsize_fixtures.py config4np    # after config4: ~10 min, ~15 GB

config3 -> ref_config3_n32768.npz : the bench workload (d = 8, n = 32768, seeds 1234 / 4321), the reference's
           NumPy-backend `Model.predict` (core/model.py:227-307) at a seeded 2048-point subset of the 50000 bench targets and
           `negative_log_likelihood_zero_mean` (core/likelihood.py:18-52).  Inputs are NOT stored (regenerated from the seeds);
           stored: the subset's indices, zpm, zpv, the NLL, a condition-number estimate of K (power / inverse iteration on
           the reference's own covariance matrix) and checksums of the inputs.
config4 -> ref_config4_n16384.npz : d = 20, n = 16384, rho_j in [0.5, 1.5]; the reference's torch-CPU backend: ML (zero mean)
           and REML (constant mean) criterion values + autograd gradients (num/torch_backend.py:574-604 through
           kernel/parameter_selection.py:35-124) at theta and at one perturbed parameter vector; cond(K) estimate.

config4np -> adds to ref_config4_n16384.npz the criterion VALUES of the NumPy backend at the same parameter vectors (the
           parity target BASELINE.json names; its `cdist` takes direct differences, the torch backend's expands the norms, so the
           two reference backends agree to ~1e-10 only): ml_val_numpy, reml_val_numpy.

Only inputs' seeds and outputs (plain arrays) are stored; no reference source is copied.
"""
import os
import resource
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
mode = sys.argv[1]
backend = {"config3": "numpy", "config4": "torch", "config4np": "numpy"}[mode]
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


def config3():
    n, m_all, m, d = 32768, 50000, 2048, 8
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    xt_all = np.random.default_rng(4321).random((m_all, d))
    idx = np.sort(np.random.default_rng(77).choice(m_all, m, replace=False))
    xt = xt_all[idx]
    th = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    model = gp.core.Model(None, kernel, None, th, "zero")
    t0 = time.time()
    zpm, zpv = model.predict(xi, zi, xt)
    print("predict %.0f s, maxrss %.1f GB" % (time.time() - t0, rss_gb()), flush=True)
    t0 = time.time()
    nll = float(model.negative_log_likelihood_zero_mean(th, xi, zi))
    print("nll %.0f s, maxrss %.1f GB: %.15g" % (time.time() - t0, rss_gb(), nll), flush=True)
    t0 = time.time()
    K = np.asarray(kernel(xi, None, th))
    lmax, lmin = cond_estimate(K)
    del K
    print("cond %.0f s: lmax %.6g lmin %.6g cond %.4g" % (time.time() - t0, lmax, lmin, lmax / lmin), flush=True)
    path = os.path.join(HERE, "ref_config3_n32768.npz")
    np.savez_compressed(path, n=np.array(n), m_all=np.array(m_all), d=np.array(d), theta=th, idx=idx,
                        zpm=np.asarray(zpm), zpv=np.asarray(zpv), nll=np.array(nll), lambda_max=np.array(lmax),
                        lambda_min=np.array(lmin), xi_sum=np.array(xi.sum()), zi_sum=np.array(zi.sum()),
                        xt_sum=np.array(xt.sum()), generator=np.array("reference gpmp 0.9.37, numpy backend"))
    print("wrote", path, os.path.getsize(path), "bytes")


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


{"config3": config3, "config4": config4, "config4np": config4np}[mode]()
