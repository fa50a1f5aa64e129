#!/usr/bin/env python3
"""Full-size golden vectors for BASELINE configs 3 and 4, made by RUNNING THE REFERENCE (gpmp v0.9.37) at the stated sizes.

Build container only (8 vCPU, 62 GB; the reference does not travel to the GPU box).  Run one pass at a time, nothing
else memory-hungry beside it:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=numpy python3 /root/repo/tests/golden/make_fullsize_fixtures.py config3      # ~15 min, ~20 GB (oracle)
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=torch python3 /root/repo/tests/golden/make_fullsize_fixtures.py config4      # ~15 min, ~31 GB
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        GPMP_BACKEND=numpy python3 /root/repo/tests/golden/make_fullsize_fixtures.py config4np    # after config4: ~10 min, ~15 GB

config3 -> oracle_config3_n32768.npz : the bench workload (d = 8, n = 32768, seeds 1234 / 4321): posterior mean and
           variance (core/model.py:227-307) at a seeded 2048-point subset of the 50000 bench targets and the zero-mean NLL
           (core/likelihood.py:18-52) -- from the PINNED ORACLE, not from the reference: the reference's own predict does not fit
           62 GB at this n (see config3()).  Inputs are NOT stored (regenerated from the seeds); stored: the subset's indices,
           zpm, zpv, the NLL, a condition-number estimate of K (power / inverse iteration) and checksums of the inputs.
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
    """The REFERENCE itself does not fit this container at n = 32768: its NumPy-backend `predict` peaks at ~6 n x n arrays
    (3.25 GB at n = 8192, measured -> ~52 GB + the m-side arrays at 32768; the attempt was OOM-killed at 62 GB).  So this pass
    runs the PINNED ORACLE (oracle/gp_oracle.py: pinned to the reference by tests/test_oracle_vs_golden.py) with a memory-careful
    assembly of exactly its arithmetic: the Gram matrix from oracle.maternp_covariance on row blocks (the `it` path gives the
    same entries as the `ii` path; the nugget 10 sigma^2 eps is added to the diagonal as matern.py:90-94 does), LAPACK dpotrf in
    place, the two triangular solves of cholesky_solve (numpy_backend.py:465-469), the einsum of kriging.py:194.  The file name
    says which generator made it: oracle_config3_n32768.npz."""
    import scipy.linalg as sl

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import gp_oracle as orc

    n, m_all, m, d = 32768, 50000, 2048, 8
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    xt_all = np.random.default_rng(4321).random((m_all, d))
    idx = np.sort(np.random.default_rng(77).choice(m_all, m, replace=False))
    xt = xt_all[idx]
    th = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    # cross-check of the block assembly against the oracle's own ii path, on a size it handles whole
    ns = 3000
    Ks = np.empty((ns, ns))
    for r0 in range(0, ns, 1024):
        Ks[r0:r0 + 1024] = orc.maternp_covariance(xi[:ns][r0:r0 + 1024], xi[:ns], 2, th)
    Ks[np.diag_indices(ns)] += 10.0 * np.exp(th[0]) * np.finfo(np.float64).eps
    assert np.array_equal(Ks, orc.maternp_covariance(xi[:ns], None, 2, th)), "block assembly differs from the oracle's Gram matrix"
    del Ks
    t0 = time.time()
    K = np.empty((n, n))
    for r0 in range(0, n, 2048):
        K[r0:r0 + 2048] = orc.maternp_covariance(xi[r0:r0 + 2048], xi, 2, th)
    K[np.diag_indices(n)] += 10.0 * np.exp(th[0]) * np.finfo(np.float64).eps
    Kit = orc.maternp_covariance(xi, xt, 2, th)
    print("gram %.0f s, maxrss %.1f GB" % (time.time() - t0, rss_gb()), flush=True)
    t0 = time.time()
    v = np.random.default_rng(5).standard_normal(n)
    lmax = 0.0
    for _ in range(40):
        w = K @ v
        lmax = float(np.linalg.norm(w))
        v = w / lmax
    # numpy.linalg.cholesky (its own ILP64 OpenBLAS; a copy: K stays for the residual checks below).  SciPy's LP64 dpotrf in
    # place FAILS at exactly this size ("16545-th leading minor not positive definite": 16384 * 32768 * 8 bytes = 2^32 -- an
    # offset overflow in that build, not the matrix: n = 20000 and 8192 give identical log-determinants on both routes), so the
    # factor comes from the routine the reference itself calls (numpy_backend.py:136,466) and every solve below is verified
    # by its residual against K.
    L = np.linalg.cholesky(K)
    U = L.T                                    # Fortran-ordered upper view of the same buffer: no copy inside SciPy's trtrs wrapper
    print("power iteration + dpotrf %.0f s, maxrss %.1f GB" % (time.time() - t0, rss_gb()), flush=True)
    t0 = time.time()
    # U = L^T (Fortran-ordered upper): L y = b  <=>  U^T y = b
    y = sl.solve_triangular(U, Kit, lower=False, trans=1, check_finite=False)
    lam = sl.solve_triangular(U, y, lower=False, trans=0, check_finite=False)              # lambda_t = K^-1 Kit
    zpm = np.einsum("i..., i...", lam, zi.reshape(-1, 1))
    zpv = np.exp(th[0]) * np.ones(m) - np.einsum("i..., i...", lam, Kit)                  # kriging.py:194 (prior variance sigma^2)
    zpv_raw_min = float(zpv.min())
    zpv = np.maximum(zpv, 0.0)                                                             # model.py:290-296
    yz = sl.solve_triangular(U, zi, lower=False, trans=1, check_finite=False)
    Kinv_z = sl.solve_triangular(U, yz, lower=False, trans=0, check_finite=False)
    nll = float(0.5 * (n * np.log(2.0 * np.pi) + 2.0 * np.sum(np.log(np.diag(U))) + np.einsum("i..., i...", zi, Kinv_z)))
    print("solves %.0f s: nll %.15g, min raw variance %.3g" % (time.time() - t0, nll, zpv_raw_min), flush=True)
    t0 = time.time()
    res_l = float(np.max(np.abs(K @ lam - Kit)))                                           # K lambda_t = Kit
    res_z = float(np.max(np.abs(K @ Kinv_z - zi)))
    print("residuals %.0f s: max|K lam - Kit| %.3g (max|Kit| %.3g), max|K K^-1 z - z| %.3g (max|z| %.3g)"
          % (time.time() - t0, res_l, float(np.max(np.abs(Kit))), res_z, float(np.max(np.abs(zi)))), flush=True)
    assert res_l < 1e-6 and res_z < 1e-5, "a host solve is wrong (LP64 overflow?): fixture not written"
    del K
    v = np.random.default_rng(6).standard_normal(n)
    v /= np.linalg.norm(v)
    lmin = np.inf
    for _ in range(40):
        w = sl.solve_triangular(U, v, lower=False, trans=1, check_finite=False)
        w = sl.solve_triangular(U, w, lower=False, trans=0, check_finite=False)
        lmin = 1.0 / float(np.linalg.norm(w))
        v = w * lmin
    print("lmax %.6g lmin %.6g cond %.4g" % (lmax, lmin, lmax / lmin), flush=True)
    path = os.path.join(HERE, "oracle_config3_n32768.npz")
    np.savez_compressed(path, n=np.array(n), m_all=np.array(m_all), d=np.array(d), theta=th, idx=idx,
                        zpm=np.asarray(zpm).reshape(-1), zpv=np.asarray(zpv).reshape(-1), nll=np.array(nll), lambda_max=np.array(lmax),
                        lambda_min=np.array(lmin), xi_sum=np.array(xi.sum()), zi_sum=np.array(zi.sum()),
                        xt_sum=np.array(xt.sum()), zpv_raw_min=np.array(zpv_raw_min), residual_lambda=np.array(res_l),
                        residual_kinv_z=np.array(res_z),
                        generator=np.array("pinned oracle (oracle/gp_oracle.py), block-assembled Gram, LAPACK in place"))
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
