#!/usr/bin/env python3
"""Full-size vector for BASELINE config 3 from the PINNED ORACLE (oracle/gp_oracle.py; pinned to the reference by
tests/test_oracle_vs_golden.py) -- needs no reference and no GPU, only a host whose LAPACK factors n = 32768 (the build
container's does not, see make_fullsize_fixtures.py; the GPU box's host does):

    gpurun -- 'python tests/golden/make_oracle_config3.py gpurun_out/oracle_config3_n32768.npz'     # ~6 min, ~45 GB of host memory
    cp gpurun_out/oracle_config3_n32768.npz tests/golden/

The bench workload (d = 8, n = 32768, seeds 1234 / 4321, Matern-5/2, the reference's nugget 10 sigma^2 eps): posterior
mean and variance (gpmp/core/model.py:227-307 through kriging.py:35-67,170-199) at a seeded 2048-point subset of the 50000 bench
targets and the zero-mean NLL (gpmp/core/likelihood.py:18-52), with the oracle's own functions: `maternp_covariance` for K and
K(xi, xt), `cholesky_solve` (numpy.linalg.cholesky + two SciPy triangular solves, numpy_backend.py:465-469) ONCE -- the NLL
re-uses that factor instead of building and factoring K a second time as likelihood.py:43-46 does (same matrix, same routine) --
the einsum reductions of kriging.py:194 / model.py:298 / likelihood.py:49.  Every solve is verified by its residual against K
before anything is written; cond(K) comes from power / inverse iteration.  Inputs are NOT stored (regenerated from the seeds)."""
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as orc  # noqa: E402


def main(path):
    from scipy.linalg import solve_triangular

    n, m_all, m, d = 32768, 50000, 2048, 8
    if os.environ.get("GPMP_ORACLE_CONFIG3_N"):          # rehearsal at a smaller size (not a fixture)
        n = int(os.environ["GPMP_ORACLE_CONFIG3_N"])
    rng = np.random.default_rng(1234)
    xi = rng.random((32768, d))[:n]
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    xt_all = np.random.default_rng(4321).random((m_all, d))
    idx = np.sort(np.random.default_rng(77).choice(m_all, m, replace=False))
    xt = xt_all[idx]
    th = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    tick = time.time()

    def lap(what):
        nonlocal tick
        now = time.time()
        print("%-28s %6.0f s   maxrss %.1f GB" % (what, now - tick, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
        tick = now

    K = orc.maternp_covariance(xi, None, 2, th)                     # kriging.py:59, likelihood.py:43
    Kit = orc.maternp_covariance(xi, xt, 2, th)                     # kriging.py:60
    lap("Gram matrices")
    v = np.random.default_rng(5).standard_normal(n)
    lmax = 0.0
    for _ in range(40):
        w = K @ v
        lmax = float(np.linalg.norm(w))
        v = w / lmax
    lap("power iteration")
    lam, L = orc.cholesky_solve(K, Kit)                             # kriging.py:62
    lap("cholesky_solve (m = %d)" % m)
    zpm = np.einsum("i..., i...", lam, zi.reshape(-1, 1)).reshape(-1)                                   # model.py:298
    zpv = orc.maternp_covariance(xt, None, 2, th, True) - np.einsum("i..., i...", lam, Kit)         # kriging.py:193-194
    zpv_raw_min = float(zpv.min())
    zpv = np.maximum(zpv, 0.0)                                                                         # model.py:290-296
    Kinv_z = solve_triangular(L.T, solve_triangular(L, zi, lower=True), lower=False)                  # numpy_backend.py:467-468
    nll = float(0.5 * (n * np.log(2.0 * np.pi) + 2.0 * np.sum(np.log(np.diag(L))) + np.einsum("i..., i...", zi, Kinv_z)))   # likelihood.py:49-51
    res_l = float(np.max(np.abs(K @ lam - Kit)))
    res_z = float(np.max(np.abs(K @ Kinv_z - zi)))
    lap("reductions, NLL, residuals")
    print("nll %.15g   min raw variance %.3g   max|K lam - Kit| %.3g   max|K K^-1 z - z| %.3g" % (nll, zpv_raw_min, res_l, res_z), flush=True)
    assert res_l < 1e-6 and res_z < 1e-5, "a host solve is wrong: fixture not written"
    del K
    v = np.random.default_rng(6).standard_normal(n)
    v /= np.linalg.norm(v)
    lmin = np.inf
    for _ in range(40):
        w = solve_triangular(L.T, solve_triangular(L, v, lower=True), lower=False)
        lmin = 1.0 / float(np.linalg.norm(w))
        v = w * lmin
    lap("inverse iteration")
    print("lambda_max %.6g  lambda_min %.6g  cond %.4g" % (lmax, lmin, lmax / lmin), flush=True)
    np.savez_compressed(path, n=np.array(n), m_all=np.array(m_all), d=np.array(d), theta=th, idx=idx, zpm=zpm, zpv=zpv, nll=np.array(nll),
                        lambda_max=np.array(lmax), lambda_min=np.array(lmin), xi_sum=np.array(xi.sum()), zi_sum=np.array(zi.sum()),
                        xt_sum=np.array(xt.sum()), zpv_raw_min=np.array(zpv_raw_min), residual_lambda=np.array(res_l),
                        residual_kinv_z=np.array(res_z),
                        generator=np.array("pinned oracle (oracle/gp_oracle.py) on the GPU box's host cores; tests/golden/make_oracle_config3.py"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_config3_n32768.npz"))
