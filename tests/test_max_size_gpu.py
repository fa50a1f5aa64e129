"""Beyond BASELINE's n: one n = 65536 factorisation (K = 34 GB, 2^32 elements: every element offset exceeds 32 bits) checked
through size-independent properties.  This is also the local matrix height of a rank of BASELINE config 5 (n = 131072 on the
2 x 4 grid), whose kernels the one-GPU box cannot otherwise check for VALUES (tools/dist_rank_emulation.py checks faults
and timing only)."""
import math

import numpy as np
import pytest

from tests.helpers import theta_aniso

pytestmark = pytest.mark.gpu

N, D = 65536, 8


def test_cholesky_and_solves_at_65536():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 120e9:
        pytest.skip("needs 120 GB of HBM")
    import gpmp_amd as gp
    import gpmp_amd.num as gnp
    from gpmp_amd.dist import HipLocalOps

    rng = np.random.default_rng(99)
    xi = gnp.asarray(rng.random((N, D)))
    zi_np = np.sin(2 * np.pi * rng.random(N)) + rng.random(N)
    zi = gnp.asarray(zi_np)
    th = theta_aniso(D)
    cov = gp.kernel.MaternCovariance(2, noise=True)
    th2 = np.concatenate(([th[0], math.log(1e-4)], th[1:]))          # noise variance 1e-4 sigma^2
    K = cov(xi, None, th2)
    assert K.shape == (N, N) and K.numel() == 2 ** 32
    F = gnp.cholesky_factor(K.clone())
    L = F.L
    ops = HipLocalOps()
    # (L L^T - K) on 256 sampled rows, including the last ones (largest offsets)
    rows_np = np.concatenate((np.random.default_rng(0).choice(N, 248, replace=False), np.arange(N - 8, N)))
    rows = torch.as_tensor(rows_np, device=L.device)
    cols = torch.arange(N, device=L.device)
    Lr = torch.where(cols[None, :] <= rows[:, None], L[rows], torch.zeros((), dtype=L.dtype, device=L.device))
    ops.lib.gpmp_tril(gnp._ptr(L), N, gnp._ld(L), gnp._stream())
    R = gnp.as_matrix(K[rows], copy=True)
    ops.gemm_nt_sub(R, gnp.as_matrix(Lr, copy=True), L)               # R = K[rows] - Lr L^T
    assert float(R.abs().max()) / float(K.abs().max()) < 1e-12
    del R, Lr
    # forward / backward single-vector solves: K alpha = z
    alpha = F.solve(zi)
    r = zi - gnp.matmul(K, alpha.reshape(-1, 1)).reshape(-1)
    assert float(r.abs().max()) < 1e-7 * float(zi.abs().max())
    # the NLL of the model layer agrees with the pieces above
    model = gp.Model(None, cov, None, th2, "zero")
    nll = float(model.negative_log_likelihood_zero_mean(th2, xi, zi))
    ref = 0.5 * (N * math.log(2 * math.pi) + F.logdet() + float((zi * alpha).sum()))
    assert abs(nll - ref) < 1e-9 * abs(ref)


def test_prediction_with_more_than_2_to_the_31_cross_covariance_entries():
    """n = 65536, m = 36000: the n x m cross-covariance, the many-right-hand-side solve and the column reductions of ONE prediction
    chunk work on 2.36e9 > 2^31 elements (the default chunk budget, 24 GB, admits 3.2e9), so every element offset in the rectangular
    kernels must be 64-bit.  Size-independent check: the same model predicts a strided subset of the points alone (a 9e7-element
    problem on the same factor) -- means and variances must agree to rounding; with weights: K lambda = K(xi, xt) on sampled rows."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs 200 GB of HBM")
    import gpmp_amd as gp
    import gpmp_amd.num as gnp

    m = 36000
    assert N * m > 2 ** 31
    rng = np.random.default_rng(7)
    xi_np, xt_np = rng.random((N, D)), rng.random((m, D))
    zi_np = np.sin(2 * np.pi * xi_np[:, 0]) + xi_np[:, 1:].sum(axis=1)
    th = theta_aniso(D)
    th2 = np.concatenate(([th[0], math.log(1e-4)], th[1:]))
    model = gp.Model(None, gp.kernel.MaternCovariance(2, noise=True), None, th2, "zero")
    xi, zi, xt = gnp.asarray(xi_np), gnp.asarray(zi_np), gnp.asarray(xt_np)
    zpm, zpv, lam = model.predict(xi, zi, xt, return_lambdas=True, convert_out=False)
    assert lam.shape == (N, m)
    sub = np.concatenate((np.arange(0, m, 24), np.arange(m - 8, m)))              # strided + the last columns (largest offsets)
    zpm_s, zpv_s, lam_s = model.predict(xi, zi, gnp.asarray(xt_np[sub]), return_lambdas=True, convert_out=False)
    sub_t = torch.as_tensor(sub, device=zpm.device)
    zs = float(np.abs(zi_np).max())
    assert float((zpm[sub_t] - zpm_s).abs().max()) < 1e-9 * zs
    assert float((zpv[sub_t] - zpv_s).abs().max()) < 1e-9
    assert float((lam[:, sub_t] - lam_s).abs().max()) < 1e-8 * float(lam_s.abs().max())
    assert bool(torch.isfinite(zpm).all()) and float(zpv.min()) >= 0.0 and float(zpv.max()) <= math.exp(th2[0]) * (1 + 1e-12)
    del lam_s
    # the one-call driver (gpmp_predict_zero_mean behind Model.predict without weights) on the same 2.36e9-element problem
    zpm_f, zpv_f = model.predict(xi, zi, xt, convert_out=False)
    assert float((zpm_f - zpm).abs().max()) < 1e-9 * zs and float((zpv_f - zpv).abs().max()) < 1e-9
    # the weights solve K lambda = K(xi, xt): sampled rows of K against ALL columns (rows at the end: the largest offsets)
    rows_np = np.concatenate((np.random.default_rng(1).choice(N, 56, replace=False), np.arange(N - 8, N)))
    cov = gp.kernel.MaternCovariance(2, noise=True)
    Krows = cov(gnp.asarray(xi_np[rows_np]), xi, th2)                              # (64, N) cross block: no noise term ...
    Krows[torch.arange(len(rows_np), device=Krows.device), torch.as_tensor(rows_np, device=Krows.device)] += math.exp(th2[1])   # ... add it
    R = gnp.matmul(Krows, lam) - cov(gnp.asarray(xi_np[rows_np]), xt, th2)
    assert float(R.abs().max()) < 1e-8 * math.exp(th2[0])
