"""Full-size checks at BASELINE.json's n = 32768 through size-independent properties (the oracle cannot
finish at this size in seconds): factor residual, interpolation at observed points, linearity of the
posterior mean in z, agreement of two independent routes to z^T K^-1 z and to the log-determinant."""
import math

import numpy as np
import pytest

from tests.helpers import theta_aniso

pytestmark = pytest.mark.gpu

N, D = 32768, 8


@pytest.fixture(scope="module")
def setup():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd as gp
    import gpmp_amd.num as gnp

    rng = np.random.default_rng(1234)
    xi = rng.random((N, D))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    th = theta_aniso(D)
    return gp, gnp, gnp.asarray(xi), gnp.asarray(zi), th


def test_cholesky_residual_full_size(setup):
    """|| (L L^T - K)[rows] ||_max / ||K||_max on 512 sampled rows, via the library GEMM."""
    import torch

    gp, gnp, xi, zi, th = setup
    cov = gp.kernel.MaternCovariance(2)
    K = cov(xi, None, th)
    F = gnp.cholesky_factor(K.clone())
    L = F.L
    torch.cuda.synchronize()
    rows = torch.as_tensor(np.random.default_rng(0).choice(N, 512, replace=False), device=L.device)
    Lr = L[rows]
    # zero the unspecified strict upper part of the sampled rows
    cols = torch.arange(N, device=L.device)
    Lr = torch.where(cols[None, :] <= rows[:, None], Lr, torch.zeros((), dtype=Lr.dtype, device=Lr.device))
    Lz = torch.tril(L)
    R = gnp.matmul(Lr, Lz.T.contiguous()) - K[rows]
    assert float(R.abs().max()) / float(K.abs().max()) < 5e-13
    # log-determinant: 2 sum log L_ii must equal the sum over diagonal-block pivots seen by the kernel
    assert abs(F.logdet() - 2.0 * float(torch.log(torch.diagonal(L)).sum())) < 1e-8 * abs(F.logdet())


def test_predict_properties_full_size(setup):
    gp, gnp, xi, zi, th = setup
    model = gp.Model(None, gp.kernel.MaternCovariance(2), None, th, "zero")
    m = 1024
    rng = np.random.default_rng(5)
    xt_new = gnp.asarray(rng.random((m, D)))
    # (i) interpolation: predicting at observed points returns the data with ~zero variance
    idx = rng.choice(N, 256, replace=False)
    zpm, zpv = model.predict(xi, zi, xi[idx.tolist()].clone())
    zi_np = gnp.to_np(zi)
    assert np.max(np.abs(zpm - zi_np[idx])) < 1e-6 * np.max(np.abs(zi_np))
    assert np.max(zpv) < 1e-6 and np.min(zpv) >= 0.0
    # (ii) linearity of the posterior mean in z; variance independent of z
    z2 = gnp.asarray(np.cos(3.0 * gnp.to_np(xi)[:, 1]))
    a, va = model.predict(xi, zi, xt_new)
    b, vb = model.predict(xi, z2, xt_new)
    c, vc = model.predict(xi, 2.0 * zi - 0.5 * z2, xt_new)
    assert np.max(np.abs(c - (2.0 * a - 0.5 * b))) < 1e-8 * max(1.0, np.max(np.abs(c)))
    assert np.array_equal(va, vb) and np.array_equal(va, vc)
    assert np.all(va >= 0.0) and np.all(va <= math.exp(th[0]) * (1 + 1e-12))
    # (iii) constant-mean universal kriging reproduces constants exactly (unbiasedness constraint)
    mc = gp.Model(lambda x, p: gnp.ones((x.shape[0], 1)), gp.kernel.MaternCovariance(2), None, th)
    const = gnp.asarray(np.full(N, 3.25))
    zc, _ = mc.predict(xi, const, xt_new)
    assert np.max(np.abs(zc - 3.25)) < 1e-7


def test_nll_routes_agree_full_size(setup):
    """z^T K^-1 z by the forward solve (||L^-1 z||^2) and by the full solve (z^T (L^-T L^-1 z))."""
    gp, gnp, xi, zi, th = setup
    model = gp.Model(None, gp.kernel.MaternCovariance(2), None, th, "zero")
    nll = float(model.negative_log_likelihood_zero_mean(th, xi, zi))
    from gpmp_amd.core.linalg import covariance_factor

    F = covariance_factor(model, xi, th)
    alpha = F.solve(zi)
    quad = float((zi * alpha).sum())
    ref = 0.5 * (N * math.log(2 * math.pi) + F.logdet() + quad)
    assert abs(nll - ref) < 1e-9 * abs(ref)
    zt, zv = model.predict(xi, zi, xi[:8].clone())
    assert np.all(np.isfinite(zt))
