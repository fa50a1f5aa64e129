"""Every single-GPU configuration of BASELINE.json at its STATED size.

  config 2  d = 8, n = 4096 / m = 10000: predict + NLL against the CPU oracle (the oracle finishes this size in seconds)
  config 3  d = 8, n = 32768 / m = 50000: the ragged last tile of the prediction (50000 = 390 * 128 + 80) against an
            aligned re-run of the same columns, variance bounds, interpolation at observed points
  config 4  d = 20, n = 16384: ML and REML gradients against 5-point central differences of the HIP value along random
            directions, and tr(K^-1 K) = n on the inverse the gradient uses

Tolerances follow SURVEY.md 8(c): NLL / REML rel 1e-12, mean abs 1e-10 |z|_inf, variance abs 1e-10 sigma^2 for
cond(K) <= 1e6, scaled by cond(K) / 1e6 above that (the condition number is measured in the test).
"""
import math

import numpy as np
import pytest

from tests.helpers import theta_aniso

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd

    return gpmp_amd


def _synth(n, m, d):
    """bench.py's generator."""
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    xt = np.random.default_rng(4321).random((m, d))
    return xi, zi, xt, theta_aniso(d)


# ---------------------------------------------------------------------------------------------- config 2
def test_config2_predict_and_nll_vs_oracle(gp):
    from oracle import gp_oracle as orc

    import gpmp_amd.num as gnp

    n, m, d = 4096, 10000, 8
    xi, zi, xt, th = _synth(n, m, d)
    kern = lambda x, y, t, pairwise=False: orc.maternp_covariance(x, y, 2, t, pairwise)  # noqa: E731
    om = orc.OracleModel(None, kern, None, th, "zero")
    ozpm, ozpv = orc.predict(om, xi, zi, xt)
    onll = float(orc.negative_log_likelihood_zero_mean(om, th, xi, zi))
    ev = np.linalg.eigvalsh(orc.maternp_covariance(xi, None, 2, th))
    cond = ev[-1] / ev[0]
    scale = max(1.0, cond / 1e6)
    model = gp.Model(None, gp.kernel.MaternCovariance(2), None, th, "zero")
    zpm, zpv = model.predict(xi, zi, xt)
    nll = float(model.negative_log_likelihood_zero_mean(th, xi, zi))
    assert abs(nll - onll) < 1e-12 * scale * abs(onll), (nll, onll, cond)
    assert np.max(np.abs(zpm - ozpm)) < 1e-10 * scale * np.max(np.abs(zi)), (np.max(np.abs(zpm - ozpm)), cond)
    assert np.max(np.abs(zpv - ozpv)) < 1e-10 * scale * math.exp(th[0]), (np.max(np.abs(zpv - ozpv)), cond)
    # universal kriging (constant mean), same size, against the oracle's sysv block solve
    cm = lambda x, p: gnp.ones((x.shape[0], 1))  # noqa: E731
    mc = gp.Model(cm, gp.kernel.MaternCovariance(2), None, th, "linear_predictor")
    oc = orc.OracleModel(lambda x, p: np.ones((x.shape[0], 1)), kern, None, th, "linear_predictor")
    zc, vc = mc.predict(xi, zi, xt[:2000])
    ozc, ovc = orc.predict(oc, xi, zi, xt[:2000])
    assert np.max(np.abs(zc - ozc)) < 1e-10 * scale * np.max(np.abs(zi))
    assert np.max(np.abs(vc - ovc)) < 1e-10 * scale * math.exp(th[0])
    oreml = float(orc.negative_log_restricted_likelihood(oc, th, xi, zi))
    reml = float(mc.negative_log_restricted_likelihood(th, xi, zi))
    assert abs(reml - oreml) < 1e-12 * scale * abs(oreml), (reml, oreml, cond)


# ---------------------------------------------------------------------------------------------- config 3
def test_config3_predict_m50000_ragged_tile_and_bounds(gp):
    import gpmp_amd.num as gnp

    n, m, d = 32768, 50000, 8
    xi, zi, xt, th = _synth(n, m, d)
    # the last 64 prediction points are observed points: interpolation inside the ragged tile
    idx = np.random.default_rng(3).choice(n, 64, replace=False)
    xt[-64:] = xi[idx]
    model = gp.Model(None, gp.kernel.MaternCovariance(2), None, th, "zero")
    xid, zid, xtd = gnp.asarray(xi), gnp.asarray(zi), gnp.asarray(xt)
    zpm, zpv = model.predict(xid, zid, xtd, convert_in=False)
    s2 = math.exp(th[0])
    assert zpm.shape == (m,) and np.all(np.isfinite(zpm)) and np.all(np.isfinite(zpv))
    assert np.min(zpv) >= 0.0 and np.max(zpv) <= s2 * (1.0 + 1e-12)
    # the last 208 columns (the 80 of the ragged tile + the full tile before it) against an aligned m = 256 re-run
    zpm2, zpv2 = model.predict(xid, zid, xtd[m - 256:].clone(), convert_in=False)
    assert np.max(np.abs(zpm[m - 208:] - zpm2[-208:])) < 1e-10 * np.max(np.abs(zi))
    assert np.max(np.abs(zpv[m - 208:] - zpv2[-208:])) < 1e-10 * s2
    # a block from the middle of the range against its own small run
    zpm3, zpv3 = model.predict(xid, zid, xtd[25000:25128].clone(), convert_in=False)
    assert np.max(np.abs(zpm[25000:25128] - zpm3)) < 1e-10 * np.max(np.abs(zi))
    assert np.max(np.abs(zpv[25000:25128] - zpv3)) < 1e-10 * s2
    # interpolation at the observed points placed in the ragged tile (cond(K) ~ 1 / nugget: agreement to cond * eps)
    assert np.max(np.abs(zpm[-64:] - zi[idx])) < 1e-6 * np.max(np.abs(zi))
    assert np.max(zpv[-64:]) < 1e-6 * s2


# ---------------------------------------------------------------------------------------------- config 4
def test_config4_gradients_vs_central_differences_n16384_d20(gp):
    import torch

    import gpmp_amd.num as gnp
    from gpmp_amd.core.gradients import MLZeroMeanAnalytic, REMLAnalytic

    n, d = 16384, 20
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    th = np.concatenate(([0.0], -np.log(0.5 + np.arange(d) / (d - 1.0))))       # rho_j in [0.5, 1.5]
    xid, zid = gnp.asarray(xi), gnp.asarray(zi)
    cm = lambda x, p: gnp.ones((x.shape[0], 1))  # noqa: E731
    crits = {
        "ml": MLZeroMeanAnalytic(gp.Model(None, gp.kernel.MaternCovariance(2), None, th, "zero")),
        "reml": REMLAnalytic(gp.Model(cm, gp.kernel.MaternCovariance(2), None, th, "linear_predictor")),
    }
    dirs = np.random.default_rng(7).standard_normal((3, d + 1))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    h = 1e-3
    for name, crit in crits.items():
        value, state = crit.value_and_state(th, xid, zid)
        g = crit.gradient_from_state(state)
        assert g.shape == (d + 1,) and np.all(np.isfinite(g)) and math.isfinite(value)
        del state
        torch.cuda.empty_cache()
        f = lambda t: crit.value_and_state(t, xid, zid)[0]  # noqa: E731
        for u in dirs:
            fd = (-f(th + 2 * h * u) + 8 * f(th + h * u) - 8 * f(th - h * u) + f(th - 2 * h * u)) / (12 * h)
            assert abs(fd - g @ u) < 2e-6 * np.linalg.norm(g), (name, fd, g @ u, np.linalg.norm(g))


def test_config4_inverse_trace_identity_n16384_d20(gp):
    """tr(K^-1 K) = n with K^-1 from potrf -> L^-1 by doubling -> T^T T (the inverse the gradient trace reads)."""
    import torch

    import gpmp_amd.num as gnp

    n, d = 16384, 20
    xi = gnp.asarray(np.random.default_rng(1234).random((n, d)))
    th = np.concatenate(([0.0], -np.log(0.5 + np.arange(d) / (d - 1.0))))
    cov = gp.kernel.MaternCovariance(2)
    K = cov(xi, None, th)
    F = gnp.cholesky_factor(K.clone(), overwrite=True)
    Kinv = F.inverse_lower()
    low = float(torch.sum(torch.tril(Kinv, -1) * K).item())
    dg = float(torch.sum(torch.diagonal(Kinv) * torch.diagonal(K)).item())
    assert abs(2.0 * low + dg - n) < 1e-6 * n, 2.0 * low + dg - n
    # one column of K^-1 K = I through the same inverse: row 12345 of Kinv (lower part mirrored) times K
    r = 12345
    row = torch.cat((Kinv[r, : r + 1], Kinv[r + 1:, r]))
    e = gnp.to_np(gnp.matmul(row, K))
    e[r] -= 1.0
    assert np.max(np.abs(e)) < 1e-6


@pytest.mark.parametrize("n", [9347, 12800, 20000])
def test_cholesky_residual_between_the_stated_sizes(n):
    """Sizes that mix every regime of the look-ahead factorisation in one run -- 1024-column panels with their look-ahead update
    cut in pieces (more than 8192 rows left), the 512 / 256-column tail, a ragged last block (9347 = 73 * 128 + 3) -- checked
    through the residual of sampled rows, || (L L^T - K)[rows] || / || K ||, and one solve against that residual's scale."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd as gp
    import gpmp_amd.num as gnp

    d = 6
    rng = np.random.default_rng(n)
    xi = gnp.asarray(rng.random((n, d)))
    th = np.concatenate(([0.2], -np.log(0.4 * (1.0 + np.arange(d) / d))))
    cov = gp.kernel.MaternCovariance(2)
    K = cov(xi, None, th)
    F = gnp.cholesky_factor(K.clone())
    L = F.L
    rows = torch.as_tensor(np.sort(rng.choice(n, 384, replace=False)), device=L.device)
    cols = torch.arange(n, device=L.device)
    Lr = torch.where(cols[None, :] <= rows[:, None], L[rows], torch.zeros((), dtype=L.dtype, device=L.device))
    R = gnp.matmul(Lr, torch.tril(L).T.contiguous()) - K[rows]
    assert float(R.abs().max()) / float(K.abs().max()) < 5e-13
    b = gnp.asarray(rng.standard_normal(n))
    x = F.solve(b)
    r = gnp.matmul(K, x.reshape(-1, 1)).reshape(-1) - b
    assert float(r.abs().max()) < 1e-7 * float(b.abs().max())
