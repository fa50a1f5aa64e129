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
