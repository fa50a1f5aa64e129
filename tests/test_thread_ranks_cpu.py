"""tools/thread_ranks.py without a GPU: the 2 x 4 grid of BASELINE config 5 as eight THREAD-ranks of this process on the CPU stand-in
of the local arithmetic -- the host-staged transport (torch.testing's in-process group) and the meeting logic of the stream-ordered
fabric (no streams on the CPU: program order), through factorisation, NLL, prediction with weights and the REML gradient.  The GPU
tests (tests/test_dist_gpu.py, tests/test_config5_gpu.py) run the same classes with the real kernels and real streams."""
import math

import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests.helpers import make_xz, theta_aniso


def _cov(a, b, t, pairwise=False):
    return orc.maternp_covariance_it(np.asarray(a), np.asarray(b), 2, t)


def _cov_full(a, b, t, pairwise=False):
    if b is None:
        return orc.maternp_covariance(np.asarray(a), None, 2, t, pairwise)
    return orc.maternp_covariance_it(np.asarray(a), np.asarray(b), 2, t, pairwise)


@pytest.mark.parametrize("which", [0, 1], ids=["host_staged", "stream_ordered_meetings"])
def test_thread_ranks_2x4_on_the_cpu_stand_in(which):
    from gpmp_amd.dist import ProcessGrid
    from tests.cpu_local_ops import CpuLocalOps
    from tools import thread_ranks

    pr, pc, n, nb, m = 2, 4, 1300, 128, 131
    x, z = make_xz(n, 3, 7)
    xt, _ = make_xz(m, 3, 8)
    th = theta_aniso(3, scale=0.4)
    P = np.ones((n, 1))
    out = {}

    def body(rank, world, fabric, classes):
        ch = classes[which](ProcessGrid(pr, pc), n, nb=nb, ops=CpuLocalOps())
        ch.backend = "gloo"                      # CPU tensors for the stand-in ops; the transport's own methods are what runs
        ch.build_local_gram(_cov, x, th, 1e-6)
        info = ch.factor()
        nll = ch.negative_log_likelihood(z)
        mean, var, (j0, j1), lam = ch.predict(_cov_full, x, z, xt, th, return_lambdas=True)
        v, g = ch.value_and_grad(x, z, th, 2, P=P)
        gathered = fabric.allgather(rank, (ch.grid.r, j0, j1, mean))
        out[rank] = (info, nll, v, g, gathered)

    errors = thread_ranks.run(pr * pc, body, limit_s=240.0)
    assert not errors, errors[0]
    K = orc.maternp_covariance_it(x, x, 2, th) + 1e-6 * np.eye(n)
    L = np.linalg.cholesky(K)
    w = np.linalg.solve(L, z)
    ref = 0.5 * (n * math.log(2 * math.pi) + 2 * np.log(np.diag(L)).sum() + w @ w)
    zpm = np.full(m, np.nan)
    for (r, j0, j1, mean) in out[0][4]:
        if r == 0:
            zpm[j0:j1] = mean
    rm = orc.maternp_covariance_it(x, xt, 2, th).T @ np.linalg.solve(K, z)
    assert all(o[0] == 0 for o in out.values())
    assert abs(out[0][1] - ref) < 1e-10 * abs(ref) and np.max(np.abs(zpm - rm)) < 1e-9
    assert all(o[1] == out[0][1] and o[2] == out[0][2] and np.array_equal(o[3], out[0][3]) for o in out.values())     # replicated scalars agree bit for bit
    assert np.all(np.isfinite(out[0][3]))
