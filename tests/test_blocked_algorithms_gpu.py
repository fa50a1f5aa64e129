"""The look-ahead Cholesky, the factor-and-solve call and the inverse factor at sizes on both sides of their schedule thresholds
(one-stream route <= 2048 columns; 256- / 1024-column panels around 4096 remaining rows; look-ahead update in three pieces above
8192 rows; solve along the panels up to n = 8192) against LAPACK.  Rounds 1-4 ran these under ~45 environment switches at
non-default settings; round 5 removed the switches whose alternative had lost every A/B (DESIGN.md section 4 "Switches"), the
schedules are constants now and this file checks them at the sizes where each branch is taken.

Reference behaviour: numpy.linalg.cholesky / scipy solve_triangular behind gpmp/num/numpy_backend.py:136,140,465-469."""
import numpy as np
import pytest

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gnp():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd.num as gnp

    return gnp


def _spd(n):
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(n)
    x = rng.random((n, 4))
    return orc.maternp_covariance(x, None, 2, np.array([0.0, 1.2, 1.0, 0.8, 1.1])) + 1e-6 * np.eye(n), rng


@pytest.mark.parametrize("n", [2300, 5003, 9300])
def test_cholesky_and_factor_and_solve_vs_lapack(gnp, n):
    """n = 2300: look-ahead with 256-column panels only; 5003: 1024-column panels, then the 256-column tail, ragged last block;
    9300: the look-ahead update of the first panels in three pieces on two streams, factor-and-solve WITHOUT the solve along"""
    import scipy.linalg as sla

    K, rng = _spd(n)
    Lref = np.linalg.cholesky(K)
    B = rng.standard_normal((n, 600))
    L = np.tril(gnp.to_np(gnp.cholesky_factor(gnp.asarray(K)).L))
    assert rel_err(L, Lref) < 1e-10 and rel_err(L @ L.T, K) < 1e-14
    F, V = gnp.cholesky_factor_solve(gnp.asarray(K), gnp.asarray(B), overwrite=False)
    assert rel_err(np.tril(gnp.to_np(F.L)), Lref) < 1e-10
    assert rel_err(gnp.to_np(V), sla.solve_triangular(Lref, B, lower=True)) < 1e-9


@pytest.mark.parametrize("n", [2500, 4096, 5003])
def test_inverse_factor_vs_lapack(gnp, n):
    """T = L^-1 by doubling (full pairs: both products of the NN kind; the ragged last pair: the W^T form)"""
    K, _ = _spd(n)
    F = gnp.cholesky_factor(gnp.asarray(K))
    T = gnp.to_np(F.inverse_factor())
    L = np.tril(gnp.to_np(F.L))
    assert np.array_equal(np.triu(T, 1), np.zeros((n, n)))
    assert rel_err(T @ L, np.eye(n)) < 1e-9
