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


def test_blocked_algorithms_random_soak(gnp):
    """Opt-in soak (GPMP_LINALG_SOAK_CASES=<count>, GPMP_LINALG_SOAK_SEED): random sizes -- tiny, one off every schedule threshold
    (128-column blocks, 1024 / 2048 / 4096 / 8192 rows), anything up to 9000 -- and random numbers of right-hand sides (1 ... 5: the
    one-launch sweep; 16 / 17; around 512: the fused leaves; up to 3000) through the factorisation, both triangular solves, the
    factor-and-solve call, K^-1 B, the inverse factor and the log-determinant, against LAPACK."""
    import os

    import scipy.linalg as sla

    ncases = int(os.environ.get("GPMP_LINALG_SOAK_CASES", "0"))
    if ncases <= 0:
        pytest.skip("opt-in: GPMP_LINALG_SOAK_CASES=<count>")
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(int(os.environ.get("GPMP_LINALG_SOAK_SEED", "5")))
    bad = []
    for i in range(ncases):
        kind = int(rng.integers(4))
        if kind == 0:
            n = int(rng.integers(1, 300))
        elif kind == 1:
            n = int(rng.choice([128, 256, 1024, 2048, 4096, 8192])) + int(rng.integers(-2, 3))
        elif kind == 2:
            n = 128 * int(rng.integers(1, 40)) + int(rng.integers(-1, 2))
        else:
            n = int(rng.integers(300, 9000))
        n = max(n, 1)
        m = int(rng.choice([1, 2, 3, 4, 5, 16, 17, 100, 511, 512, 513, int(rng.integers(1, 3000))]))
        x = rng.random((n, 4))
        K = orc.maternp_covariance(x, None, 2, np.array([0.0, 1.2, 1.0, 0.8, 1.1])) + 1e-6 * np.eye(n)
        B = rng.standard_normal((n, m))
        Lref = np.linalg.cholesky(K)
        Yref = sla.solve_triangular(Lref, B, lower=True)
        Xref = sla.solve_triangular(Lref, Yref, lower=True, trans="T")
        F = gnp.cholesky_factor(gnp.asarray(K))
        L = np.tril(gnp.to_np(F.L))
        Bd = gnp.asarray(B)
        errs = {"L": rel_err(L, Lref), "LLt": rel_err(L @ L.T, K),
                "fwd": rel_err(gnp.to_np(F.solve_lower(Bd)), Yref),
                "bwd": rel_err(gnp.to_np(F.solve_lower(gnp.asarray(Yref), trans=True)), Xref),
                "solve": rel_err(gnp.to_np(F.solve(Bd)), Xref),
                "vec": rel_err(gnp.to_np(F.solve_lower(Bd[:, 0].contiguous())), Yref[:, 0]),
                "logdet": abs(F.logdet() - 2 * np.sum(np.log(np.diag(Lref)))) / max(1.0, abs(2 * np.sum(np.log(np.diag(Lref)))))}
        F2, V = gnp.cholesky_factor_solve(gnp.asarray(K), Bd, overwrite=False)
        errs["factor_solve_L"], errs["factor_solve_V"] = rel_err(np.tril(gnp.to_np(F2.L)), Lref), rel_err(gnp.to_np(V), Yref)
        if n <= 5200:
            T = gnp.to_np(F.inverse_factor())
            errs["T_upper_zero"] = float(np.max(np.abs(np.triu(T, 1)))) if n > 1 else 0.0
            errs["TL"] = rel_err(T @ L, np.eye(n))
        # cond(K) ~ 1e6: entries of L to 1e-10, solutions to 1e-8 (the backward solve amplifies by cond(L) ~ 1e3 once more)
        tol = {"L": 1e-10, "LLt": 1e-14, "fwd": 1e-9, "bwd": 1e-8, "solve": 1e-7, "vec": 1e-9, "logdet": 1e-12, "factor_solve_L": 1e-10,
               "factor_solve_V": 1e-9, "T_upper_zero": 0.0, "TL": 1e-9}
        over = {k: v for k, v in errs.items() if not v <= tol[k]}
        if over:
            bad.append((i, n, m, over))
        print(f"[linalg soak {i:3d}] n={n} m={m}: " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()) + (" FAILED" if over else ""), flush=True)
    assert not bad, bad
