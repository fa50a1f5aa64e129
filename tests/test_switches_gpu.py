"""Every schedule switch of the look-ahead Cholesky and of the factor-and-solve call that the library reads at every call
("Switches"), at a non-default setting: the factor must still be the LAPACK factor and the solve the LAPACK solve.
The defaults are what the rest of the suite runs; this file keeps the measured-and-kept alternatives from rotting.

Reference behaviour: numpy.linalg.cholesky / scipy solve_triangular behind gpmp/num/numpy_backend.py:136,140,465-469."""
import numpy as np
import pytest

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu

SETTINGS = [
    {},
    {"GPMP_POTRF_TAIL_BLOCKED_BELOW": "2048"},
    {"GPMP_POTRF_W128_BELOW": "4096"},
    {"GPMP_POTRF_W256_BELOW": "0"},
    {"GPMP_POTRF_WIDE_ABOVE": "1000000"},
    {"GPMP_POTRF_WIDE_ABOVE": "2048", "GPMP_POTRF_LA_SPLIT_ABOVE": "1024"},
    {"GPMP_POTRF_LA_SPLIT": "0"},
    {"GPMP_POTRF_LEAN_ABOVE": "1024"},
    {"GPMP_POTRF_LEAN_ABOVE": "1000000"},
    {"GPMP_POTRF_MAIN_AFTER_LA_BELOW": "0"},
    {"GPMP_POTRF_MAIN_AFTER_LA_BELOW": "1000000"},
    {"GPMP_POTRF_BLOCKED_BELOW": "100000"},
    {"GPMP_GEMM_SMALL_ROWS16_BELOW": "0"},
    {"GPMP_GEMM_SMALL_ROWS16_BELOW": "1000000"},
    {"GPMP_GEMM_TRI_BLOCK": "0"},
    {"GPMP_GEMM_EARLY_ISSUE": "1"},                   # (round 4) operand tiles requested one barrier earlier, at every K
    {"GPMP_GEMM_EARLY_ISSUE": "-1"},                     # (round 4) lower-triangular tile sets row by row (the order of rounds 1-3)
    # (round 4) two-level 2048-column panels: binary blocking inside the panel, rank-2048 trailing updates, four look-ahead pieces
    {"GPMP_POTRF_SUPER_ABOVE": "2048", "GPMP_POTRF_WIDE_ABOVE": "2048", "GPMP_POTRF_LA_SPLIT_ABOVE": "1024"},
    {"GPMP_POTRF_SUPER_ABOVE": "2048", "GPMP_POTRF_WIDE_ABOVE": "2048", "GPMP_POTRF_LA_SPLIT": "0"},
]
SOLVE_SETTINGS = [
    {},
    {"GPMP_POTRF_ALONG_LEAN": "1"},
    {"GPMP_POTRF_ALONG_ROWS": "512"},
    {"GPMP_POTRF_ALONG_ROWS": "1536"},
    {"GPMP_POTRF_SOLVE_ALONG_ABOVE": "100000"},
    {"GPMP_POTRF_SOLVE_ALONG_BELOW": "0", "GPMP_POTRF_SOLVE_OVERLAP": "1"},
    {"GPMP_TRSM_LEAF_MIN_STRIPS": "0"},
    {"GPMP_TRSM_LEAF_MIN_STRIPS": "1000000"},
    {"GPMP_TRSM_FUSED_LEAF": "1", "GPMP_TRSM_LEAF_NARROW_BELOW": "0"},
    {"GPMP_POTRF_ALONG_RIGHT": "0"},                  # (round 3) left-looking updates in the panel-by-panel solve
    {"GPMP_POTRF_ALONG_RIGHT": "0", "GPMP_POTRF_ALONG_ROWS": "512"},
    {"GPMP_POTRF_SUPER_ABOVE": "2048", "GPMP_POTRF_WIDE_ABOVE": "2048"},      # (round 4) 2048-column panels under the panel-by-panel solve
    {"GPMP_GEMM_EARLY_ISSUE": "1"},
    {"GPMP_GEMM_EARLY_ISSUE": "-1"},
]


@pytest.fixture(scope="module")
def problem():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd.num as gnp
    from oracle import gp_oracle as orc

    out = {}
    for n in (2300, 5003):
        rng = np.random.default_rng(n)
        x = rng.random((n, 4))
        K = orc.maternp_covariance(x, None, 2, np.array([0.0, 1.2, 1.0, 0.8, 1.1])) + 1e-6 * np.eye(n)
        B = rng.standard_normal((n, 600))
        out[n] = (K, np.linalg.cholesky(K), B)
    return gnp, out


def _ident(s):
    return ",".join(f"{k[5:]}={v}" for k, v in s.items()) or "defaults"


@pytest.mark.parametrize("setting", SETTINGS, ids=_ident)
def test_cholesky_under_every_schedule_switch(problem, setting, monkeypatch):
    gnp, probs = problem
    for k, v in setting.items():
        monkeypatch.setenv(k, v)
    for n, (K, Lref, _) in probs.items():
        L = np.tril(gnp.to_np(gnp.cholesky_factor(gnp.asarray(K)).L))
        assert rel_err(L, Lref) < 1e-10, (n, setting)
        assert rel_err(L @ L.T, K) < 1e-14, (n, setting)


@pytest.mark.parametrize("setting", SOLVE_SETTINGS, ids=_ident)
def test_factor_and_solve_under_every_schedule_switch(problem, setting, monkeypatch):
    import scipy.linalg as sla

    gnp, probs = problem
    for k, v in setting.items():
        monkeypatch.setenv(k, v)
    for n, (K, Lref, B) in probs.items():
        F, V = gnp.cholesky_factor_solve(gnp.asarray(K), gnp.asarray(B), overwrite=False)
        L = np.tril(gnp.to_np(F.L))
        assert rel_err(L, Lref) < 1e-10, (n, setting)
        ref = sla.solve_triangular(Lref, B, lower=True)
        assert rel_err(gnp.to_np(V), ref) < 1e-9, (n, setting)


@pytest.mark.parametrize("n", [2500, 4096, 5003])
def test_inverse_factor_both_forms(problem, n, monkeypatch):
    """GPMP_TRTRI_NN (round 3): W = L21 T11 kept untransposed, both products of the NN kind (default) against the W^T form --
    the same k order in every accumulation, so the two inverse factors are IDENTICAL, and both are the LAPACK inverse"""
    import torch
    gnp, _ = problem
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(n)
    x = rng.random((n, 4))
    K = orc.maternp_covariance(x, None, 2, np.array([0.0, 1.2, 1.0, 0.8, 1.1])) + 1e-6 * np.eye(n)
    F = gnp.cholesky_factor(gnp.asarray(K))
    monkeypatch.setenv("GPMP_TRTRI_NN", "1")
    T1 = F.inverse_factor().clone()
    monkeypatch.setenv("GPMP_TRTRI_NN", "0")
    T0 = F.inverse_factor().clone()
    assert torch.equal(T0, T1)
    L = np.tril(gnp.to_np(F.L))
    assert rel_err(gnp.to_np(T1) @ L, np.eye(n)) < 1e-9

