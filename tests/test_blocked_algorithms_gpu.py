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


def test_linalg_entry_points_random_soak_at_the_c_abi(gnp):
    """Opt-in soak (GPMP_LINALG_ABI_SOAK_CASES=<count>, GPMP_LINALG_ABI_SOAK_SEED) of the factorisation and solve entry points called as
    a C host would: matrices inside larger buffers (leading dimensions wider than n, odd ones too; 8-byte-aligned starts), guard
    values around every operand, dinv recomputed or passed, scratch passed or NULL, both solve directions, the right-hand solve,
    trtri / lauum, the log-determinant -- against LAPACK; and a matrix whose leading minor of a random order k is not positive
    definite must report exactly that k (LAPACK's dpotrf info), from the one-stream route and from the look-ahead route."""
    import ctypes
    import os

    import scipy.linalg as sla
    import torch

    from gpmp_amd import _lib

    ncases = int(os.environ.get("GPMP_LINALG_ABI_SOAK_CASES", "0"))
    if ncases <= 0:
        pytest.skip("opt-in: GPMP_LINALG_ABI_SOAK_CASES=<count>")
    lib = _lib.load()
    dev = gnp._dev()
    rng = np.random.default_rng(int(os.environ.get("GPMP_LINALG_ABI_SOAK_SEED", "5")))
    bad = []

    def place(a, fill=-7.0):
        r, c = a.shape
        ld, off = c + int(rng.choice([0, 1, 3, 8])), int(rng.choice([0, 1, 2]))
        buf = torch.full((off + r * ld + 4,), fill, dtype=torch.float64, device=dev)
        view = buf[off:off + r * ld].view(r, ld)
        view[:, :c] = torch.as_tensor(a, device=dev)
        return buf, view, ld, off

    def window(buf, view_shape, ld, off, cols):
        got = buf.cpu().numpy()
        r = view_shape[0]
        W = got[off:off + r * ld].reshape(r, ld)
        clean = bool(np.all(got[:off] == -7.0) and np.all(got[off + r * ld:] == -7.0) and np.all(W[:, cols:] == -7.0))
        return W[:, :cols], clean

    for i in range(ncases):
        kind = int(rng.integers(3))
        n = int(rng.integers(1, 300)) if kind == 0 else (int(rng.choice([128, 1024, 2048, 4096])) + int(rng.integers(-2, 3)) if kind == 1 else int(rng.integers(300, 5200)))
        m = int(rng.choice([1, 2, 3, 4, 5, 16, 100, 511, 512, 513, int(rng.integers(1, 1500))]))
        G = rng.standard_normal((n, max(n // 2, 1)))
        K = G @ G.T / max(n // 2, 1) + np.eye(n)                      # cond ~ 10
        B = rng.standard_normal((n, m))
        Lref = np.linalg.cholesky(K)
        errs, clean = {}, True
        Abuf, Av, lda, aoff = place(K)
        ndinv = max(int(lib.gpmp_dinv_elems(n)), 1)
        dinv = torch.empty(ndinv, dtype=torch.float64, device=dev)
        info = torch.full((1,), -1, dtype=torch.int32, device=dev)
        _lib.check(lib.gpmp_potrf_lower_async(Av.data_ptr(), n, lda, gnp._ptr(dinv), gnp._ptr(info), gnp._stream()), "potrf")
        Lw, c = window(Abuf, (n,), lda, aoff, n)
        clean &= c
        L = np.tril(Lw)
        errs["info"] = float(int(info.item()) != 0)
        errs["L"] = rel_err(L, Lref)
        for trans in (0, 1):
            Bbuf, Bv, ldb, boff = place(B)
            use_dinv, use_scratch = bool(rng.integers(2)), bool(rng.integers(2))
            scratch = torch.empty(ndinv, dtype=torch.float64, device=dev) if (use_scratch or not use_dinv) else None
            _lib.check(lib.gpmp_trsm_lower(Av.data_ptr(), n, lda, gnp._ptr(dinv) if use_dinv else None, Bv.data_ptr(), m, ldb, trans,
                                           None if scratch is None else gnp._ptr(scratch), gnp._stream()), "trsm")
            Xw, c = window(Bbuf, (n,), ldb, boff, m)
            clean &= c
            errs[f"trsm{trans}"] = rel_err(Xw, sla.solve_triangular(Lref, B, lower=True, trans=trans))
        # X L^T = R for a k x n right-hand side R (the panel solve of the blocked factorisation): X = R L^-T
        k = int(rng.integers(1, 400))
        R = rng.standard_normal((k, n))
        Rbuf, Rv, ldr, roff = place(R)
        _lib.check(lib.gpmp_trsm_right_lower(Av.data_ptr(), n, lda, gnp._ptr(dinv), Rv.data_ptr(), k, ldr, gnp._stream()), "trsm_right")
        Xw, c = window(Rbuf, (k,), ldr, roff, n)
        clean &= c
        errs["trsm_right"] = rel_err(Xw, sla.solve_triangular(Lref, R.T, lower=True).T)
        if n <= 3000:
            Tbuf, Tv, ldt, toff = place(np.full((n, n), np.nan))
            _lib.check(lib.gpmp_trtri_lower(Av.data_ptr(), n, lda, gnp._ptr(dinv), Tv.data_ptr(), ldt, gnp._stream()), "trtri")
            Tw, c = window(Tbuf, (n,), ldt, toff, n)
            clean &= c
            errs["trtri"] = rel_err(Tw @ Lref, np.eye(n)) + (float(np.max(np.abs(np.triu(Tw, 1)))) if n > 1 else 0.0)
            Ibuf, Iv, ldi, ioff = place(np.full((n, n), 9.0))
            _lib.check(lib.gpmp_lauum_lower(Tv.data_ptr(), n, ldt, Iv.data_ptr(), ldi, gnp._stream()), "lauum")
            Iw, c = window(Ibuf, (n,), ldi, ioff, n)
            clean &= c
            errs["lauum"] = rel_err(np.tril(Iw), np.tril(np.linalg.inv(K)))
        out = torch.zeros(1, dtype=torch.float64, device=dev)
        _lib.check(lib.gpmp_logdet_chol(Av.data_ptr(), n, lda, gnp._ptr(out), gnp._stream()), "logdet")
        errs["logdet"] = abs(float(out.item()) - np.linalg.slogdet(K)[1]) / max(1.0, abs(np.linalg.slogdet(K)[1]))
        # a leading minor of random order j + 1 that is not positive definite: info = j + 1, as LAPACK
        j = int(rng.integers(0, n))
        Kb = K.copy()
        Kb[j, j] -= Lref[j, j] ** 2 + 1.0                            # the pivot of step j becomes -1
        _, lapack_info = sla.lapack.dpotrf(Kb, lower=True)
        Bb, Bvw, ldbb, _ = place(Kb)
        info.fill_(-1)
        _lib.check(lib.gpmp_potrf_lower_async(Bvw.data_ptr(), n, ldbb, gnp._ptr(dinv), gnp._ptr(info), gnp._stream()), "potrf (not PD)")
        errs["info_not_pd"] = float(int(info.item()) != lapack_info or lapack_info != j + 1)
        errs["guards"] = float(not clean)
        tol = {"info": 0.0, "L": 1e-12, "trsm0": 1e-11, "trsm1": 1e-11, "trsm_right": 1e-11, "trtri": 1e-11, "lauum": 1e-10, "logdet": 1e-13,
               "info_not_pd": 0.0, "guards": 0.0}
        over = {k_: v for k_, v in errs.items() if not v <= tol[k_]}
        if over:
            bad.append((i, n, m, k, j, over))
        print(f"[abi soak {i:3d}] n={n} m={m} k={k} lda={lda} not-PD minor {j + 1}: " + " ".join(f"{k_} {v:.1e}" for k_, v in errs.items())
              + (" FAILED" if over else ""), flush=True)
    assert not bad, bad
