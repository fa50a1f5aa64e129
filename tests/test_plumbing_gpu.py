"""Round 4: the last backend-namespace names that ran on vendor libraries now run on the library's own kernels --
``gnp.qr`` (Householder reflectors on the fp64 GEMM + column dots; gpmp/core/linalg.py:49-110 call it with mode="complete")
and the matrix x matrix ``einsum("i..., i...")`` (gpmp/core/kriging.py:194) through ``gpmp_coldots_pair``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gnp():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd.num as gnp

    return gnp


@pytest.mark.parametrize("n,m", [(1, 1), (7, 3), (300, 257), (4097, 1000), (129, 5003)])
def test_coldots_pair_vs_numpy(gnp, n, m):
    rng = np.random.default_rng(n * 31 + m)
    a, b = rng.standard_normal((n, m)), rng.standard_normal((n, m))
    ref = np.einsum("ij,ij->j", a, b)
    out = gnp.to_np(gnp.einsum("i..., i...", gnp.asarray(a), gnp.asarray(b)))
    assert out.shape == (m,)
    scale = np.sqrt(n) * np.max(np.abs(a)) * np.max(np.abs(b))
    assert np.max(np.abs(out - ref)) < 1e-14 * max(scale, 1.0) * np.sqrt(n)
    # strided views (a column block of a wider matrix) take the same route without a copy of the parent
    wide = gnp.asarray(rng.standard_normal((n, m + 5)))
    v = wide[:, 2 : 2 + m]
    out2 = gnp.to_np(gnp.einsum("i..., i...", v, gnp.asarray(b)))
    np.testing.assert_allclose(out2, np.einsum("ij,ij->j", gnp.to_np(v), b), atol=1e-13 * max(scale, 1.0))


def test_generic_qr_returns_factors_for_rank_deficient_input_like_lapack(gnp):
    """(round 5, ADVICE) ``gnp.qr`` of a rank-deficient tall matrix returns factors with Q R = A, as the reference's LAPACK-backed
    qr does; the rank check lives in the kriging callers (compute_contrast_matrix raises); mode="r" returns R alone for any shape."""
    from gpmp_amd.core import linalg as L

    rng = np.random.default_rng(17)
    A = rng.random((40, 4))
    A[:, 2] = 0.0                                   # a zero column: geqrf proceeds with tau = 0
    Q, R = (gnp.to_np(t) for t in gnp.qr(gnp.asarray(A), mode="reduced"))
    np.testing.assert_allclose(Q @ R, A, atol=1e-13)
    np.testing.assert_allclose(Q.T @ Q, np.eye(4), atol=1e-13)
    B = rng.random((40, 3))
    B[:, 2] = B[:, 0] + B[:, 1]                     # linearly dependent columns
    Q, R = (gnp.to_np(t) for t in gnp.qr(gnp.asarray(B), mode="reduced"))
    np.testing.assert_allclose(Q @ R, B, atol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        L.compute_contrast_matrix(gnp.asarray(A))
    Rw = gnp.qr(gnp.asarray(rng.random((3, 5))), mode="r")
    Rt = gnp.qr(gnp.asarray(rng.random((30, 5))), mode="r")
    assert not isinstance(Rw, tuple) and not isinstance(Rt, tuple) and tuple(Rw.shape) == (3, 5) and tuple(Rt.shape) == (5, 5)


@pytest.mark.parametrize("n,q", [(50, 1), (300, 4), (1000, 9)])
def test_qr_on_library_kernels_vs_lapack(gnp, n, q):
    rng = np.random.default_rng(n + q)
    A = np.hstack((np.ones((n, 1)), rng.random((n, q - 1)))) if q > 1 else np.ones((n, 1))
    Qc, Rc = (gnp.to_np(t) for t in gnp.qr(gnp.asarray(A), mode="complete"))
    Qr, Rr = (gnp.to_np(t) for t in gnp.qr(gnp.asarray(A), mode="reduced"))
    assert Qc.shape == (n, n) and Rc.shape == (n, q) and Qr.shape == (n, q) and Rr.shape == (q, q)
    np.testing.assert_allclose(Qc @ Rc, A, atol=1e-13)
    np.testing.assert_allclose(Qr @ Rr, A, atol=1e-13)
    np.testing.assert_allclose(Qc.T @ Qc, np.eye(n), atol=1e-13)
    assert np.allclose(np.tril(Rr, -1), 0.0) and np.all(Rc[q:] == 0.0)
    # LAPACK's sign convention: the factors themselves agree, not only the subspaces
    Ql, Rl = np.linalg.qr(A, mode="complete")
    np.testing.assert_allclose(Rc, Rl, atol=1e-12)
    np.testing.assert_allclose(Qc[:, :q], Ql[:, :q], atol=1e-12)
    # the contrast projector W W^T = I - Q1 Q1^T is basis-independent
    from gpmp_amd.core import linalg as L

    W = gnp.to_np(L.compute_contrast_matrix(gnp.asarray(A)))
    assert W.shape == (n, n - q)
    np.testing.assert_allclose(W @ W.T, np.eye(n) - Ql[:, :q] @ Ql[:, :q].T, atol=1e-12)
    np.testing.assert_allclose(W.T @ A, 0.0, atol=1e-12)


def test_per_device_state_two_host_threads_and_release(gnp):
    """Round 4: the look-ahead factorisation's helper streams / events are kept per device ordinal behind a per-device mutex.
    Two host threads enqueue look-ahead factorisations (n > 2048) on their own streams of the one device at the same time;
    both factors must match LAPACK; gpmp_device_release returns the device's state and the next call rebuilds it."""
    import threading

    import torch

    from gpmp_amd import _lib
    from oracle import gp_oracle as orc

    lib = _lib.load()
    d = 4
    th = np.concatenate(([0.0], -np.log(0.4 * (1.0 + np.arange(d) / d))))
    sizes = (3000, 2700)
    mats = [orc.maternp_covariance(np.random.default_rng(s).random((s, d)), None, 2, th) for s in sizes]
    refs = [np.linalg.cholesky(K) for K in mats]
    out, errs = [None, None], []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    F = gnp.cholesky_factor(gnp.asarray(mats[i]))
                    out[i] = torch.tril(F.L).cpu().numpy()
            s.synchronize()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert np.max(np.abs(out[i] - refs[i])) < 1e-10 * np.max(np.abs(refs[i]))
    torch.cuda.synchronize()
    assert lib.gpmp_device_state_count() == 1
    assert lib.gpmp_device_release() == 0 and lib.gpmp_device_state_count() == 0
    assert lib.gpmp_device_release() == 0                      # nothing held: a no-op
    F = gnp.cholesky_factor(gnp.asarray(mats[0]))
    assert np.max(np.abs(torch.tril(F.L).cpu().numpy() - refs[0])) < 1e-10 * np.max(np.abs(refs[0]))
    assert lib.gpmp_device_state_count() == 1


@pytest.mark.parametrize("n,kind", [(1, "spd"), (2, "general"), (257, "general"), (300, "psd_rank_deficient"), (1100, "spd")])
def test_svd_by_one_sided_jacobi_vs_lapack(gnp, n, kind):
    """gnp.svd on the library's own kernel (gpmp_jacobi_sweep) against LAPACK: singular values, orthogonality, reconstruction;
    the positive SEMI-definite case is the one the path has (gpmp/core/sample_paths.py:54-58: repeated points)."""
    rng = np.random.default_rng(n)
    if kind == "general":
        A = rng.standard_normal((n, n))
    else:
        from oracle import gp_oracle as orc

        x = rng.random((n, 3))
        if kind == "psd_rank_deficient":
            x[n // 2:] = x[: n - n // 2]                      # every point twice: rank n / 2
        A = orc.maternp_covariance_it(x, x, 2, np.array([0.3, 1.0, 0.7, 1.2]))
    U, s, Vt = (gnp.to_np(t) for t in gnp.svd(gnp.asarray(A), full_matrices=True, hermitian=(kind != "general")))
    sref = np.linalg.svd(A, compute_uv=False)
    scale = sref[0]
    assert U.shape == (n, n) and s.shape == (n,) and Vt.shape == (n, n)
    assert np.all(np.diff(s) <= 0.0) and np.all(s >= 0.0)
    assert np.max(np.abs(s - sref)) < 1e-12 * scale * max(1.0, np.sqrt(n))
    assert np.max(np.abs(U.T @ U - np.eye(n))) < 5e-12 and np.max(np.abs(Vt @ Vt.T - np.eye(n))) < 5e-12    # ~ n x sweeps rotations of eps each
    assert np.max(np.abs((U * s) @ Vt - A)) < 1e-12 * scale
    if kind != "general":
        # the symmetric square root the sample paths use: C = U sqrt(s) Vt, C C^T = A (sample_paths.py:57)
        C = (U * np.sqrt(s)) @ Vt
        assert np.max(np.abs(C @ C.T - A)) < 1e-11 * scale


@pytest.mark.parametrize("kind", ["nonsymmetric", "symmetric_indefinite_pm_pairs", "nonsymmetric_rank_deficient", "antidiagonal"])
def test_svd_with_default_arguments_does_not_trust_the_hermitian_flag(gnp, kind):
    """(round 5, ADVICE) ``gnp.svd(A)`` with DEFAULT arguments (hermitian=True) on matrices that are not symmetric positive
    semi-definite: the reference backends ignore the flag and return a general SVD (gpmp/num/torch_backend.py:833-834), so
    U diag(s) Vt = A must hold -- also for a symmetric indefinite matrix with +-lambda pairs (two-dimensional singular subspaces)."""
    rng = np.random.default_rng(5)
    n = 64
    if kind == "nonsymmetric":
        A = rng.standard_normal((n, n))
    elif kind == "symmetric_indefinite_pm_pairs":
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.concatenate((np.repeat(np.linspace(1.0, 3.0, n // 4), 2) * np.tile([1.0, -1.0], n // 4), rng.standard_normal(n // 2)))
        A = (Q * lam) @ Q.T
        A = 0.5 * (A + A.T)
    elif kind == "nonsymmetric_rank_deficient":
        A = rng.standard_normal((n, n - 7)) @ rng.standard_normal((n - 7, n))
    else:
        A = np.array([[0.0, 1.0], [1.0, 0.0]])
        n = 2
    U, s, Vt = (gnp.to_np(t) for t in gnp.svd(gnp.asarray(A)))
    sref = np.linalg.svd(A, compute_uv=False)
    assert np.max(np.abs(s - sref)) < 1e-12 * sref[0] * n
    assert np.max(np.abs(U.T @ U - np.eye(n))) < 1e-11 and np.max(np.abs(Vt @ Vt.T - np.eye(n))) < 1e-10
    assert np.max(np.abs((U * s) @ Vt - A)) < 1e-12 * sref[0] * n


@pytest.mark.parametrize("m,width", [(40000, 80), (50000, 112), (46000, 96)])
def test_forward_solve_leaves_with_fitted_strip_widths(gnp, m, width):
    """Round 4: with many right-hand sides the fused 512-row leaves of the forward solve take 112- / 96- / 80-column strips when
    that fills the machine better than 128 (gemm_f64.hip: launch_trsm_leaf_forward); the solution must be LAPACK's on sampled
    columns (gpmp/num/numpy_backend.py:467) and the launch-per-block route's (no scratch area: no fused leaves) everywhere."""
    import scipy.linalg as sla
    import torch

    from gpmp_amd import _lib
    from oracle import gp_oracle as orc

    lib = _lib.load()
    n = 1536                                                    # three leaves of 512 rows + the updates between them
    rng = np.random.default_rng(m)
    x = rng.random((n, 3))
    K = orc.maternp_covariance(x, None, 2, np.array([0.0, 1.0, 0.8, 1.3])) + 1e-6 * np.eye(n)
    F = gnp.cholesky_factor(gnp.asarray(K))
    B = torch.randn((n, m), dtype=torch.float64, device=F.L.device, generator=torch.Generator(device=F.L.device).manual_seed(m))
    X = F.solve_lower(B.clone())
    X0 = gnp.as_matrix(B, copy=True)
    _lib.check(lib.gpmp_trsm_lower(gnp._ptr(F.L), n, gnp._ld(F.L), gnp._ptr(F.dinv), gnp._ptr(X0), m, gnp._ld(X0), 0, None, gnp._stream()),
               "gpmp_trsm_lower")
    assert float((X - X0).abs().max()) < 1e-11 * float(X0.abs().max())
    cols = np.concatenate((np.arange(0, 300), np.arange(m - 300, m), rng.choice(m, 400, replace=False)))
    ref = sla.solve_triangular(np.tril(gnp.to_np(F.L)), B[:, torch.as_tensor(cols, device=B.device)].cpu().numpy(), lower=True)
    assert np.max(np.abs(X[:, torch.as_tensor(cols, device=B.device)].cpu().numpy() - ref)) < 1e-10 * np.max(np.abs(ref))


@pytest.mark.parametrize("M,N,K", [(512, 50000, 512), (1024, 50000, 1024), (2048, 30002, 512), (640, 46000, 1024)])
def test_gemm_with_fitted_tile_width_vs_numpy(gnp, M, N, K):
    """Round 4: plain products on the LDS-direct kernel take 112- or 96-column tiles when that fills the last round of the machine
    better (gemm_f64.hip: launch_t).  Against NumPy on sampled columns, incl. a ragged last tile."""
    import torch

    rng = np.random.default_rng(M + N + K)
    A = gnp.asarray(rng.standard_normal((M, K)))
    B = gnp.asarray(rng.standard_normal((K, N)))
    C1 = gnp.matmul(A, B)
    cols = np.concatenate((np.arange(0, 260), np.arange(N - 260, N), rng.choice(N, 300, replace=False)))
    ref = gnp.to_np(A) @ gnp.to_np(B)[:, cols]
    assert np.max(np.abs(C1[:, torch.as_tensor(cols, device=C1.device)].cpu().numpy() - ref)) < 1e-12 * np.sqrt(K) * 10


def test_backend_functions_accept_strided_device_views(gnp):
    """The dense-algebra names of the backend (cholesky, cholesky_solve, solve_triangular, matmul, coldots, scaled_distance, logdet,
    inv, einsum) on every memory form a device tensor can take -- transposed storage, row- / column-strided views, a wider leading
    dimension, strided vectors, a column of a matrix -- in every combination of their operands: the kernels read row-major memory
    through raw pointers, so each form must be brought to that layout (or passed with its leading dimension) first."""
    import itertools

    import scipy.linalg as sla
    import torch

    dev = gnp._dev()
    rng = np.random.default_rng(3)
    n, m = 300, 70
    A = rng.standard_normal((n, n)); K = A @ A.T / n + np.eye(n)
    B = rng.standard_normal((n, m)); v = rng.standard_normal(n); x = rng.random((n, 3)); y = rng.random((m, 3))
    def views(a):
        t = torch.as_tensor(a, device=dev)
        out = {"plain": t}
        if a.ndim == 2:
            out["transposed storage"] = torch.as_tensor(np.ascontiguousarray(a.T), device=dev).T
            out["row-strided"] = torch.as_tensor(np.repeat(a, 2, axis=0), device=dev)[::2]
            out["wider ld"] = torch.as_tensor(np.hstack((a, a)), device=dev)[:, :a.shape[1]]
            out["col-strided"] = torch.as_tensor(np.repeat(a, 2, axis=1), device=dev)[:, ::2]
        else:
            out["strided"] = torch.as_tensor(np.repeat(a, 2), device=dev)[::2]
            out["column of a matrix"] = torch.as_tensor(np.stack((a, a), axis=1), device=dev)[:, 0]
        return out
    L = np.linalg.cholesky(K)
    checks = {
      "cholesky": (lambda K_: np.tril(gnp.to_np(gnp.cholesky(K_))), [K], L),
      "cholesky_solve": (lambda K_, B_: gnp.to_np(gnp.cholesky_solve(K_, B_)[0]) if isinstance(gnp.cholesky_solve(K_, B_), tuple) else gnp.to_np(gnp.cholesky_solve(K_, B_)), [K, B], np.linalg.solve(K, B)),
      "cholesky_solve vec": (lambda K_, v_: (lambda r: gnp.to_np(r[0] if isinstance(r, tuple) else r))(gnp.cholesky_solve(K_, v_)), [K, v], np.linalg.solve(K, v)),
      "solve_triangular": (lambda L_, B_: gnp.to_np(gnp.solve_triangular(L_, B_, lower=True)), [L, B], sla.solve_triangular(L, B, lower=True)),
      "solve_triangular vec": (lambda L_, v_: gnp.to_np(gnp.solve_triangular(L_, v_, lower=True)), [L, v], sla.solve_triangular(L, v, lower=True)),
      "solve_triangular upper": (lambda U_, B_: gnp.to_np(gnp.solve_triangular(U_, B_, lower=False)), [L.T.copy(), B], sla.solve_triangular(L.T, B, lower=False)),
      "matmul": (lambda A_, B_: gnp.to_np(gnp.matmul(A_, B_)), [A, B], A @ B),
      "matmul vec": (lambda A_, v_: gnp.to_np(gnp.matmul(A_, v_)), [A, v], A @ v),
      "coldots": (lambda B_, v_: gnp.to_np(gnp.coldots(B_, v_.reshape(-1, 1))), [B, v], np.vstack((v @ B, np.sum(B * B, axis=0)))),
      "scaled_distance": (lambda x_, y_: gnp.to_np(gnp.scaled_distance(np.zeros(3), x_, y_)), [x, y], np.sqrt(((x[:, None, :] - y[None, :, :]) ** 2).sum(-1))),
      "logdet": (lambda K_: np.array(gnp.logdet(K_)), [K], np.linalg.slogdet(K)[1]),
      "inv": (lambda K_: gnp.to_np(gnp.inv(K_)), [K], np.linalg.inv(K)),
      "einsum": (lambda B_, C_: gnp.to_np(gnp.einsum("i..., i...", B_, C_)), [B, B], np.einsum("i..., i...", B, B)),
    }
    bad = []
    for name, (fn, args, ref) in checks.items():
        vs = [views(a) for a in args]
        for combo in itertools.product(*[list(v.items()) for v in vs]):
            label = ", ".join(k for k, _ in combo)
            try:
                got = fn(*[t for _, t in combo])
                err = float(np.max(np.abs(np.asarray(got) - ref)) / max(1.0, np.max(np.abs(ref))))
                ok = err < 1e-9
            except Exception as e:  # noqa: BLE001
                ok, err = False, repr(e)[:200]
            if not ok:
                bad.append((name, label, err))
    assert not bad, bad


def test_repeated_calls_do_not_grow_device_memory_host_memory_or_descriptors(gnp):
    """400 rounds of every model-level entry point (predict with and without weights, NLL, REML, leave-one-out, criterion value and
    gradient) on the same inputs: allocated device bytes, the process's resident set and its open descriptors stay where they were
    after the warm-up (the library owns per-device streams / events / flag blocks; none may be created per call)."""
    import os

    import gpmp_amd as gp
    import torch

    from tests.helpers import make_xz, theta_aniso

    def lin(x, p):
        return gnp.hstack((gnp.ones((x.shape[0], 1)), gnp.asarray(x)))

    def rss_mb():
        return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6

    x, z = make_xz(600, 3, 1)
    xt, _ = make_xz(200, 3, 2)
    th = theta_aniso(3)
    mz = gp.Model(None, gp.kernel.MaternCovariance(2), None, th, "zero")
    ml = gp.Model(lin, gp.kernel.MaternCovariance(2), None, th, "linear_predictor")
    crit = gp.kernel.make_selection_criterion_with_gradient(ml, gp.kernel.negative_log_restricted_likelihood, x, z)

    def step():
        mz.predict(x, z, xt)
        ml.predict(x, z, xt, return_lambdas=True)
        mz.negative_log_likelihood_zero_mean(th, x, z)
        ml.negative_log_restricted_likelihood(th, x, z)
        ml.loo(x, z)
        crit[1](th)
        crit[3](th)

    for _ in range(40):
        step()
    torch.cuda.synchronize()
    a0, r0, f0 = torch.cuda.memory_allocated(), rss_mb(), len(os.listdir("/proc/self/fd"))
    for _ in range(400):
        step()
    torch.cuda.synchronize()
    a1, r1, f1 = torch.cuda.memory_allocated(), rss_mb(), len(os.listdir("/proc/self/fd"))
    assert a1 == a0 and f1 == f0 and r1 - r0 < 20.0, (a0, a1, r0, r1, f0, f1)
