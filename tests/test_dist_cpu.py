"""Multi-process (gloo, CPU) tests of the N > 1 paths: xt-sharded prediction and the 2-D block-cyclic
Cholesky / NLL schedule.  The local arithmetic is a torch-CPU stand-in (tests/cpu_local_ops.py); what is
tested is the distributed host logic that the GPU path shares verbatim."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gp_oracle as orc
from tests.helpers import make_xz, theta_aniso


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)          # up to 8 ranks on the CPUs of the test box: one BLAS thread each, not 8 x 8
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cov(x, y, t, pairwise=False):
    return orc.maternp_covariance_it(np.asarray(x), np.asarray(y), 2, t, pairwise)


def _chol_worker(rank, world, port, pr, pc, n, nb, out, transport="bcast", lookahead=True):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        x, z = make_xz(n, 3, 7)
        th = theta_aniso(3, scale=0.4)
        nugget = 1e-6
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=CpuLocalOps(), transport=transport, lookahead=lookahead)
        ch.build_local_gram(_cov, x, th, nugget)
        info = ch.factor()
        nll = ch.negative_log_likelihood(z)
        # reassemble the factor on rank 0 for comparison
        ri, ci = ch.global_row_index(), ch.global_col_index()
        payload = (ri, ci, ch.A.numpy().copy(), info, nll, ch.bytes_received)
        gathered = [None] * world
        dist.all_gather_object(gathered, payload)
        if rank == 0:
            L = np.zeros((n, n))
            for (r_i, c_i, a, _, _, _) in gathered:
                if len(r_i) and len(c_i):
                    L[np.ix_(r_i, c_i)] = a
            np.save(out, np.tril(L))
            np.save(out + ".meta.npy", np.array([info, nll, sum(g[5] for g in gathered)], dtype=np.float64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb,transport,lookahead", [
    (1, 2, 700, 128, "bcast", True), (2, 1, 700, 128, "bcast", True), (2, 2, 1000, 128, "bcast", True),
    (2, 2, 1100, 256, "p2p", True), (1, 4, 900, 128, "p2p", True), (2, 2, 1000, 128, "bcast", False),
    (2, 2, 200, 128, "p2p", False),      # fewer blocks than the grid is wide: some ranks own nothing in a column
    (3, 2, 900, 128, "bcast", True), (2, 3, 900, 128, "p2p", True),     # non-square, Pr and Pc coprime: lcm(Pr, Pc) strides in the exchange
    (2, 4, 1300, 128, "bcast", True),    # the grid of BASELINE config 5 (8 ranks) ...
    (2, 4, 1300, 128, "p2p", True)])     # ... with the root fanning out over separate links
def test_block_cyclic_cholesky_and_nll(tmp_path, pr, pc, n, nb, transport, lookahead):
    world = pr * pc
    out = str(tmp_path / "L.npy")
    mp.spawn(_chol_worker, args=(world, _free_port(), pr, pc, n, nb, out, transport, lookahead), nprocs=world, join=True)
    L = np.load(out)
    info, nll, recv = np.load(out + ".meta.npy")
    x, z = make_xz(n, 3, 7)
    th = theta_aniso(3, scale=0.4)
    K = orc.maternp_covariance_it(x, x, 2, th) + 1e-6 * np.eye(n)
    Lref = np.linalg.cholesky(K)
    assert info == 0
    assert np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)) < 1e-9
    w = np.linalg.solve(Lref, z)
    ref_nll = 0.5 * (n * math.log(2 * math.pi) + 2 * np.sum(np.log(np.diag(Lref))) + w @ w)
    assert abs(nll - ref_nll) < 1e-9 * abs(ref_nll)
    assert recv > 0   # panels really travelled


def _nonpd_worker(rank, world, port, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        n = 512
        grid = ProcessGrid(1, 2)
        ch = BlockCyclicCholesky(grid, n, nb=128, ops=CpuLocalOps())
        A = np.eye(n)
        A[300, 300] = -1.0
        ch.set_local(torch.as_tensor(A[np.ix_(ch.global_row_index(), ch.global_col_index())].copy()))
        info = ch.factor()
        nll = ch.negative_log_likelihood(np.ones(n))
        if rank == 0:
            np.save(out, np.array([info, nll]))
    finally:
        dist.destroy_process_group()


def _profile_worker(rank, world, port, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        x, z = make_xz(400, 3, 7)
        ch = BlockCyclicCholesky(ProcessGrid(1, 2), 400, nb=128, ops=CpuLocalOps(), profile=True, reserve_cus=8)
        ch.build_local_gram(_cov, x, theta_aniso(3, scale=0.4), 1e-6)
        info = ch.factor()
        if rank == 0:
            np.save(out, np.array([info, len(ch.phase_times())]))
    finally:
        dist.destroy_process_group()


def test_profile_and_reserve_options_are_inert_without_a_gpu(tmp_path):
    """profile=True / reserve_cus only act on HIP streams; with CPU local ops they must neither fail nor report phases"""
    out = str(tmp_path / "p.npy")
    mp.spawn(_profile_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    info, nphases = np.load(out)
    assert info == 0 and nphases == 0


def test_block_cyclic_not_positive_definite(tmp_path):
    out = str(tmp_path / "r.npy")
    mp.spawn(_nonpd_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    info, nll = np.load(out)
    assert info == 301 and math.isinf(nll)      # LAPACK-style 1-based leading minor; criterion -> +inf


def _cov_full(x, y, t, pairwise=False):
    """covariance callable with the reference's signature (y None -> the tt / pairwise path)"""
    return orc.maternp_covariance(np.asarray(x), None if y is None else np.asarray(y), 2, t, pairwise)


def _dist_predict_worker(rank, world, port, pr, pc, n, m, nb, out, transport):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        x, z = make_xz(n, 3, 7)
        xt, _ = make_xz(m, 3, 8)
        th = theta_aniso(3, scale=0.4)
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=CpuLocalOps(), transport=transport)
        ch.build_local_gram(_cov, x, th, 1e-6)
        assert ch.factor() == 0
        mean, var, (j0, j1) = ch.predict_zero_mean(_cov_full, x, z, xt, th)
        # (round 4) the kriging weights from the backward solve on the same factor; mean / variance must not change
        mean2, var2, (a2, b2), lam_loc = ch.predict(_cov_full, x, z, xt, th, return_lambdas=True)
        assert (a2, b2) == (j0, j1) and np.array_equal(mean2, mean) and np.array_equal(var2, var)
        gathered = [None] * world
        dist.all_gather_object(gathered, (grid.r, grid.c, j0, j1, mean, var, ch.global_row_index(), lam_loc.numpy()))
        if rank == 0:
            zpm, zpv, lam = np.full(m, np.nan), np.full(m, np.nan), np.full((n, m), np.nan)
            for (r, c, a, b, mu, v, rows, blk) in gathered:
                if r == 0:
                    zpm[a:b], zpv[a:b] = mu, v
                else:       # every rank of a process column holds the same shard
                    assert np.array_equal(zpm[a:b], mu) or np.allclose(zpm[a:b], mu, rtol=0, atol=1e-13)
                if b > a and len(rows):
                    lam[np.ix_(rows, np.arange(a, b))] = blk
            assert not np.isnan(lam).any()             # every (row block, point shard) arrived exactly once
            np.save(out, np.stack([zpm, zpv]))
            np.save(out + ".lam.npy", lam)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,m,nb,transport", [(1, 2, 500, 77, 128, "bcast"), (2, 1, 500, 77, 128, "bcast"), (2, 2, 1000, 301, 128, "bcast"),
                                                    (2, 2, 700, 130, 256, "p2p"), (2, 4, 1300, 203, 128, "bcast"), (2, 4, 1300, 3, 128, "p2p")])
def test_block_cyclic_many_rhs_solve_and_predict(tmp_path, pr, pc, n, m, nb, transport):
    """predict mean + variance from the 2-D block-cyclic factor (many-right-hand-side forward solve with the prediction
    points split over the process columns) against the oracle's two-solve route; m = 3 on 4 process columns leaves one
    column without any point"""
    world = pr * pc
    out = str(tmp_path / "p.npy")
    mp.spawn(_dist_predict_worker, args=(world, _free_port(), pr, pc, n, m, nb, out, transport), nprocs=world, join=True)
    got = np.load(out)
    x, z = make_xz(n, 3, 7)
    xt, _ = make_xz(m, 3, 8)
    th = theta_aniso(3, scale=0.4)
    K = orc.maternp_covariance_it(x, x, 2, th) + 1e-6 * np.eye(n)
    Kit = orc.maternp_covariance_it(x, xt, 2, th)
    lam = np.linalg.solve(K, Kit)
    ref_mean = lam.T @ z
    ref_var = orc.maternp_covariance(xt, None, 2, th, True) - np.einsum("ij,ij->j", lam, Kit)
    assert np.max(np.abs(got[0] - ref_mean)) < 1e-8 * np.max(np.abs(z))
    assert np.max(np.abs(got[1] - ref_var)) < 1e-8
    assert np.max(np.abs(np.load(out + ".lam.npy") - lam)) < 1e-7 * np.max(np.abs(lam))


class _OracleBackedModel:
    """predict() with the reference semantics, computed by the CPU oracle (stand-in for gpmp_amd.Model)."""

    def __init__(self, th):
        self.m = orc.OracleModel(None, lambda x, y, t, pairwise=False: orc.maternp_covariance(x, y, 2, t, pairwise), None, th, "zero")

    def predict(self, xi, zi, xt):
        if xt.shape[0] == 0:
            return np.zeros(0), np.zeros(0)
        return orc.predict(self.m, xi, zi, xt)


def _predict_worker(rank, world, port, m, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import sharded_predict, shard_bounds

        xi, zi = make_xz(200, 3, 1)
        xt, _ = make_xz(m, 3, 2)
        zpm, zpv, (lo, hi) = sharded_predict(_OracleBackedModel(theta_aniso(3)), xi, zi, xt)
        assert (lo, hi) == shard_bounds(m, world, rank)
        if rank == world - 1:
            np.save(out, np.stack([zpm, zpv]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,m", [(2, 101), (3, 7), (2, 1)])
def test_sharded_predict_matches_single_process(tmp_path, world, m):
    out = str(tmp_path / "p.npy")
    mp.spawn(_predict_worker, args=(world, _free_port(), m, out), nprocs=world, join=True)
    got = np.load(out)
    xi, zi = make_xz(200, 3, 1)
    xt, _ = make_xz(m, 3, 2)
    ref = _OracleBackedModel(theta_aniso(3)).predict(xi, zi, xt)
    np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[1], ref[1], rtol=0, atol=1e-12)


def test_shard_bounds_cover_and_balance():
    from gpmp_amd.dist import shard_bounds

    for m in (0, 1, 7, 50000, 50001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(m, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == m
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_grid_shapes():
    from gpmp_amd.dist import ProcessGrid

    assert ProcessGrid.default_shape(8) == (2, 4)
    assert ProcessGrid.default_shape(4) == (2, 2)
    assert ProcessGrid.default_shape(2) == (1, 2)
    assert ProcessGrid.default_shape(1) == (1, 1)


def _oplog_worker(rank, world, port, pr, pc, n, m, nb, out, transport, lookahead):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        x, z = make_xz(n, 3, 7)
        xt, _ = make_xz(m, 3, 8)
        th = theta_aniso(3, scale=0.4)
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=CpuLocalOps(), transport=transport, lookahead=lookahead)
        ch.oplog = []
        ch.build_local_gram(_cov, x, th, 1e-6)
        assert ch.factor() == 0
        ch.negative_log_likelihood(z)
        ch.predict_zero_mean(_cov_full, x, z, xt, th)
        ch.predict(_cov_full, x, z, xt, th, return_lambdas=True)     # (round 4) backward solve: row broadcasts + one reduce per block column
        ch.value_and_grad(x, z, th, 2, P=np.ones((n, 1)))           # REML gradient: ring shifts inside the process rows
        gathered = [None] * world
        dist.all_gather_object(gathered, (grid.r, grid.c, ch.oplog))
        if rank == 0:
            import pickle

            with open(out, "wb") as f:
                pickle.dump(gathered, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["bcast", "p2p"])
@pytest.mark.parametrize("lookahead", [True, False])
def test_issue_order_is_identical_on_all_members_of_every_communicator(tmp_path, transport, lookahead):
    """What RCCL needs when several communicators are live on one device: every member of a communicator issues the SAME
    sequence of operations on it (same kind, root, size, step).  The schedule logs (communicator, operation, root, doubles,
    step) in host issue order on every rank of the 2 x 4 grid of BASELINE config 5 -- factorisation, NLL and the
    many-right-hand-side solve, look-ahead on and off, both transports -- and the per-communicator sequences are compared
    across the members.  Also: the four kinds of communicator are all exercised, a p2p "broadcast" is one message per
    send/recv group, and ranks that own nothing in a step still take part in the collectives of their communicators."""
    import pickle

    pr, pc, n, m, nb = 2, 4, 1300, 131, 128
    out = str(tmp_path / "log.pkl")
    mp.spawn(_oplog_worker, args=(pr * pc, _free_port(), pr, pc, n, m, nb, out, transport, lookahead), nprocs=pr * pc, join=True)
    with open(out, "rb") as f:
        gathered = pickle.load(f)
    per_comm = {}
    for (r, c, log) in gathered:
        assert log, "every rank communicates"
        for tag in {e[0] for e in log}:
            per_comm.setdefault(tag, []).append(((r, c), [e[:5] for e in log if e[0] == tag]))
    expect_members = {**{f"row{r}": pc for r in range(pr)}, **{f"col{c}": pr for c in range(pc)},
                      **{f"diag{c}": pr for c in range(pc)}, "world": pr * pc}
    assert set(per_comm) == set(expect_members)
    for tag, seqs in per_comm.items():
        assert len(seqs) == expect_members[tag], (tag, [who for who, _ in seqs])      # every member issued on it
        first = seqs[0][1]
        assert len(first) > 0
        for who, seq in seqs[1:]:
            assert seq == first, (tag, who, next((a, b) for a, b in zip(seq, first) if a != b) if len(seq) == len(first) else (len(seq), len(first)))
    kinds = {e[1] for _, _, log in gathered for e in log}
    assert ("p2p_bcast" in kinds) == (transport == "p2p") and ("broadcast" in kinds)     # (the NLL's vector solve always broadcasts)
    # the roots rotate as the block-cyclic layout says: row communicator r sees every process column as a root
    roots = {e[2] for e in per_comm["row0"][0][1] if e[1] in ("broadcast", "p2p_bcast")}
    assert roots == set(range(pc))
    # the backward solve reduces once per block column inside every process column, to the process row that owns the block row
    red = [e for e in per_comm["col0"][0][1] if e[1] == "reduce:bsolve"]
    assert len(red) == (n + nb - 1) // nb and {e[2] for e in red} == {0 * pc + 0, 1 * pc + 0}
    # the gradient's ring inside the process rows: Pc / 2 shifts, logged identically by every member
    assert [e[1] for e in per_comm["row0"][0][1] if e[1].startswith("ring_shift")] == [f"ring_shift{s}" for s in range(1, pc // 2 + 1)]


def _reml_loo_worker(rank, world, port, pr, pc, n, nb, q, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        x, z = make_xz(n, 3, 7)
        th = theta_aniso(3, scale=0.4)
        P = None if q == 0 else np.hstack((np.ones((n, 1)), x))[:, :q]
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=CpuLocalOps())
        ch.build_local_gram(_cov, x, th, 1e-6)
        assert ch.factor() == 0
        reml = ch.negative_log_restricted_likelihood(z, P) if q else ch.negative_log_likelihood(z)
        zloo, s2, eloo, idx = ch.loo(z, P)
        gathered = [None] * world
        dist.all_gather_object(gathered, (grid.r, idx, zloo, s2, eloo, reml))
        if rank == 0:
            full = np.full((3, n), np.nan)
            for (r, ix, zl, s, e, v) in gathered:
                assert v == reml                                  # the criterion is replicated
                if r == 0:
                    full[0, ix], full[1, ix], full[2, ix] = zl, s, e
            np.save(out, np.vstack((full, np.full((1, n), reml))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb,q", [(2, 2, 700, 128, 0), (2, 2, 700, 128, 1), (1, 2, 500, 128, 3), (2, 1, 500, 128, 1), (2, 4, 1100, 128, 2)])
def test_block_cyclic_reml_and_leave_one_out(tmp_path, pr, pc, n, nb, q):
    """REML (likelihood.py:92-129) and leave-one-out (loo.py:65-83,103-130) from the 2-D block-cyclic factor against the
    oracle's complete-QR / explicit-inverse routes; q = 0 is the zero-mean pair (NLL, zero-mean LOO)"""
    world = pr * pc
    out = str(tmp_path / "r.npy")
    mp.spawn(_reml_loo_worker, args=(world, _free_port(), pr, pc, n, nb, q, out), nprocs=world, join=True)
    got = np.load(out)
    x, z = make_xz(n, 3, 7)
    th = theta_aniso(3, scale=0.4)
    # K(x, x) + 1e-6 I on the ii path (identity dispatch as gpmp/kernel/matern.py:124-141: `y is x or y is None`)
    cov = lambda a, b, t, pairwise=False: (orc.maternp_covariance_it(a, a if b is None else b, 2, t, pairwise)            # noqa: E731
                                           + (1e-6 * np.eye(len(a)) if ((b is None or b is a) and not pairwise) else 0.0))
    if q == 0:
        om = orc.OracleModel(None, cov, None, th, "zero")
        ref_v = float(orc.negative_log_likelihood_zero_mean(om, th, x, z))
    else:
        mean = lambda a, p: np.hstack((np.ones((len(a), 1)), a))[:, :q]      # noqa: E731
        om = orc.OracleModel(mean, cov, None, th, "linear_predictor")
        ref_v = float(orc.negative_log_restricted_likelihood(om, th, x, z))
    rz, rs, re_ = orc.loo(om, x, z)
    assert abs(got[3, 0] - ref_v) < 1e-9 * abs(ref_v)
    assert np.max(np.abs(got[0] - rz)) < 1e-7 * np.max(np.abs(z))
    assert np.max(np.abs(got[1] - rs) / rs) < 1e-7
    assert np.max(np.abs(got[2] - re_)) < 1e-7 * np.max(np.abs(z))


def _grad_worker(rank, world, port, pr, pc, n, nb, q, noise, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        x, z = make_xz(n, 3, 7)
        th = theta_aniso(3, scale=0.4)
        if noise:
            th = np.concatenate(([th[0], math.log(1e-3)], th[1:]))
        P = None if q == 0 else np.hstack((np.ones((n, 1)), x))[:, :q]
        diag = math.exp(th[1]) if noise else 10.0 * math.exp(th[0]) * np.finfo(float).eps
        cov_it = (lambda a, b, t, pairwise=False: orc.maternp_covariance_it(np.asarray(a), np.asarray(b), 2, np.concatenate(([t[0]], t[2:])) if noise else t, pairwise))
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=CpuLocalOps())
        ch.build_local_gram(cov_it, x, th, diag)
        assert ch.factor() == 0
        v, gvec = ch.value_and_grad(x, z, th, 2, noise=noise, P=P)
        gathered = [None] * world
        dist.all_gather_object(gathered, (v, gvec))
        if rank == 0:
            for (v2, g2) in gathered:
                assert v2 == v and np.array_equal(g2, gvec)          # replicated
            np.save(out, np.concatenate(([v], gvec)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb,q,noise", [(2, 2, 600, 128, 0, False), (2, 2, 600, 128, 1, False), (1, 3, 500, 128, 2, True),
                                                (2, 1, 400, 128, 1, True), (2, 4, 1100, 128, 0, False), (2, 4, 1100, 128, 3, False)])
def test_block_cyclic_value_and_gradient(tmp_path, pr, pc, n, nb, q, noise):
    """ML (q = 0) and REML value + analytic gradient from the block-cyclic factor (T = L^-1 by the many-RHS solve, the blocks
    of T^T T formed around the process row and traced against dK block by block) against the oracle's analytic gradient
    (itself pinned to the reference's autograd: ref_gradients.npz); odd and even process-column counts, with / without a noise
    parameter"""
    world = pr * pc
    out = str(tmp_path / "g.npy")
    mp.spawn(_grad_worker, args=(world, _free_port(), pr, pc, n, nb, q, noise, out), nprocs=world, join=True)
    got = np.load(out)
    x, z = make_xz(n, 3, 7)
    th = theta_aniso(3, scale=0.4)
    if noise:
        th = np.concatenate(([th[0], math.log(1e-3)], th[1:]))
    ni = 1 if noise else None
    if q == 0:
        v, g = orc.nll_zero_mean_value_and_grad(x, z, 2, th, noise_index=ni)
    else:
        v, g = orc.reml_value_and_grad(x, z, np.hstack((np.ones((n, 1)), x))[:, :q], 2, th, noise_index=ni)
    assert abs(got[0] - v) < 1e-8 * abs(v)
    assert np.max(np.abs(got[1:] - g)) < 1e-6 * np.max(np.abs(g)), (got[1:], g)


def _uk_worker(rank, world, port, pr, pc, n, m, nb, q, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        x, z = make_xz(n, 3, 7)
        xt, _ = make_xz(m, 3, 8)
        th = theta_aniso(3, scale=0.4)
        mean = lambda a: np.hstack((np.ones((len(a), 1)), a))[:, :q]        # noqa: E731
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=CpuLocalOps())
        ch.build_local_gram(_cov, x, th, 1e-6)
        assert ch.factor() == 0
        mu, var, (j0, j1) = ch.predict(_cov_full, x, z, xt, th, P=mean(x), Pt=mean(xt))
        gathered = [None] * world
        dist.all_gather_object(gathered, (grid.r, j0, j1, mu, var))
        if rank == 0:
            zpm, zpv = np.full(m, np.nan), np.full(m, np.nan)
            for (r, a, b, mm, vv) in gathered:
                if r == 0:
                    zpm[a:b], zpv[a:b] = mm, vv
            np.save(out, np.stack([zpm, zpv]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,m,nb,q", [(2, 2, 700, 211, 128, 1), (1, 2, 500, 77, 128, 4), (2, 4, 1100, 301, 128, 2)])
def test_block_cyclic_universal_kriging(tmp_path, pr, pc, n, m, nb, q):
    """prediction with a linear predictor (constant / linear mean) from the block-cyclic factor -- the Schur-complement route
    on distributed pieces -- against the oracle's block-system solve (gpmp/core/kriging.py:70-116)"""
    world = pr * pc
    out = str(tmp_path / "p.npy")
    mp.spawn(_uk_worker, args=(world, _free_port(), pr, pc, n, m, nb, q, out), nprocs=world, join=True)
    got = np.load(out)
    x, z = make_xz(n, 3, 7)
    xt, _ = make_xz(m, 3, 8)
    th = theta_aniso(3, scale=0.4)
    cov = lambda a, b, t, pairwise=False: (orc.maternp_covariance_it(a, a if b is None else b, 2, t, pairwise)            # noqa: E731
                                           + (1e-6 * np.eye(len(a)) if ((b is None or b is a) and not pairwise) else 0.0))
    om = orc.OracleModel(lambda a, p: np.hstack((np.ones((len(a), 1)), a))[:, :q], cov, None, th, "linear_predictor")
    rm, rv = orc.predict(om, x, z, xt, zero_neg_variances=False)
    assert np.max(np.abs(got[0] - rm)) < 1e-8 * np.max(np.abs(z))
    assert np.max(np.abs(got[1] - rv)) < 1e-8


def _fit_problem(n):
    """noisy observations of a smooth function, noisy-Matern model: an interior, well-conditioned optimum"""
    x, z = make_xz(n, 2, 5)
    z = z + 0.1 * np.random.default_rng(9).standard_normal(n)
    return x, z, np.array([0.3, math.log(0.05), 0.2, -0.1])


def _cov_noisy_it(a, b, t, pairwise=False):
    """cross-covariance of the noisy model: theta = [log s2, log s2_noise, log 1/rho...] (the noise sits on the diagonal only)"""
    return orc.maternp_covariance_it(np.asarray(a), np.asarray(b), 2, np.concatenate(([t[0]], t[2:])), pairwise)


def _fit_worker(rank, world, port, pr, pc, n, nb, q, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import ProcessGrid, fit_covparam
        from tests.cpu_local_ops import CpuLocalOps

        x, z, th0 = _fit_problem(n)
        P = None if q == 0 else np.ones((n, 1))
        th, info = fit_covparam(ProcessGrid(pr, pc), _cov_noisy_it, x, z, th0, P=P, nb=nb, ops=CpuLocalOps(), p=2, noise=True,
                                options={"maxiter": 15})
        gathered = [None] * world
        dist.all_gather_object(gathered, (th, info["fun"], info["nfev"]))
        if rank == 0:
            for (t2, f2, e2) in gathered:
                assert np.array_equal(t2, th) and f2 == info["fun"] and e2 == info["nfev"]      # lockstep without a broadcast
            np.save(out, np.concatenate((th, [info["fun"], info["nfev"]])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,q", [(1, 2, 1), (2, 2, 0)])
def test_distributed_parameter_fit_follows_the_single_process_optimiser(tmp_path, pr, pc, q):
    """fit_covparam (SciPy L-BFGS-B over the distributed REML / ML value + gradient) against the same optimiser on the oracle's
    analytic criterion in one process: same starting point, same iteration cap -> the same selected parameters, and every rank
    ends with bit-identical results (no parameter broadcast is needed)"""
    from scipy.optimize import minimize

    n, nb = 260, 128
    world = pr * pc
    out = str(tmp_path / "fit.npy")
    mp.spawn(_fit_worker, args=(world, _free_port(), pr, pc, n, nb, q, out), nprocs=world, join=True)
    got = np.load(out)
    x, z, th0 = _fit_problem(n)

    def fun(th):
        try:
            if q:
                return orc.reml_value_and_grad(x, z, np.ones((n, 1)), 2, th, noise_index=1)
            return orc.nll_zero_mean_value_and_grad(x, z, 2, th, noise_index=1)
        except np.linalg.LinAlgError:           # the same back-off as fit_covparam at a point where K has no factor
            return 1e300, np.zeros_like(th)

    ref = minimize(fun, th0, jac=True, method="L-BFGS-B", options={"ftol": 1e-6, "maxiter": 15})
    assert ref.fun < fun(th0)[0] - 1.0                                   # the fit really moved
    assert abs(got[4] - ref.fun) < 1e-6 * abs(ref.fun)
    assert np.max(np.abs(got[:4] - ref.x)) < 1e-3


def _model_worker(rank, world, port, pr, pc, meantype, out):
    _init(rank, world, port)
    try:
        from gpmp_amd.dist import DistributedModel, ProcessGrid
        from tests.cpu_local_ops import CpuLocalOps

        n, m = 420, 57
        x, z = make_xz(n, 3, 7)
        xt, _ = make_xz(m, 3, 8)
        th = theta_aniso(3, scale=0.4)
        mean = {"zero": None, "parameterized": lambda a, p: (p[0] + p[1] * a[:, 0]).reshape(-1, 1),
                "linear_predictor": lambda a, p: np.hstack((np.ones((len(a), 1)), a))}[meantype]
        mp_ = np.array([0.3, -0.7]) if meantype == "parameterized" else None
        model = DistributedModel(ProcessGrid(pr, pc), mean, _cov_full, mp_, th, meantype, nb=128, ops=CpuLocalOps())
        zpm, zpv = model.predict(x, z, xt)
        # (round 4) the kriging weights: the full n x m matrix on every rank, and this rank's block on request
        zpm_l, zpv_l, lam = model.predict(x, z, xt, return_lambdas=True)
        assert np.array_equal(zpm_l, zpm) and np.array_equal(zpv_l, zpv) and lam.shape == (n, m)
        _, _, (blk, rows, (j0, j1)) = model.predict(x, z, xt, return_lambdas="local")
        assert np.allclose(blk.cpu().numpy(), lam[np.ix_(rows, np.arange(j0, j1))], rtol=0, atol=1e-13)
        zloo, s2, eloo = model.loo(x, z)
        crit = {"zero": lambda: model.negative_log_likelihood_zero_mean(th, x, z),
                "parameterized": lambda: model.negative_log_likelihood(mp_, th, x, z),
                "linear_predictor": lambda: model.negative_log_restricted_likelihood(th, x, z)}[meantype]()
        if rank == world - 1:                          # every rank holds the full results: take them from the LAST one
            np.save(out, np.concatenate((zpm, zpv, zloo, s2, eloo, [crit])))
            np.save(out + ".lam.npy", lam)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,meantype", [(2, 2, "zero"), (1, 2, "parameterized"), (2, 2, "linear_predictor")])
def test_distributed_model_surface_matches_the_oracle_model(tmp_path, pr, pc, meantype):
    """DistributedModel (gpmp/core/model.py's predict / loo / likelihood surface on the block-cyclic factor) against the oracle's
    Model for the three mean types: same calls on every rank, full-length results on every rank"""
    world = pr * pc
    out = str(tmp_path / "m.npy")
    mp.spawn(_model_worker, args=(world, _free_port(), pr, pc, meantype, out), nprocs=world, join=True)
    got = np.load(out)
    n, m = 420, 57
    x, z = make_xz(n, 3, 7)
    xt, _ = make_xz(m, 3, 8)
    th = theta_aniso(3, scale=0.4)
    mean = {"zero": None, "parameterized": lambda a, p: (p[0] + p[1] * a[:, 0]).reshape(-1, 1),
            "linear_predictor": lambda a, p: np.hstack((np.ones((len(a), 1)), a))}[meantype]
    mp_ = np.array([0.3, -0.7]) if meantype == "parameterized" else None
    om = orc.OracleModel(mean, _cov_full, mp_, th, meantype)
    rm, rv, rlam = orc.predict(om, x, z, xt, return_lambdas=True)
    lam = np.load(out + ".lam.npy")
    assert np.max(np.abs(lam - rlam)) < 1e-7 * np.max(np.abs(rlam)), np.max(np.abs(lam - rlam))      # lambda_t rel 1e-7
    rz, rs, re_ = orc.loo(om, x, z)
    rc = {"zero": lambda: orc.negative_log_likelihood_zero_mean(om, th, x, z), "parameterized": lambda: orc.negative_log_likelihood(om, mp_, th, x, z),
          "linear_predictor": lambda: orc.negative_log_restricted_likelihood(om, th, x, z)}[meantype]()
    zs = np.max(np.abs(z))
    o = 0
    for ref, tol in ((rm, 1e-8 * zs), (rv, 1e-8), (rz, 1e-7 * zs), (rs, 1e-7 * np.max(rs)), (re_, 1e-7 * zs)):
        seg = got[o: o + len(ref)]
        o += len(ref)
        assert np.max(np.abs(seg - ref)) < tol
    assert abs(got[-1] - float(rc)) < 1e-9 * abs(float(rc))
