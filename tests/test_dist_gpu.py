"""GPU tests of the distributed layer with the real local arithmetic (HipLocalOps through the C ABI).  RCCL needs one GPU per
rank and this pool has one-GPU boxes, so the ranks SHARE the test GPU: as processes over gloo (host-staged messages; at most five
beside the test runner: the pool's process guard), as thread-ranks on the stream-ordered in-process fabric of tools/thread_ranks.py
(device-resident messages ordered by stream events only: RCCL's stream semantics, the 2 x 4 grid of config 5), and -- one rank --
through ProcessGroupNCCL itself.  The schedule and the kernels are the ones the 8-GPU run uses."""
import math
import os
import socket

import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc
from tests.helpers import make_xz, theta_aniso

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, pr, pc, n, nb, out, transport, lookahead, backend="gloo", step_abi="1"):
    import torch.distributed as dist

    # gloo: every rank on the one test GPU; nccl (= RCCL): one GPU per rank
    local = str(rank) if backend == "nccl" else "0"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=local,
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(int(local))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", int(local)))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpmp_amd.num as gnp
        from gpmp_amd.dist import BlockCyclicCholesky, HipLocalOps, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance

        x, z = make_xz(n, 4, 11)
        th = theta_aniso(4, scale=0.5)
        cov = MaternCovariance(2)
        nugget = 10.0 * math.exp(th[0]) * gnp.eps
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=HipLocalOps(), transport=transport, lookahead=lookahead, step_abi=(step_abi == "1"))
        ch.build_local_gram(cov, x, th, nugget)
        info = ch.factor()
        nll = ch.negative_log_likelihood(z)
        payload = (ch.global_row_index(), ch.global_col_index(), gnp.to_np(ch.A).copy(), info, nll)
        gathered = [None] * world
        dist.all_gather_object(gathered, payload)
        if rank == 0:
            L = np.zeros((n, n))
            for (r_i, c_i, a, _, _) in gathered:
                if len(r_i) and len(c_i):
                    L[np.ix_(r_i, c_i)] = a
            np.save(out, np.tril(L))
            np.save(out + ".meta.npy", np.array([info, nll], dtype=np.float64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb,transport,lookahead", [(1, 1, 1500, 512, "bcast", True), (1, 2, 1500, 256, "bcast", True),
                                                             (2, 2, 2000, 256, "p2p", True), (2, 2, 2000, 256, "bcast", False)])
def test_block_cyclic_cholesky_hip(tmp_path, pr, pc, n, nb, transport, lookahead):
    _run_and_check(tmp_path, pr, pc, n, nb, transport, lookahead, "gloo")


@pytest.mark.parametrize("pr,pc,n,nb,transport,lookahead", [(1, 1, 1500, 512, "bcast", True),
                                                             (1, 2, 1500, 256, "bcast", True), (2, 1, 1500, 256, "p2p", True),
                                                             (1, 2, 1500, 256, "p2p", False), (2, 1, 2000, 256, "bcast", False),
                                                             (2, 2, 2000, 256, "p2p", True), (2, 2, 2000, 256, "bcast", True),
                                                             (2, 4, 4000, 256, "bcast", True), (2, 4, 4000, 256, "p2p", True)])
def test_block_cyclic_cholesky_rccl(tmp_path, pr, pc, n, nb, transport, lookahead):
    """The same schedule over RCCL, one GPU per rank: both transports, look-ahead on and off.  Needs pr * pc GPUs --
    skipped on the one-GPU boxes of this pool (only the 1 x 1 grid runs there), so until a multi-GPU node has run it
    the RCCL path with more than one rank is unproven.  What the 1 x 1 grid CAN prove under RCCL -- every other distributed
    entry point through ProcessGroupNCCL -- is test_every_distributed_entry_point_under_rccl below (round 5)."""
    if torch.cuda.device_count() < pr * pc:
        pytest.skip(f"needs {pr * pc} GPUs, have {torch.cuda.device_count()}")
    _run_and_check(tmp_path, pr, pc, n, nb, transport, lookahead, "nccl")


@pytest.mark.parametrize("pr,pc,n,nb,lookahead", [(1, 2, 1500, 256, True), (2, 1, 1500, 256, False), (2, 2, 3100, 256, True)])
def test_step_abi_equals_the_tensor_level_schedule(tmp_path, pr, pc, n, nb, lookahead):
    """gpmp_dist_* (diag_factor / panel_solve / exchange_pack + unpack / trailing_update: what a C++ RCCL host calls) against
    the tensor-level code of gpmp_amd/dist for the same schedule: the same kernels on the same operands in the same order,
    so the assembled factor must be IDENTICAL bit for bit (ragged last block: n = 2000 / 256 leaves 208 rows, whose panel
    solve takes the substitution route; n = 3100: two block columns per rank and a ragged last block)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = pr * pc
    outs = []
    for abi in ("1", "0"):
        out = str(tmp_path / f"L{abi}.npy")
        _spawn_bounded(_worker, (world, _free_port(), pr, pc, n, nb, out, "bcast", lookahead, "gloo", abi), world, 600.0)
        outs.append((np.load(out), np.load(out + ".meta.npy")))
    (L1, m1), (L0, m0) = outs
    assert m1[0] == 0 and m0[0] == 0
    assert np.array_equal(L1, L0) and m1[1] == m0[1]


def _spawn_bounded(fn, args, nprocs, limit_s):
    """mp.spawn with a deadline: a collective that never completes (the RCCL path with more than one rank has not run on
    hardware yet) must fail THIS test, not hang the suite and the GPU box -- the workers are killed and the test fails."""
    import time
    import torch.multiprocessing as mp

    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.monotonic() + limit_s
    while not ctx.join(timeout=2.0):
        if time.monotonic() > deadline:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            for p in ctx.processes:
                p.join(10)
            pytest.fail(f"distributed workers still running after {limit_s:.0f} s: killed (wedged collective?)")


def _run_and_check(tmp_path, pr, pc, n, nb, transport, lookahead, backend):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = pr * pc
    out = str(tmp_path / "L.npy")
    _spawn_bounded(_worker, (world, _free_port(), pr, pc, n, nb, out, transport, lookahead, backend), world, 240.0 if backend == "nccl" else 600.0)
    L = np.load(out)
    info, nll = np.load(out + ".meta.npy")
    x, z = make_xz(n, 4, 11)
    th = theta_aniso(4, scale=0.5)
    K = orc.maternp_covariance(x, None, 2, th)
    Lref = np.linalg.cholesky(K)
    assert info == 0
    assert np.max(np.abs(L @ L.T - K)) / np.max(np.abs(K)) < 1e-13
    om = orc.OracleModel(None, lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise), None, th, "zero")
    ref = float(orc.negative_log_likelihood_zero_mean(om, th, x, z))
    assert abs(nll - ref) < 1e-8 * abs(ref)


def _predict_worker(rank, world, port, pr, pc, n, m, nb, out, transport, overlap=True):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpmp_amd.num as gnp
        from gpmp_amd.dist import BlockCyclicCholesky, HipLocalOps, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance

        x, z = make_xz(n, 4, 11)
        xt, _ = make_xz(m, 4, 12)
        th = theta_aniso(4, scale=0.5)
        cov = MaternCovariance(2)
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=HipLocalOps(), transport=transport)
        ch.oplog = []
        ch.build_local_gram(cov, x, th, 10.0 * math.exp(th[0]) * gnp.eps)
        assert ch.factor() == 0
        os.environ["GPMP_DIST_SOLVE_OVERLAP"] = "1" if overlap else "0"
        mean, var, (j0, j1) = ch.predict_zero_mean(cov, x, z, xt, th)
        mean_l, var_l, _, lam = ch.predict(cov, x, z, xt, th, return_lambdas=True)     # (round 5) + the overlapped BACKWARD solve
        assert np.array_equal(mean_l, mean) and np.array_equal(var_l, var)
        # every communicator is driven from ONE stream per phase (what keeps RCCL's per-communicator streams independent):
        # factorisation: row / column communicators from the side stream, the diagonal-block communicator from the diagonal
        # stream; many-right-hand-side solve: row communicators from the prefetch (= diagonal) stream, column from the side
        for (tag, op, root, numel, step, role) in ch.oplog:
            if step.startswith(("panel", "diag")):
                assert role == ("diag" if tag.startswith("diag") else "side"), (tag, step, role)
            elif step.startswith("solve_pre"):
                assert tag.startswith("row") and role == ("diag" if overlap else "host"), (tag, step, role)
            elif step.startswith("solve_chain"):
                assert tag.startswith("col") and role == ("side" if overlap else "host"), (tag, step, role)
            elif step.startswith("bsolve_pre"):          # backward solve: factor data along the process rows, prefetch stream
                assert tag.startswith("row") and role == ("diag" if overlap else "host"), (tag, step, role)
            elif step.startswith("bsolve"):              # its chain: ONE reduce per block column inside the process column, side stream
                assert tag.startswith("col") and op == "reduce:bsolve" and role == ("side" if overlap else "host"), (tag, step, role)
        gathered = [None] * world
        dist.all_gather_object(gathered, (grid.r, j0, j1, mean, var, ch.global_row_index(), lam.cpu().numpy()))
        if rank == 0:
            zpm, zpv, full = np.full(m, np.nan), np.full(m, np.nan), np.full((n, m), np.nan)
            for (r, a, b, mu, v, rows, blk) in gathered:
                if r == 0:
                    zpm[a:b], zpv[a:b] = mu, v
                if len(rows) and b > a:
                    full[np.ix_(rows, np.arange(a, b))] = blk
            np.save(out, np.stack([zpm, zpv]))
            np.save(out + ".lam.npy", full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,m,nb,transport,overlap", [(1, 1, 1500, 700, 512, "bcast", True),
                                                            (1, 2, 1500, 333, 256, "p2p", True), (2, 2, 2000, 901, 256, "p2p", False),
                                                            (2, 2, 3100, 1201, 256, "bcast", True), (2, 1, 2304, 513, 256, "bcast", True)])
def test_block_cyclic_predict_hip(tmp_path, pr, pc, n, m, nb, transport, overlap):
    """many-right-hand-side solve on the block-cyclic factor with the real kernels (ranks share the test GPU over gloo):
    posterior mean / variance against the oracle's predict; with the prefetch / chain / update streams overlapped (default)
    and in program order on one stream"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = pr * pc
    out = str(tmp_path / "p.npy")
    _spawn_bounded(_predict_worker, (world, _free_port(), pr, pc, n, m, nb, out, transport, overlap), world, 600.0)
    got = np.load(out)
    x, z = make_xz(n, 4, 11)
    xt, _ = make_xz(m, 4, 12)
    th = theta_aniso(4, scale=0.5)
    om = orc.OracleModel(None, lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise), None, th, "zero")
    rm, rv, rlam = orc.predict(om, x, z, xt, return_lambdas=True, zero_neg_variances=False)
    assert np.max(np.abs(got[0] - rm)) < 1e-8 * np.max(np.abs(z))
    assert np.max(np.abs(got[1] - rv)) < 1e-8
    lam = np.load(out + ".lam.npy")
    ev = np.linalg.eigvalsh(orc.maternp_covariance(x, None, 2, th))
    assert np.max(np.abs(lam - rlam)) < 1e-7 * max(1.0, float(ev[-1] / ev[0]) / 1e6) * np.max(np.abs(rlam))


@pytest.mark.parametrize("n,nb", [(4096, 512), (3000, 256), (2048, 1024)])
def test_cpp_rccl_host_on_the_step_abi(tmp_path, n, nb):
    """examples/dist_potrf_rccl.cpp: a C++ host that owns the RCCL communicators and drives gpmp_dist_* (no torch, no
    Python) -- here on the 1 x 1 grid a one-GPU box allows (RCCL initialised with one rank, every broadcast a no-op): info 0
    and log|K| equal to the single-GPU factorisation of the same matrix.  With N GPUs the same binary runs as N processes."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "dist_potrf_rccl.bin")
    if not torch.cuda.is_available() or not os.path.exists(exe):
        pytest.skip("no GPU or example not built (run __graft_entry__.build())")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "0", "1", "1", "1", str(n), str(nb), str(tmp_path / "nccl_id")], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-1000:], p.stderr[-2000:])
    assert re.search(r"info=0 ", p.stdout)
    rel = float(re.search(r"rel_diff=([-+0-9.eE]+)", p.stdout).group(1))
    assert rel < 1e-12, p.stdout


def _reml_loo_worker(rank, world, port, pr, pc, n, nb, q, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpmp_amd.num as gnp
        from gpmp_amd.dist import BlockCyclicCholesky, HipLocalOps, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance

        x, z = make_xz(n, 4, 11)
        th = theta_aniso(4, scale=0.5)
        P = None if q == 0 else np.hstack((np.ones((n, 1)), x))[:, :q]
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=HipLocalOps())
        ch.build_local_gram(MaternCovariance(2), x, th, 10.0 * math.exp(th[0]) * gnp.eps)
        assert ch.factor() == 0
        v = ch.negative_log_restricted_likelihood(z, P) if q else ch.negative_log_likelihood(z)
        zloo, s2, eloo, idx = ch.loo(z, P)
        gathered = [None] * world
        dist.all_gather_object(gathered, (grid.r, idx, zloo, s2, eloo))
        if rank == 0:
            full = np.full((3, n), np.nan)
            for (r, ix, zl, s, e) in gathered:
                if r == 0:
                    full[0, ix], full[1, ix], full[2, ix] = zl, s, e
            np.save(out, np.vstack((full, np.full((1, n), v))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb,q", [(2, 2, 2000, 256, 1), (1, 2, 1500, 256, 0), (2, 1, 1800, 512, 3)])
def test_block_cyclic_reml_and_loo_hip(tmp_path, pr, pc, n, nb, q):
    """REML and leave-one-out from the block-cyclic factor with the real kernels (ranks share the test GPU over gloo) against the
    oracle (complete-QR REML, explicit-inverse LOO); LOO's T = L^-1 goes through the overlapped many-right-hand-side solve"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = pr * pc
    out = str(tmp_path / "r.npy")
    _spawn_bounded(_reml_loo_worker, (world, _free_port(), pr, pc, n, nb, q, out), world, 600.0)
    got = np.load(out)
    x, z = make_xz(n, 4, 11)
    th = theta_aniso(4, scale=0.5)
    cov = lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise)       # noqa: E731
    if q == 0:
        om = orc.OracleModel(None, cov, None, th, "zero")
        ref_v = float(orc.negative_log_likelihood_zero_mean(om, th, x, z))
    else:
        om = orc.OracleModel(lambda a, p: np.hstack((np.ones((len(a), 1)), a))[:, :q], cov, None, th, "linear_predictor")
        ref_v = float(orc.negative_log_restricted_likelihood(om, th, x, z))
    rz, rs, re_ = orc.loo(om, x, z)
    ev = np.linalg.eigvalsh(orc.maternp_covariance(x, None, 2, th))
    cs = max(1.0, float(ev[-1] / ev[0]) / 1e6)
    assert abs(got[3, 0] - ref_v) < 1e-11 * cs * abs(ref_v)
    assert np.max(np.abs(got[0] - rz)) < 1e-8 * cs * np.max(np.abs(z))
    assert np.max(np.abs(got[1] - rs) / rs) < 1e-8 * cs
    assert np.max(np.abs(got[2] - re_)) < 1e-8 * cs * np.max(np.abs(z))


def _grad_worker(rank, world, port, pr, pc, n, nb, q, noise, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpmp_amd.num as gnp
        from gpmp_amd.dist import BlockCyclicCholesky, HipLocalOps, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance

        x, z = make_xz(n, 4, 11)
        th = theta_aniso(4, scale=0.5)
        if noise:
            th = np.concatenate(([th[0], math.log(1e-3)], th[1:]))
        P = None if q == 0 else np.hstack((np.ones((n, 1)), x))[:, :q]
        diag = math.exp(th[1]) if noise else 10.0 * math.exp(th[0]) * gnp.eps
        ch = BlockCyclicCholesky(ProcessGrid(pr, pc), n, nb=nb, ops=HipLocalOps())
        ch.build_local_gram(MaternCovariance(2, noise=noise), x, th, diag)
        assert ch.factor() == 0
        v, gvec = ch.value_and_grad(x, z, th, 2, noise=noise, P=P)
        if rank == 0:
            np.save(out, np.concatenate(([v], gvec)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb,q,noise", [(2, 2, 2000, 256, 1, False), (1, 2, 1500, 256, 0, True), (1, 3, 1700, 256, 2, False), (1, 1, 1300, 512, 1, False)])
def test_block_cyclic_value_and_gradient_hip(tmp_path, pr, pc, n, nb, q, noise):
    """ML / REML value + analytic gradient from the block-cyclic factor with the real kernels (gpmp_matern_grad_trace_cross on
    the blocks of T^T T; ranks share the test GPU over gloo) against the oracle's analytic gradient"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = pr * pc
    out = str(tmp_path / "g.npy")
    _spawn_bounded(_grad_worker, (world, _free_port(), pr, pc, n, nb, q, noise, out), world, 600.0)
    got = np.load(out)
    x, z = make_xz(n, 4, 11)
    th = theta_aniso(4, scale=0.5)
    if noise:
        th = np.concatenate(([th[0], math.log(1e-3)], th[1:]))
    ni = 1 if noise else None
    if q == 0:
        v, g = orc.nll_zero_mean_value_and_grad(x, z, 2, th, noise_index=ni)
    else:
        v, g = orc.reml_value_and_grad(x, z, np.hstack((np.ones((n, 1)), x))[:, :q], 2, th, noise_index=ni)
    cov = orc.noisy_maternp_covariance if noise else orc.maternp_covariance
    ev = np.linalg.eigvalsh(cov(x, None, 2, th))
    cs = max(1.0, float(ev[-1] / ev[0]) / 1e6)
    assert abs(got[0] - v) < 1e-11 * cs * abs(v)
    assert np.max(np.abs(got[1:] - g)) < 1e-8 * cs * np.max(np.abs(g)), (got[1:], g, cs)


def _uk_worker(rank, world, port, pr, pc, n, m, nb, q, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpmp_amd.num as gnp
        from gpmp_amd.dist import BlockCyclicCholesky, HipLocalOps, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance

        x, z = make_xz(n, 4, 11)
        xt, _ = make_xz(m, 4, 12)
        th = theta_aniso(4, scale=0.5)
        mean = lambda a: np.hstack((np.ones((len(a), 1)), a))[:, :q]        # noqa: E731
        cov = MaternCovariance(2)
        grid = ProcessGrid(pr, pc)
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=HipLocalOps())
        ch.build_local_gram(cov, x, th, 10.0 * math.exp(th[0]) * gnp.eps)
        assert ch.factor() == 0
        mu, var, (j0, j1) = ch.predict(cov, x, z, xt, th, P=mean(x), Pt=mean(xt))
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


@pytest.mark.parametrize("pr,pc,n,m,nb,q", [(2, 2, 2000, 901, 256, 1), (1, 2, 1500, 333, 256, 5)])
def test_block_cyclic_universal_kriging_hip(tmp_path, pr, pc, n, m, nb, q):
    """universal kriging (constant / linear mean) from the block-cyclic factor with the real kernels against the oracle's
    block-system solve (gpmp/core/kriging.py:70-116)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = pr * pc
    out = str(tmp_path / "p.npy")
    _spawn_bounded(_uk_worker, (world, _free_port(), pr, pc, n, m, nb, q, out), world, 600.0)
    got = np.load(out)
    x, z = make_xz(n, 4, 11)
    xt, _ = make_xz(m, 4, 12)
    th = theta_aniso(4, scale=0.5)
    om = orc.OracleModel(lambda a, p: np.hstack((np.ones((len(a), 1)), a))[:, :q],
                         lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise), None, th, "linear_predictor")
    rm, rv = orc.predict(om, x, z, xt, zero_neg_variances=False)
    ev = np.linalg.eigvalsh(orc.maternp_covariance(x, None, 2, th))
    cs = max(1.0, float(ev[-1] / ev[0]) / 1e6)
    assert np.max(np.abs(got[0] - rm)) < 1e-9 * cs * np.max(np.abs(z))
    assert np.max(np.abs(got[1] - rv)) < 1e-9 * cs


@pytest.mark.parametrize("pr,pc,n,nb", [(2, 3, 1900, 128), (3, 2, 1900, 128), (2, 4, 1900, 128), (4, 2, 1900, 128), (3, 3, 1900, 128), (1, 4, 1900, 128),
                                        (4, 1, 1900, 128), (2, 3, 2304, 256), (3, 3, 2304, 256),
                                        # nb = 1024: the trailing update is ONE launch over the staircase tile set (round 5)
                                        (2, 4, 7000, 1024), (2, 3, 6500, 1024), (3, 2, 5200, 1024), (1, 4, 5120, 1024), (4, 1, 4500, 1024)])
def test_step_abi_local_kernels_on_any_grid_single_process(pr, pc, n, nb):
    """gpmp_dist_exchange_pack / _unpack / _trailing_update for EVERY rank coordinate of non-square grids (lcm(Pr, Pc) strides,
    ragged last blocks), in ONE process: the buffers a rank would hold after the broadcasts are cut out of a global panel
    with NumPy, the C entry points run on them, and the results are compared with the block-cyclic definition -- the grids
    the multi-process GPU tests cannot reach on a one-GPU box (at most six processes may share it)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd.num as gnp
    from gpmp_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(pr * 100 + pc * 10 + nb)
    nblk = (n + nb - 1) // nb
    bs = lambda I: min(nb, n - I * nb)                 # noqa: E731
    rows_of = lambda blocks: np.concatenate([np.arange(I * nb, I * nb + bs(I)) for I in blocks]) if blocks else np.zeros(0, dtype=np.int64)   # noqa: E731
    A = rng.standard_normal((n, n))
    for k in sorted({0, nblk // 2, nblk - 2}):
        bk = bs(k)
        Lk = rng.standard_normal((n, bk))             # the (already solved) block column k: rows = global rows
        for r in range(pr):
            for c in range(pc):
                rb, cb = list(range(r, nblk, pr)), list(range(c, nblk, pc))
                gr, gc = rows_of(rb), rows_of(cb)
                if len(gr) == 0 or len(gc) == 0:
                    continue
                Aloc = gnp.as_matrix(gnp.asarray(A[np.ix_(gr, gc)]), copy=True)
                prow, crow = rows_of([I for I in rb if I > k]), rows_of([J for J in cb if J > k])
                panel = gnp.alloc_matrix(max(len(prow), 1), nb, zero=True)
                if len(prow):
                    panel[: len(prow), :bk] = gnp.asarray(Lk[prow])
                colop = gnp.alloc_matrix(max(len(crow), 1), nb, zero=True)
                # ---- column operand through pack (as the holder rank (rp, c) would) + unpack (as this rank does)
                for rp in range(pr):
                    rows = int(lib.gpmp_dist_exchange_rows(n, nb, pr, pc, rp, c, k))
                    want_rows = rows_of([J for J in cb if J > k and J % pr == rp])
                    assert rows == len(want_rows)
                    if rows == 0:
                        continue
                    hb = list(range(rp, nblk, pr))                                  # the holder's block rows
                    hrow = rows_of([I for I in hb if I > k])
                    hpanel = gnp.alloc_matrix(max(len(hrow), 1), nb, zero=True)
                    hpanel[: len(hrow), :bk] = gnp.asarray(Lk[hrow])
                    piece = gnp.alloc_matrix(rows, nb, zero=True)
                    _lib.check(lib.gpmp_dist_exchange_pack(gnp._ptr(hpanel), gnp._ld(hpanel), gnp._ptr(piece), gnp._ld(piece), n, nb, pr, pc, rp, c, k,
                                                           bk, gnp._stream()), "pack")
                    assert np.array_equal(gnp.to_np(piece[:, :bk]), Lk[want_rows])
                    _lib.check(lib.gpmp_dist_exchange_unpack(gnp._ptr(piece), gnp._ld(piece), gnp._ptr(colop), gnp._ld(colop), n, nb, pr, pc, rp, c, k,
                                                             bk, gnp._stream()), "unpack")
                if len(crow):
                    assert np.array_equal(gnp.to_np(colop[: len(crow), :bk]), Lk[crow])
                # ---- trailing update of the local blocks I >= J > k
                _lib.check(lib.gpmp_dist_trailing_update(gnp._ptr(Aloc), gnp._ld(Aloc), n, nb, pr, pc, r, c, k, gnp._ptr(panel), gnp._ld(panel),
                                                         gnp._ptr(colop), gnp._ld(colop), 0, -1, -1, gnp._stream()), "update")
                got = gnp.to_np(Aloc)
                want = A[np.ix_(gr, gc)].copy()
                full = A - Lk @ Lk.T
                for li, I in enumerate(rb):
                    for lj, J in enumerate(cb):
                        if I >= J > k:
                            r0, c0 = li * nb, lj * nb
                            blk_w = full[I * nb: I * nb + bs(I), J * nb: J * nb + bs(J)]
                            blk_g = got[r0: r0 + bs(I), c0: c0 + bs(J)]
                            assert np.max(np.abs(blk_g - blk_w)) < 1e-11 * np.max(np.abs(blk_w)), (pr, pc, r, c, k, I, J)
                # blocks in block columns <= k are never touched
                for lj, J in enumerate(cb):
                    if J <= k:
                        assert np.array_equal(got[:, lj * nb: lj * nb + bs(J)], want[:, lj * nb: lj * nb + bs(J)])
                # the split the look-ahead schedule makes -- ONE block column first (restricted to the rows below block k + 1 when this
                # rank owns the diagonal block (k + 1, k + 1)), the rest after -- gives the same needed blocks bit for bit
                jcol = [lj for lj, J in enumerate(cb) if J > k]
                if jcol:
                    Asp = gnp.as_matrix(gnp.asarray(A[np.ix_(gr, gc)]), copy=True)
                    j1 = jcol[0]
                    ra = cb[j1] if (cb[j1] in rb) else -1
                    for (a_, b_, rows_after) in ((j1, j1 + 1, ra), (j1 + 1, len(cb), -1)):
                        _lib.check(lib.gpmp_dist_trailing_update(gnp._ptr(Asp), gnp._ld(Asp), n, nb, pr, pc, r, c, k, gnp._ptr(panel), gnp._ld(panel),
                                                                 gnp._ptr(colop), gnp._ld(colop), a_, b_, rows_after, gnp._stream()), "update (split)")
                    gsp = gnp.to_np(Asp)
                    for li, I in enumerate(rb):
                        for lj, J in enumerate(cb):
                            if I >= J > k and not (lj == j1 and ra >= 0 and I <= ra):
                                assert np.array_equal(gsp[li * nb: li * nb + bs(I), lj * nb: lj * nb + bs(J)],
                                                      got[li * nb: li * nb + bs(I), lj * nb: lj * nb + bs(J)]), (pr, pc, r, c, k, I, J)


def _model_worker(rank, world, port, pr, pc, meantype, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpmp_amd.dist import DistributedModel, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance

        n, m = 1800, 333
        x, z = make_xz(n, 4, 11)
        xt, _ = make_xz(m, 4, 12)
        th = theta_aniso(4, scale=0.5)
        mean = None if meantype == "zero" else (lambda a, p: np.hstack((np.ones((len(a), 1)), a)))
        model = DistributedModel(ProcessGrid(pr, pc), mean, MaternCovariance(2), None, th, meantype, nb=256)
        zpm, zpv = model.predict(x, z, xt)
        zpm_l, zpv_l, lam = model.predict(x, z, xt, return_lambdas=True)       # (round 4) weights: the backward solve with the real kernels
        assert np.array_equal(zpm_l, zpm) and np.array_equal(zpv_l, zpv)
        zloo, s2, eloo = model.loo(x, z)
        crit = model.negative_log_likelihood_zero_mean(th, x, z) if meantype == "zero" else model.negative_log_restricted_likelihood(th, x, z)
        if rank == world - 1:
            np.save(out, np.concatenate((zpm, zpv, zloo, s2, eloo, [crit])))
            np.save(out + ".lam.npy", lam)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,meantype", [(2, 2, "linear_predictor"), (1, 2, "zero")])
def test_distributed_model_surface_hip(tmp_path, pr, pc, meantype):
    """DistributedModel.predict / loo / criterion with the real kernels (ranks share the test GPU over gloo) against the oracle's Model"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = pr * pc
    out = str(tmp_path / "m.npy")
    _spawn_bounded(_model_worker, (world, _free_port(), pr, pc, meantype, out), world, 600.0)
    got = np.load(out)
    n, m = 1800, 333
    x, z = make_xz(n, 4, 11)
    xt, _ = make_xz(m, 4, 12)
    th = theta_aniso(4, scale=0.5)
    mean = None if meantype == "zero" else (lambda a, p: np.hstack((np.ones((len(a), 1)), a)))
    om = orc.OracleModel(mean, lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise), None, th, meantype)
    rm, rv, rlam = orc.predict(om, x, z, xt, return_lambdas=True)
    rz, rs, re_ = orc.loo(om, x, z)
    rc = float(orc.negative_log_likelihood_zero_mean(om, th, x, z)) if meantype == "zero" else float(orc.negative_log_restricted_likelihood(om, th, x, z))
    ev = np.linalg.eigvalsh(orc.maternp_covariance(x, None, 2, th))
    cs = max(1.0, float(ev[-1] / ev[0]) / 1e6)
    lam = np.load(out + ".lam.npy")
    assert lam.shape == rlam.shape and np.max(np.abs(lam - rlam)) < 1e-7 * cs * np.max(np.abs(rlam))
    zs = np.max(np.abs(z))
    o = 0
    for ref, tol in ((rm, 1e-9 * cs * zs), (rv, 1e-9 * cs), (rz, 1e-8 * cs * zs), (rs, 1e-8 * cs * np.max(rs)), (re_, 1e-8 * cs * zs)):
        seg = got[o: o + len(ref)]
        o += len(ref)
        assert np.max(np.abs(seg - ref)) < tol
    assert abs(got[-1] - rc) < 1e-11 * cs * abs(rc)


@pytest.mark.parametrize("pr,pc,transport", [(1, 2, "p2p")])
def test_device_resident_communication_branches_over_gloo_cuda(pr, pc, transport):
    """(round 4) The branches that only run under RCCL -- comm tensors, scalars and the few-column solve kept on the DEVICE,
    collectives called with CUDA tensors from the three streams -- exercised without RCCL: gloo moves CUDA tensors too, so the
    ranks share the one GPU with ``backend`` forced to the RCCL code path (tools/gloo_cuda_comm_probe.py).  Factorisation, NLL,
    REML, universal kriging with weights, leave-one-out and the REML gradient against the oracle.  Kept for the ``p2p`` transport
    (grouped send / recv of CUDA tensors through torch.distributed); the ``bcast`` form of this branch is checked more strictly
    since round 5 by test_every_entry_point_on_the_stream_ordered_fabric (no host synchronisation around a message)."""
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gloo_cuda_comm_probe.py"), str(pr), str(pc), transport],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "DEVICE-COMM PROBE OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("pr,pc,transport", [(1, 1, "bcast"), (1, 1, "p2p"), (1, 2, "bcast"), (2, 2, "p2p")])
def test_every_distributed_entry_point_under_rccl(pr, pc, transport):
    """(round 5) Every distributed entry point through ``ProcessGroupNCCL`` itself: factorisation, NLL, REML, universal kriging
    WITH weights (forward + backward many-right-hand-side solves), leave-one-out, the REML and the noisy-ML value + gradient, the
    ``DistributedModel`` surface (predict with ``return_lambdas=True``, loo, criterion) and a short ``fit_covparam`` run, against
    the oracle.  RCCL wants one GPU per rank, so a one-GPU box runs the 1 x 1 grid (both transports) and skips the rest: that
    covers communicator creation with the high-priority options (gpmp_amd/dist/grid.py:14-45), the four communicator kinds, the
    device-resident comm tensors (`_comm_tensor` / `_gloo_cuda_guard`, gpmp_amd/dist/streams.py) and the work / stream waits under RCCL's own streams --
    not an exchange between two ranks."""
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.device_count() < pr * pc:
        pytest.skip(f"needs {pr * pc} GPUs, have {torch.cuda.device_count()}")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gloo_cuda_comm_probe.py"), str(pr), str(pc), transport, "nccl"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "DEVICE-COMM PROBE OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def _stream_ordered_entry_points(pr, pc, delay_cycles, n=2100, m=333, nb=256, with_model=False, **factor_kw):
    """every distributed entry point of the DEVICE-RESIDENT branch on the stream-ordered in-process fabric (tools/thread_ranks.py),
    thread-ranks sharing the GPU; -> dict of results assembled from the ranks"""
    import gpmp_amd.num as gnp  # noqa: F401 -- library loaded before the rank threads start
    from tools import thread_ranks

    d = 4
    x, z = make_xz(n, d, 11)
    xt, _ = make_xz(m, d, 12)
    th = theta_aniso(d, scale=0.5)
    P = np.hstack((np.ones((n, 1)), x[:, :1]))
    Pt = np.hstack((np.ones((m, 1)), xt[:, :1]))
    out = {}

    def body(rank, world, fabric, classes):
        from gpmp_amd.dist import HipLocalOps, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance

        cov = MaternCovariance(2)
        Ch = classes[1]
        Ch.max_delay_cycles = delay_cycles
        grid = ProcessGrid(pr, pc)
        ch = Ch(grid, n, nb=nb, ops=HipLocalOps(), **factor_kw)
        ch.build_local_gram(cov, x, th, 10.0 * math.exp(th[0]) * gnp.eps)
        info = ch.factor()
        nll = ch.negative_log_likelihood(z)
        reml = ch.negative_log_restricted_likelihood(z, P)
        mean, var, (j0, j1), lam = ch.predict(cov, x, z, xt, th, P=P, Pt=Pt, return_lambdas=True)
        zloo, s2, eloo, idx = ch.loo(z, P)
        val, grad = ch.value_and_grad(x, z, th, 2, P=P)
        torch.cuda.synchronize()
        out[rank] = (grid.r, j0, j1, mean, var, ch.global_row_index(), lam.cpu().numpy(), idx, zloo, info, nll, reml, val, grad)
        if with_model:
            # the Model surface and a short parameter fit on the same fabric (every rank makes the same calls)
            from gpmp_amd.dist import DistributedModel

            mean_fn = lambda a, p: np.hstack((np.ones((len(a), 1)), np.asarray(a)[:, :1]))        # noqa: E731
            model = DistributedModel(grid, mean_fn, cov, None, th, "linear_predictor", nb=nb, ops=HipLocalOps(), factor_class=Ch)
            m_zpm, m_zpv, m_lam = model.predict(x, z, xt, return_lambdas=True)
            m_zloo, _, _ = model.loo(x, z)
            m_reml = model.negative_log_restricted_likelihood(th, x, z)
            model.covparam = th + 0.3
            th_fit, fit = model.select_parameters(x, z, options={"maxiter": 2})
            torch.cuda.synchronize()
            model_out[rank] = (m_zpm, m_zpv, m_lam, m_zloo, m_reml, fit["history"][0][1], fit["fun"], th_fit)

    model_out = {}
    errors = thread_ranks.run(pr * pc, body, limit_s=300.0)
    assert not errors, errors[0]
    zpm, zpv, L, zl = np.full(m, np.nan), np.full(m, np.nan), np.full((n, m), np.nan), np.full(n, np.nan)
    for (r, a, b, mu, v, rows, blk, ix, zz, *_rest) in out.values():
        zpm[a:b], zpv[a:b] = mu, v
        if len(rows) and b > a:
            L[np.ix_(rows, np.arange(a, b))] = blk
        zl[ix] = zz
    o = out[0]
    res = dict(info=o[9], nll=o[10], reml=o[11], val=o[12], grad=o[13], zpm=zpm, zpv=zpv, lam=L, zloo=zl,
               same_scalars=all(v[10] == o[10] and v[11] == o[11] and np.array_equal(v[13], o[13]) for v in out.values()))
    if with_model:
        mo = model_out[0]
        res["model"] = dict(zpm=mo[0], zpv=mo[1], lam=mo[2], zloo=mo[3], reml=mo[4], fit_first=mo[5], fit_last=mo[6],
                            same_on_all_ranks=all(np.array_equal(v[0], mo[0]) and np.array_equal(v[7], mo[7]) for v in model_out.values()))
    return res


@pytest.mark.parametrize("pr,pc,delay_cycles,factor_kw", [(2, 4, 0, {}), (2, 4, 3_000_000, {}), (3, 2, 1_000_000, {}), (4, 2, 1_000_000, {}),
                                                           (3, 3, 1_000_000, {}), (1, 8, 1_000_000, {}), (2, 4, 1_000_000, {"lookahead": False}),
                                                           (2, 4, 1_000_000, {"reserve_cus": 16}), (2, 4, 1_000_000, {"_n": 4096, "_nb": 1024}),
                                                           (3, 2, 0, {"_n": 5003, "_nb": 1024})],
                         ids=lambda v: "-".join(f"{k}={w}" for k, w in v.items()) if isinstance(v, dict) else str(v))
def test_every_entry_point_on_the_stream_ordered_fabric(pr, pc, delay_cycles, factor_kw):
    """(round 5) The device-resident branch of gpmp_amd/dist (``backend == "nccl"``: what runs under RCCL) with RCCL's STREAM
    semantics and without RCCL: thread-ranks sharing the GPU, every collective enqueued on the member's current stream and ordered by
    HIP events only (tools/thread_ranks.py) -- no host synchronisation around a message, unlike gloo.  A missing dependency between
    the schedule's three streams (a panel buffer reused too early, a consumer that does not wait for its message) shows up as wrong
    values; ``delay_cycles`` > 0 holds every incoming message back by a pseudo-random time of up to ~1 ms on the receiving stream to
    widen any such window.  2 x 4 is the grid of BASELINE config 5; 3 x 2, 4 x 2, 3 x 3 and 1 x 8 are grids no multi-process test can reach
    on a one-GPU box (Pr > Pc, odd counts, a single process row of eight: the gradient ring with four shifts); the last two cases run
    the factorisation without look-ahead and with its bulk updates on a CU-masked stream (``reserve_cus``).  Factorisation,
    NLL, REML, universal kriging with weights, leave-one-out, REML value + gradient against the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    factor_kw = dict(factor_kw)
    # (_n / _nb: the block size of config 5, 1024 -- the one-launch staircase trailing update and the one-launch blocks of T^T T)
    n, nb, m, d = factor_kw.pop("_n", 2100), factor_kw.pop("_nb", 256), 333, 4
    with_model = (pr, pc, delay_cycles) == (2, 4, 0) and nb == 256
    g = _stream_ordered_entry_points(pr, pc, delay_cycles, n=n, m=m, nb=nb, with_model=with_model, **factor_kw)
    x, z = make_xz(n, d, 11)
    xt, _ = make_xz(m, d, 12)
    th = theta_aniso(d, scale=0.5)
    mean_fn = lambda a, p: np.hstack((np.ones((len(a), 1)), a[:, :1]))  # noqa: E731
    kern = lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise)  # noqa: E731
    om = orc.OracleModel(mean_fn, kern, None, th, "linear_predictor")
    oz = orc.OracleModel(None, kern, None, th, "zero")
    rm, rv, rl = orc.predict(om, x, z, xt, return_lambdas=True)
    rz, _, _ = orc.loo(om, x, z)
    rnll = float(orc.negative_log_likelihood_zero_mean(oz, th, x, z))
    rreml, rgrad = orc.reml_value_and_grad(x, z, mean_fn(x, None), 2, th)
    assert g["info"] == 0 and g["same_scalars"]
    assert abs(g["nll"] - rnll) < 1e-9 * abs(rnll) and abs(g["reml"] - rreml) < 1e-9 * abs(rreml) and abs(g["val"] - g["reml"]) < 1e-9 * abs(rreml)
    assert np.max(np.abs(g["zpm"] - rm)) < 1e-7 and np.max(np.abs(g["zpv"] - rv)) < 1e-7
    assert np.max(np.abs(g["lam"] - rl)) < 1e-6 * np.max(np.abs(rl))
    assert np.max(np.abs(g["zloo"] - rz)) < 1e-6
    assert np.max(np.abs(g["grad"] - rgrad)) < 1e-7 * np.linalg.norm(rgrad)
    if with_model:
        # DistributedModel.predict(return_lambdas=True) / loo / criterion / select_parameters on the same fabric
        mo = g["model"]
        assert mo["same_on_all_ranks"]
        assert np.max(np.abs(mo["zpm"] - rm)) < 1e-7 and np.max(np.abs(mo["zpv"] - np.maximum(rv, 0.0))) < 1e-7
        assert mo["lam"].shape == rl.shape and np.max(np.abs(mo["lam"] - rl)) < 1e-6 * np.max(np.abs(rl))
        assert np.max(np.abs(mo["zloo"] - rz)) < 1e-6 and abs(mo["reml"] - rreml) < 1e-9 * abs(rreml)
        assert mo["fit_last"] < mo["fit_first"]


def test_stream_ordered_fabric_random_soak():
    """Opt-in soak (GPMP_DIST_SOAK_CASES=<count>, GPMP_DIST_SOAK_SEED): random draws of grid (up to nine thread-ranks), size, block size,
    look-ahead and message delay -- including sizes so small that ranks hold no block at all, and sizes one off a multiple of the block
    -- each through every entry point of the test above, against the oracle.  The suite itself runs none (the cases above are fixed)."""
    import time
    import traceback

    ncases = int(os.environ.get("GPMP_DIST_SOAK_CASES", "0"))
    if ncases <= 0:
        pytest.skip("opt-in: GPMP_DIST_SOAK_CASES=<count>")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.default_rng(int(os.environ.get("GPMP_DIST_SOAK_SEED", "5")))
    grids = [(1, 1), (1, 2), (2, 1), (2, 2), (1, 3), (3, 1), (2, 3), (3, 2), (1, 5), (5, 1), (2, 4), (4, 2), (1, 7), (1, 8), (8, 1), (3, 3)]
    failures = []
    for i in range(ncases):
        pr, pc = grids[int(rng.integers(len(grids)))]
        nb = int(rng.choice([128, 256, 512, 1024]))
        kind = int(rng.integers(4))
        if kind == 0:            # fewer blocks than process rows / columns: ranks without a block
            n = int(rng.integers(40, nb * max(1, min(pr, pc)) + 2))
        elif kind == 1:          # a multiple of the block size (+- 1)
            n = nb * int(rng.integers(1, max(2, 2600 // nb))) + int(rng.integers(-1, 2))
        else:
            n = int(rng.integers(300, 2600))
        n = max(n, 12)
        delay = int(rng.choice([0, 300_000, 2_000_000]))
        kw = {"_n": n, "_nb": nb}
        if rng.random() < 0.25:
            kw["lookahead"] = False
        t0 = time.perf_counter()
        try:
            test_every_entry_point_on_the_stream_ordered_fabric(pr, pc, delay, kw)
            status = "ok"
        except Exception:  # noqa: BLE001 -- the soak reports every failing draw, not the first
            status = "FAILED"
            failures.append((i, pr, pc, n, nb, delay, kw, traceback.format_exc()[-1200:]))
        print(f"[soak {i:3d}] grid {pr}x{pc} n={n} nb={nb} delay={delay} {'no look-ahead ' if 'lookahead' in kw else ''}"
              f"{time.perf_counter() - t0:5.1f} s {status}", flush=True)
    assert not failures, failures


def test_stream_ordered_fabric_detects_a_missing_stream_dependency():
    """(round 5) mutation test of the evidence itself (tools/stream_order_mutation_probe.py): with the waits of the bulk stream, or of
    the side stream, removed from the schedule, the stream-ordered fabric must give a WRONG factorisation on the 2 x 4 grid (and the
    intact schedule the right one) -- the same mutations pass unnoticed on the host-staged fabric, i.e. over gloo"""
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_order_mutation_probe.py")], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "MUTATION PROBE OK" in r.stdout, (r.stdout[-2000:], r.stderr[-1500:])


@pytest.mark.parametrize("pr,pc,n", [(2, 4, 9000), (2, 3, 7000), (3, 2, 6200), (1, 4, 5120), (4, 1, 4600)])
def test_inverse_gram_blocks_one_launch_on_any_grid_single_process(pr, pc, n):
    """gpmp_dist_inverse_gram (round 5): the blocks of T^T T2 on a block-cyclic inverse factor in ONE launch -- staircase tile set
    (lower_only: the blocks J <= I) + the contraction start of every block row / column inside the GEMM's k loop -- for EVERY rank
    coordinate and every pair of column sets, against NumPy on the local pieces of a random lower-triangular T (ragged last blocks)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpmp_amd.num as gnp
    from gpmp_amd import _lib

    lib = _lib.load()
    nb = 1024
    rng = np.random.default_rng(pr * 10 + pc + n)
    nblk = (n + nb - 1) // nb
    bs = lambda I: min(nb, n - I * nb)                 # noqa: E731
    idx = lambda blocks: np.concatenate([np.arange(I * nb, I * nb + bs(I)) for I in blocks]) if blocks else np.zeros(0, dtype=np.int64)   # noqa: E731
    T = np.tril(rng.standard_normal((n, n)))
    for r in range(pr):
        rows = idx(list(range(r, nblk, pr)))
        for c in range(pc):
            cb = list(range(c, nblk, pc))
            Tl = gnp.as_matrix(gnp.asarray(T[np.ix_(rows, idx(cb))]), copy=True) if len(rows) and cb else None
            for c2 in sorted({c, (c + 1) % pc, (c + pc // 2) % pc}):
                cb2 = list(range(c2, nblk, pc))
                if not cb or not cb2:
                    continue
                T2l = Tl if c2 == c else (gnp.as_matrix(gnp.asarray(T[np.ix_(rows, idx(cb2))]), copy=True) if len(rows) else None)
                want = T[np.ix_(rows, idx(cb))].T @ T[np.ix_(rows, idx(cb2))] if len(rows) else np.zeros((len(idx(cb)), len(idx(cb2))))
                for lower in (0, 1):
                    M = gnp.alloc_matrix(len(idx(cb)), len(idx(cb2)))
                    M.fill_(-7.0)
                    _lib.check(lib.gpmp_dist_inverse_gram(gnp._ptr(Tl), 0 if Tl is None else gnp._ld(Tl), gnp._ptr(T2l), 0 if T2l is None else gnp._ld(T2l),
                                                          gnp._ptr(M), gnp._ld(M), n, nb, pr, pc, r, c, c2, lower, gnp._stream()), "inverse_gram")
                    got = gnp.to_np(M)
                    scale = max(1.0, np.max(np.abs(want)))
                    for li, I in enumerate(cb):
                        for lj, J in enumerate(cb2):
                            blk_g = got[li * nb: li * nb + bs(I), lj * nb: lj * nb + bs(J)]
                            blk_w = want[li * nb: li * nb + bs(I), lj * nb: lj * nb + bs(J)]
                            if lower and J > I:
                                assert np.all(blk_g == -7.0), (pr, pc, r, c, c2, I, J)          # not computed, not touched
                            else:
                                assert np.max(np.abs(blk_g - blk_w)) < 1e-11 * scale, (pr, pc, r, c, c2, I, J, lower)
