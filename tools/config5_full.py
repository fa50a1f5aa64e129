#!/usr/bin/env python3
"""BASELINE config 5 (d = 8, n = 131072, 2-D block-cyclic Cholesky) AT ITS OWN SIZE on the ONE GPU of this pool, WITH VALUES.

Two sides, run one after the other (each side alone fits the 288 GB of one MI355X; both together do not):

  single   the single-GPU product path at n = 131072: lower-triangle Gram build (K = 137 GB), gpmp_potrf_lower_async in place,
           log-determinant, NLL, (L L^T - K) on sampled rows, one zero-mean prediction with the kriging weights at a few thousand
           points, cond(K) by power / inverse iteration, REML and universal kriging with a constant mean; and ML value + gradient
           and leave-one-out (zero mean, linear mean) at the largest n the DISTRIBUTED gradient fits on a shared GPU (--grad-n).
           Results -> an .npz.
  dist     the real BlockCyclicCholesky (gpmp_amd/dist/cholesky.py: the schedule, streams and kernels of the 8-GPU run) with the
           ranks SHARING the GPU over gloo: local Gram, factor, NLL, prediction (+ weights: the backward solve), REML, universal
           kriging, then value + gradient and leave-one-out at --grad-n; compared with the .npz of ``single``.

The pool's process guard allows at most SIX processes on the card, so with one PROCESS per rank the grid is 2 x 3 (six ranks,
22.9 GB of local matrix each).  ``--threads`` runs the ranks as THREADS of one process instead (tools/thread_ranks.py: every rank has
its own HipLocalOps, streams and buffers on the shared GPU) -- that is how the 2 x 4 grid of BASELINE config 5 itself runs here:
eight ranks, 17.2 GB each -- with host-staged messages (the branch of the product code that runs over gloo) or, with
``--device-comm``, with device-resident messages on an in-process fabric that orders them by stream events only, as RCCL does.
The gradient's working set (factor + T + the neighbour's T + one block of T^T T per rank) is 3.4 x the local matrix: at
n = 131072 that is 500 GB over the ranks -- it needs the eight GPUs -- so the value + gradient is compared at --grad-n.

What is checked is what the reference computes at gpmp/num/numpy_backend.py:465-469 (cholesky_solve) and
gpmp/core/likelihood.py:18-52 (the zero-mean NLL), :92-129 (REML); gpmp/core/kriging.py:35-67 (mean / variance / weights), :105-200
(universal kriging); gpmp/core/loo.py:65-130 (leave-one-out).

    python tools/config5_full.py all                                   # both sides + comparison ("CONFIG5 FULL OK"), 2 x 3 processes over gloo
    python tools/config5_full.py all --threads --grid 2x4              # the same on the 2 x 4 grid, eight thread-ranks in one process
    python tools/config5_full.py all --threads --device-comm --grid 2x4      # ... device-resident messages, stream-ordered (RCCL's semantics)
    python tools/config5_full.py all --size-n 32768 --grad-n 16384 --m 2048     # the same at a size for the default GPU suite
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

D, NB, NOISE = 8, 1024, 1e-4


def inputs(n, m):
    """The inputs of bench.py's distributed extra: U[0,1]^8 points, seed 1234; anisotropic Matern-5/2; noise variance
    1e-4 sigma^2 on the diagonal."""
    rng = np.random.default_rng(1234)
    x = rng.random((n, D))
    z = np.sin(2 * np.pi * x[:, 0]) + x[:, 1:].sum(axis=1)
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(D) / D))))
    xt = np.random.default_rng(4321).random((m, D))
    return x, z, theta, xt


def samples(n, m):
    """Entries of L, rows of K and rows of the weights that both sides report: seeded, and always including the last rows /
    columns (the largest offsets)."""
    rng = np.random.default_rng(7)
    rows = np.unique(np.concatenate((rng.choice(n, min(n, 504), replace=False), np.arange(n - 8, n))))
    cols = np.unique(np.concatenate((rng.choice(n, min(n, 2040), replace=False), np.arange(n - 8, n), np.arange(8))))
    return rows, cols


def log(msg):
    print(f"[config5 {time.strftime('%H:%M:%S')}] {msg}", flush=True)


# ------------------------------------------------------------------------------------------------------------------
# single-GPU side
# ------------------------------------------------------------------------------------------------------------------
def single(a):
    import torch

    import gpmp_amd as gp
    import gpmp_amd.num as gnp
    from gpmp_amd.core.gradients import MLZeroMeanAnalytic
    from gpmp_amd.dist import HipLocalOps
    from gpmp_amd.kernel import MaternCovariance

    n, m = a.n, a.m
    x, z, theta, xt = inputs(n, m)
    th2 = np.concatenate(([theta[0], math.log(NOISE)], theta[1:]))
    cov = MaternCovariance(2, noise=True)
    ops = HipLocalOps()
    lib = ops.lib
    xd, zd, xtd = gnp.asarray(x), gnp.asarray(z), gnp.asarray(xt)
    out = {"n": n, "m": m}
    sec = {}

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    t0 = sync()
    K = cov.gram_lower(xd, th2)
    t1 = sync()
    F = gnp.cholesky_factor(K, overwrite=True)
    t2 = sync()
    sec["gram"], sec["potrf"] = t1 - t0, t2 - t1
    log(f"single: n={n}: K {K.numel() * 8 / 1e9:.1f} GB; gram {sec['gram']:.2f} s, potrf {sec['potrf']:.2f} s = "
        f"{n ** 3 / 3 / sec['potrf'] / 1e12:.1f} TFLOP/s")
    L = F.L
    out["logdet"] = F.logdet()
    w = F.solve_lower(zd)
    alpha = F.solve_lower(w, trans=True)
    out["nll"] = 0.5 * (n * math.log(2 * math.pi) + out["logdet"] + float((w * w).sum().item()))
    sec["nll_tail"] = sync() - t2
    log(f"single: logdet {out['logdet']!r} nll {out['nll']!r}")

    # ---- (L L^T - K) on sampled rows; K rows are rebuilt from the points (K itself was overwritten by its factor)
    t3 = sync()
    rows_np, cols_np = samples(n, m)
    rows = torch.as_tensor(rows_np, device=L.device)
    cols = torch.as_tensor(cols_np, device=L.device)
    lib.gpmp_tril(gnp._ptr(L), n, gnp._ld(L), gnp._stream())
    Lr = gnp.as_matrix(L[rows], copy=True)
    R = gnp.as_matrix(cov(xd[rows].contiguous(), xd, th2), copy=True)            # cross-covariance path: no diagonal term
    R[torch.arange(len(rows_np), device=L.device), rows] += NOISE
    kmax = float(R.abs().max())
    ops.gemm_nt_sub(R, Lr, L)                                                     # K[rows] - L[rows] L^T
    out["residual_rel"] = float(R.abs().max()) / kmax
    out["L_rows"], out["L_cols"] = rows_np, cols_np
    out["L_sample"] = gnp.to_np(Lr[:, cols])
    del R, Lr
    sec["residual"] = sync() - t3
    log(f"single: max |K - L L^T| / max |K| on {len(rows_np)} sampled rows: {out['residual_rel']:.2e}")

    # ---- cond(K): power iteration on K v = L (L^T v), inverse iteration through the factor
    t4 = sync()
    v = gnp.asarray(np.random.default_rng(3).standard_normal(n))
    lam_max = lam_min_inv = 0.0
    for _ in range(30):
        v = v / torch.linalg.vector_norm(v)
        u = gnp.matmul(L, v.reshape(-1, 1), ta=True)
        v = gnp.matmul(L, u).reshape(-1)
        lam_max = float(torch.linalg.vector_norm(v))
    v = gnp.asarray(np.random.default_rng(4).standard_normal(n))
    for _ in range(30):
        v = v / torch.linalg.vector_norm(v)
        v = F.solve(v)
        lam_min_inv = float(torch.linalg.vector_norm(v))
    out["cond"] = lam_max * lam_min_inv
    sec["cond"] = sync() - t4
    log(f"single: cond(K) ~ {out['cond']:.3e} (lambda_max {lam_max:.4e}, lambda_min {1.0 / lam_min_inv:.4e})")

    # ---- one zero-mean prediction with its weights: V = L^-1 Kit, mean = V^T w, var = sigma^2 - colsumsq(V), lambda = L^-T V
    t5 = sync()
    Kit = cov(xd, xtd, th2)
    V = F.solve_lower(Kit, overwrite=True)
    dots = gnp.coldots(V, w.reshape(-1, 1))
    out["mean"] = gnp.to_np(dots[0])
    out["var"] = math.exp(theta[0]) - gnp.to_np(dots[1])
    lam = F.solve_lower(V, trans=True, overwrite=True)
    out["lam_sample"] = gnp.to_np(lam[rows])
    out["lam_max"] = float(lam.abs().max())
    del lam, V, Kit
    sec["predict"] = sync() - t5
    log(f"single: prediction at {m} points with weights: {sec['predict']:.2f} s")
    # ---- REML with a constant mean on the same factor (gpmp/core/likelihood.py:92-129 through the Schur identity)
    from gpmp_amd.core.linalg import MeanSpace

    ms = MeanSpace(F, zd, gnp.ones((n, 1)))
    out["reml"] = 0.5 * ((n - 1) * math.log(2 * math.pi) + ms.logdet_contrast() + ms.quad())
    del F, L, K, w, alpha, ms
    torch.cuda.empty_cache()
    # ---- universal kriging (constant mean) at the first points: the single-GPU Model itself (builds and factors K again)
    t5u = sync()
    mu_pts = min(m, a.m_uk)
    const = lambda xx, prm: gnp.ones((xx.shape[0], 1))       # noqa: E731
    uk = gp.Model(const, cov, None, th2, "linear_predictor")
    out["uk_mean"], out["uk_var"] = uk.predict(x, z, xt[:mu_pts], zero_neg_variances=False)
    del uk
    torch.cuda.empty_cache()
    sec["uk_predict_incl_second_factorisation"] = sync() - t5u
    log(f"single: REML {out['reml']!r}; universal kriging at {mu_pts} points (Model.predict, second factorisation included): "
        f"{sec['uk_predict_incl_second_factorisation']:.2f} s")

    # ---- ML value + gradient at the size the distributed gradient fits on the shared GPU
    if a.grad_n:
        t6 = sync()
        gx, gz = x[: a.grad_n], z[: a.grad_n]
        model = gp.Model(None, cov, None, th2, "zero")
        crit = MLZeroMeanAnalytic(model)
        val, state = crit.value_and_state(th2, gnp.asarray(gx), gnp.asarray(gz))
        out["grad_value"], out["grad"] = val, crit.gradient_from_state(state)
        sec["value_and_grad"] = sync() - t6
        log(f"single: ML value + gradient at n={a.grad_n}: {sec['value_and_grad']:.2f} s; value {val!r}")
        del crit, state
        torch.cuda.empty_cache()
        # leave-one-out at the same size: zero mean, and with a linear mean (1, x_1 .. x_8) through the Schur form (gpmp/core/loo.py:65-130)
        from gpmp_amd.core.loo import loo as loo_single

        t7 = sync()
        lin = lambda xx, prm: gnp.hstack((gnp.ones((xx.shape[0], 1)), gnp.asarray(xx)))      # noqa: E731
        for tag, mdl in (("zero", gp.Model(None, cov, None, th2, "zero")), ("lin", gp.Model(lin, cov, None, th2, "linear_predictor"))):
            zl, s2, el = loo_single(mdl, gnp.asarray(gx), gnp.asarray(gz))
            out[f"loo_{tag}_z"], out[f"loo_{tag}_s2"], out[f"loo_{tag}_e"] = gnp.to_np(zl), gnp.to_np(s2), gnp.to_np(el)
            del zl, s2, el
            torch.cuda.empty_cache()
        sec["loo_two_mean_types"] = sync() - t7
        log(f"single: leave-one-out at n={a.grad_n}, zero mean + linear mean (two factorisations + inverse factors): {sec['loo_two_mean_types']:.2f} s")
    out["seconds"] = json.dumps(sec)
    np.savez(a.out, **out)
    log(f"single: wrote {a.out}")


# ------------------------------------------------------------------------------------------------------------------
# distributed side
# ------------------------------------------------------------------------------------------------------------------
def dist_worker(rank, world, port, a):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpmp_amd.dist import BlockCyclicCholesky

        def exchange(obj):
            parts = [None] * world
            dist.all_gather_object(parts, obj)
            return parts

        dist_body(rank, world, a, BlockCyclicCholesky, exchange)
    finally:
        dist.destroy_process_group()


def dist_body(rank, world, a, Cholesky, exchange):
    """One rank of the distributed side (a process over gloo, or a thread of the one process under --threads).
    ``exchange(obj)`` -> the objects of all ranks, in rank order."""
    import torch
    import torch.distributed as dist

    import gpmp_amd.num as gnp
    from gpmp_amd.dist import HipLocalOps, ProcessGrid
    from gpmp_amd.kernel import MaternCovariance

    pr, pc = (int(v) for v in a.grid.split("x"))
    n, m = a.n, a.m
    x, z, theta, xt = inputs(n, m)
    cov = MaternCovariance(2)
    grid = ProcessGrid(pr, pc)
    xd = gnp.asarray(x)
    sec = {}

    def tick():
        torch.cuda.synchronize()
        dist.barrier()
        return time.perf_counter()

    def say(msg):
        if rank == 0:
            log("dist: " + msg)

    ch = Cholesky(grid, n, nb=NB, ops=HipLocalOps(), transport=a.transport, profile=True)
    if a.device_comm:
        ch.backend = "nccl"        # the device-resident communication branches (what runs under RCCL), moved by gloo
    t0 = tick()
    ch.build_local_gram(cov, xd, theta, NOISE)
    t1 = tick()
    sec["gram"] = t1 - t0
    say(f"local Gram {tuple(ch.A.shape)} = {ch.A.numel() * 8 / 1e9:.1f} GB per rank: {sec['gram']:.2f} s")
    info = ch.factor()
    phases = ch.phase_times()
    t2 = tick()
    sec["factor"] = t2 - t1
    say(f"factor: info {info}, {sec['factor']:.1f} s = {n ** 3 / 3 / sec['factor'] / 1e12:.1f} TFLOP/s aggregate over the shared GPU {' (device-resident messages, stream-ordered)' if getattr(a, 'stream_ordered', False) else ' (messages through host memory)'}; "
        f"rank 0 phases (ms, summed HIP-event spans): { {k: round(v) for k, v in phases.items()} }")
    nll = ch.negative_log_likelihood(z)
    logdet = ch.logdet()
    t3 = tick()
    sec["nll"] = t3 - t2
    say(f"nll {nll!r} logdet {logdet!r}: {sec['nll']:.1f} s")
    # ---- sampled entries of the factor
    rows_np, cols_np = samples(n, m)
    ri, ci = ch.global_row_index(), ch.global_col_index()
    rsel, csel = np.nonzero(np.isin(ri, rows_np))[0], np.nonzero(np.isin(ci, cols_np))[0]
    Lloc = gnp.to_np(ch.A[torch.as_tensor(rsel, device=ch.A.device)][:, torch.as_tensor(csel, device=ch.A.device)]) if len(rsel) and len(csel) else np.zeros((0, 0))
    # ---- prediction with weights
    mean, var, (j0, j1), lam = ch.predict(cov, xd, z, xt, theta, return_lambdas=True)
    lam_rows = gnp.to_np(lam[torch.as_tensor(rsel, device=lam.device)]) if len(rsel) and j1 > j0 else np.zeros((len(rsel), j1 - j0))
    del lam
    t4 = tick()
    sec["predict_with_weights"] = t4 - t3
    say(f"prediction at {m} points + weights (forward + backward many-RHS solves): {sec['predict_with_weights']:.1f} s")
    # ---- REML and universal kriging (constant mean) on the same distributed factor
    mu_pts = min(m, a.m_uk)
    reml = ch.negative_log_restricted_likelihood(z, np.ones((n, 1)))
    uk_mean, uk_var, (u0, u1) = ch.predict(cov, xd, z, xt[:mu_pts], theta, P=np.ones((n, 1)), Pt=np.ones((mu_pts, 1)))
    t4 = tick()
    sec["reml_and_universal_kriging"] = t4 - t3 - sec["predict_with_weights"]
    say(f"REML {reml!r}; universal kriging at {mu_pts} points: {sec['reml_and_universal_kriging']:.1f} s")
    received = ch.bytes_received
    local_shape = tuple(ch.A.shape)
    del ch
    torch.cuda.empty_cache()
    # ---- value + gradient at grad_n
    val = grad = None
    loo_parts = {}
    if a.grad_n:
        gn = a.grad_n
        th2 = np.concatenate(([theta[0], math.log(NOISE)], theta[1:]))
        chg = Cholesky(grid, gn, nb=NB, ops=HipLocalOps(), transport=a.transport)
        if a.device_comm:
            chg.backend = "nccl"
        chg.build_local_gram(cov, gnp.asarray(x[:gn]), theta, NOISE)
        ginfo = chg.factor()
        t5 = tick()
        val, grad = chg.value_and_grad(x[:gn], z[:gn], th2, 2, noise=True)
        t6 = tick()
        sec["grad_factor"], sec["value_and_grad"] = t5 - t4, t6 - t5
        say(f"n={gn}: factor (info {ginfo}) {sec['grad_factor']:.1f} s, ML value + gradient {sec['value_and_grad']:.1f} s")
        # leave-one-out on the same distributed factor: zero mean, linear mean (each forms T = L^-1 in the block-cyclic layout)
        Plin = np.hstack((np.ones((gn, 1)), x[:gn]))
        loo_parts = {"zero": chg.loo(z[:gn]), "lin": chg.loo(z[:gn], P=Plin)}
        t7 = tick()
        sec["loo_two_mean_types"] = t7 - t6
        say(f"n={gn}: leave-one-out, zero mean + linear mean (two distributed inverse factors): {sec['loo_two_mean_types']:.1f} s")
        received += chg.bytes_received
        del chg
    import threading

    rec = {"rank": rank, "pid": os.getpid(), "thread": threading.get_ident() if a.threads else None, "coords": (grid.r, grid.c), "local_matrix": local_shape, "GB_received": received / 1e9,
           "device": getattr(a, "device_name", None) or torch.cuda.get_device_name(torch.cuda.current_device()),
           "peak_GB_allocated": torch.cuda.max_memory_allocated(torch.cuda.current_device()) / 1e9}
    parts = exchange((rec, ri[rsel], ci[csel], Lloc, (j0, j1), grid.r, mean, var, lam_rows, (u0, u1), uk_mean, uk_var, loo_parts if grid.r == 0 else {}))
    if rank == 0:
        Ls = np.full((len(rows_np), len(cols_np)), np.nan)
        zpm, zpv = np.full(m, np.nan), np.full(m, np.nan)
        lam_s = np.full((len(rows_np), m), np.nan)
        ukm, ukv = np.full(mu_pts, np.nan), np.full(mu_pts, np.nan)
        rpos = {int(g): i for i, g in enumerate(rows_np)}
        cpos = {int(g): i for i, g in enumerate(cols_np)}
        recs = []
        loo_full = {f"loo_{tag}_{w}": np.full(a.grad_n, np.nan) for tag in ("zero", "lin") for w in ("z", "s2", "e")} if a.grad_n else {}
        for (rc, gr, gc, blk, (b0, b1), r_, mu, vv, lr, (c0, c1), um, uv, lp) in parts:
            for tag, (zl, s2l, el, idx) in lp.items():              # one rank per process column reports its column set
                loo_full[f"loo_{tag}_z"][idx], loo_full[f"loo_{tag}_s2"][idx], loo_full[f"loo_{tag}_e"][idx] = zl, s2l, el
            if c1 > c0:
                ukm[c0:c1], ukv[c0:c1] = um, uv
            recs.append(rc)
            ir, ic = [rpos[int(g)] for g in gr], [cpos[int(g)] for g in gc]
            if len(ir) and len(ic):
                Ls[np.ix_(ir, ic)] = blk
            if b1 > b0:
                zpm[b0:b1], zpv[b0:b1] = mu, vv
                if len(ir):
                    lam_s[np.ix_(ir, np.arange(b0, b1))] = lr
        np.savez(a.dist_out, info=info, nll=nll, logdet=logdet, reml=reml, uk_mean=ukm, uk_var=ukv, L_sample=Ls, mean=zpm, var=zpv, lam_sample=lam_s,
                 grad_value=np.nan if val is None else val, grad=np.zeros(0) if grad is None else grad,
                 **loo_full,
                 seconds=json.dumps(sec), phases=json.dumps(phases), ranks=json.dumps(recs))
        for rc in recs:
            who = f"pid {rc['pid']}" + (f" thread {rc['thread']}" if rc.get("thread") else "")
            log(f"dist: rank {rc['rank']} {who} coords {tuple(rc['coords'])} local matrix {tuple(rc['local_matrix'])} "
                f"received {rc['GB_received']:.1f} GB; peak allocated {rc['peak_GB_allocated']:.1f} GB"
                f"{' (all thread-ranks of the process together)' if rc.get('thread') else ''} on {rc['device']}")


def run_dist_threads(a):
    """The distributed side with the ranks as THREADS of this process (tools/thread_ranks.py): the grid of eight ranks the process
    guard denies to processes.  ``--device-comm``: the device-resident branch of the product code on the stream-ordered in-process
    fabric (RCCL's stream semantics, no host synchronisation); otherwise the host-staged branch on host copies between the threads."""
    import torch

    import gpmp_amd.num  # noqa: F401 -- the library is loaded and its signatures declared ONCE, before the rank threads start
    from gpmp_amd import _lib
    from tools import thread_ranks

    pr, pc = (int(v) for v in a.grid.split("x"))
    world = pr * pc
    if a.transport != "bcast" and not a.device_comm:
        raise SystemExit("--threads without --device-comm: the in-process group has no point-to-point operation: bcast transport only")
    _lib.load()
    torch.cuda.init()
    a.device_name = torch.cuda.get_device_name(0)           # (asked here: torch's device-property cache is filled by the thread that initialised it)
    board = [None] * world

    def body(rank, world_, fabric, classes):
        Cholesky = classes[1] if a.device_comm else classes[0]
        a_rank = argparse.Namespace(**vars(a))
        a_rank.stream_ordered = bool(a.device_comm)
        a_rank.device_comm = False                         # (the class carries the branch; dist_body's own switch is for the gloo form)

        def exchange(obj):
            return fabric.allgather(rank, obj)

        dist_body(rank, world_, a_rank, Cholesky, exchange)

    errors = thread_ranks.run(world, body, limit_s=a.limit)
    if errors:
        raise SystemExit("distributed side failed:\n" + errors[0])


def run_dist(a):
    import torch.multiprocessing as mp

    pr, pc = (int(v) for v in a.grid.split("x"))
    world = pr * pc
    if world > 6:
        raise SystemExit("the pool's process guard allows at most 6 processes on the GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.spawn(dist_worker, args=(world, port, a), nprocs=world, join=False)
    deadline = time.monotonic() + a.limit
    while not ctx.join(timeout=5.0):
        if time.monotonic() > deadline:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            for p in ctx.processes:
                p.join(10)
            raise SystemExit(f"distributed side still running after {a.limit:.0f} s: killed")


# ------------------------------------------------------------------------------------------------------------------
def compare(a):
    s, d = np.load(a.out), np.load(a.dist_out)
    cond = float(s["cond"])
    cs = max(1.0, cond / 1e6)
    x, z, theta, xt = inputs(a.n, a.m)
    zs = float(np.max(np.abs(z)))
    lower = s["L_rows"][:, None] >= s["L_cols"][None, :]
    Ld, Lsg = np.where(lower, d["L_sample"], 0.0), np.where(lower, s["L_sample"], 0.0)
    errs = {
        "info": int(d["info"]),
        "cond": cond,
        "single_residual_rel": float(s["residual_rel"]),
        "logdet_rel": abs(float(d["logdet"]) - float(s["logdet"])) / abs(float(s["logdet"])),
        "nll_rel": abs(float(d["nll"]) - float(s["nll"])) / abs(float(s["nll"])),
        "L_entries_rel": float(np.max(np.abs(Ld - Lsg)) / np.max(np.abs(Lsg))),
        "L_entries_compared": int(lower.sum()),
        "mean_abs": float(np.max(np.abs(d["mean"] - s["mean"]))),
        "var_abs": float(np.max(np.abs(d["var"] - s["var"]))),
        "lambda_rel": float(np.max(np.abs(d["lam_sample"] - s["lam_sample"])) / float(s["lam_max"])),
        "reml_rel": abs(float(d["reml"]) - float(s["reml"])) / abs(float(s["reml"])),
        "uk_mean_abs": float(np.max(np.abs(d["uk_mean"] - s["uk_mean"]))),
        "uk_var_abs": float(np.max(np.abs(d["uk_var"] - s["uk_var"]))),
    }
    tol = {"single_residual_rel": 1e-12, "logdet_rel": 1e-12 * cs, "nll_rel": 1e-12 * cs, "L_entries_rel": 1e-10 * cs,
           "mean_abs": 1e-10 * cs * zs, "var_abs": 1e-10 * cs, "lambda_rel": 1e-7,
           "reml_rel": 1e-12 * cs, "uk_mean_abs": 1e-10 * cs * zs, "uk_var_abs": 1e-10 * cs}
    if a.grad_n:
        g1, g0 = d["grad"], s["grad"]
        errs["grad_value_rel"] = abs(float(d["grad_value"]) - float(s["grad_value"])) / abs(float(s["grad_value"]))
        errs["grad_rel"] = float(np.max(np.abs(g1 - g0)) / np.max(np.abs(g0)))
        tol.update(grad_value_rel=1e-11 * cs, grad_rel=1e-7)
        # leave-one-out (rel 1e-8 at cond <= 1e6): predictions on the scale of z, variances and errors relative to their largest value
        for tag in ("zero", "lin"):
            errs[f"loo_{tag}_z_abs"] = float(np.max(np.abs(d[f"loo_{tag}_z"] - s[f"loo_{tag}_z"])))
            errs[f"loo_{tag}_s2_rel"] = float(np.max(np.abs(d[f"loo_{tag}_s2"] - s[f"loo_{tag}_s2"])) / np.max(np.abs(s[f"loo_{tag}_s2"])))
            errs[f"loo_{tag}_e_rel"] = float(np.max(np.abs(d[f"loo_{tag}_e"] - s[f"loo_{tag}_e"])) / np.max(np.abs(s[f"loo_{tag}_e"])))
            tol.update({f"loo_{tag}_z_abs": 1e-8 * cs * zs, f"loo_{tag}_s2_rel": 1e-8 * cs, f"loo_{tag}_e_rel": 1e-8 * cs})
    ok = errs["info"] == 0 and bool(np.isfinite(d["mean"]).all() and np.isfinite(d["lam_sample"]).all()) and all(errs[k] <= t for k, t in tol.items())
    log("single-GPU seconds: " + str(s["seconds"]))
    log(f"distributed seconds (max-synchronised phases, ranks sharing ONE GPU, {'thread-ranks' if a.threads else 'over gloo'}): " + str(d["seconds"]))
    log("deviations (block-cyclic vs single GPU): " + json.dumps(errs))
    log("tolerances: " + json.dumps(tol))
    how = ("thread-ranks of one process, " + ("device-resident messages ordered by stream events only (RCCL's semantics, in process)" if a.device_comm
                                              else "host-staged messages copied between the threads")) if a.threads else "one process per rank over gloo"
    print(f"CONFIG5 FULL {'OK' if ok else 'FAILED'}: n={a.n} grid {a.grid} block {NB} transport {a.transport}"
          f"{' device-resident comm' if a.device_comm else ''}, {how}; gradient at n={a.grad_n}", flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=("single", "dist", "compare", "all"))
    ap.add_argument("--size-n", dest="n", type=int, default=131072)
    ap.add_argument("--m", type=int, default=6144)
    ap.add_argument("--grad-n", type=int, default=73728)
    ap.add_argument("--m-uk", type=int, default=1024, help="prediction points of the universal-kriging (constant mean) comparison")
    ap.add_argument("--grid", default="2x3")
    ap.add_argument("--transport", default="bcast")
    ap.add_argument("--device-comm", action="store_true")
    ap.add_argument("--threads", action="store_true", help="ranks as threads of ONE process (in-process 'threaded' process group): the 2 x 4 grid")
    ap.add_argument("--limit", type=float, default=1500.0, help="seconds allowed to the distributed side before its workers are killed")
    ap.add_argument("--out", default="/tmp/config5_single.npz")
    ap.add_argument("--dist-out", default="/tmp/config5_dist.npz")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    if a.what == "single":
        return single(a)
    if a.what == "dist":
        return run_dist_threads(a) if a.threads else run_dist(a)
    if a.what == "compare":
        return compare(a)
    # all: the single-GPU side in a child process of its own (its 137 GB are gone when it ends), then the ranks, then the comparison
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [sys.executable, os.path.abspath(__file__), "single"] + sys.argv[2:]
    r = subprocess.run(args, env=env, timeout=a.limit)
    if r.returncode != 0:
        raise SystemExit(f"single-GPU side failed ({r.returncode})")
    if a.threads:
        # (in a child as well: the thread-ranks initialise the GPU in their process; this one stays a GPU-free coordinator)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "dist"] + sys.argv[2:], env=env, timeout=a.limit + 60)
        if r.returncode != 0:
            raise SystemExit(f"distributed side failed ({r.returncode})")
    else:
        run_dist(a)
    return compare(a)


if __name__ == "__main__":
    sys.exit(main() or 0)
