#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X exact-GP inner loop (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one pass of the hot path over one batch of synthetic inputs already resident in HBM:
    Model.predict(xi, zi, xt)          Gram build -> Cholesky -> solve -> posterior mean / variance
  + Model.negative_log_likelihood_zero_mean(theta, xi, zi)   Gram build -> Cholesky -> solve -> NLL
Workload (BASELINE.json configs[2], the configuration the metric is quoted on): d = 8, Matern-5/2
anisotropic, n = 32768 observations, m = 50000 prediction points per GPU, fp64.
Multi-GPU (weak scaling): the prediction set shards over ranks (m points per rank), the observations
are replicated and every rank factors K itself -- no collective on the data path.

value = (N * m) / (max over ranks of the time of K steps / K)   [points/s].
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet fp64 matrix peak; tools/mfma_f64_probe2.hip measures 77 (98 %)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def synth(n, m, d, rank):
    """default_rng(1234); U[0,1]^d inputs; z = sin(2 pi x0) + sum_j x_j; rho_j = 0.5 (1 + j/d)."""
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d))
    zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1)
    xt = np.random.default_rng(4321 + rank).random((m, d))
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    return xi, zi, xt, theta


def pmc_traffic_per_launch(kernel_substr):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*/bench_v*_pmc_{fetch,write}_size_by_kernel.csv, latest; FETCH_SIZE / WRITE_SIZE are in KB and, on gfx950,
    FETCH_SIZE reports half of a wide streaming read -- MI355X_MICROARCH.md, HBM section).  None if absent."""
    import csv

    import glob

    tot = 0.0
    n_disp = None
    pmc_traffic_per_launch.source = None
    for name, factor in (("fetch", 2.0), ("write", 1.0)):
        import re

        def version(path):      # (round, profile version), numerically: v10 comes after v9
            m = re.search(r"r(\d+)[/\\]bench_v(\d+)_pmc", path)
            return (int(m.group(1)), int(m.group(2))) if m else (0, 0)

        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"bench_v*_pmc_{name}_size_by_kernel.csv")), key=version)
        if not found:
            return None
        path = found[-1]   # the latest committed pass
        pmc_traffic_per_launch.source = os.path.relpath(os.path.dirname(path), ROOT) + "/" + re.sub(r"_(fetch|write)_", "_{fetch,write}_", os.path.basename(path))
        for row in csv.DictReader(open(path)):
            if kernel_substr in row["kernel"]:
                tot += factor * 1024.0 * float(row["per_dispatch_KB_raw"])
                n_disp = int(row["dispatches"])
    return tot if n_disp else None


def _host_threads():
    try:
        from threadpoolctl import threadpool_info

        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(n, m, d, threads, m_sample=512, n_sample=8192):
    """The oracle (NumPy / SciPy restatement of the reference's NumPy backend: cdist -> Matern ufuncs -> cholesky ->
    2 x solve_triangular -> einsum) on the host cores, on a BOUNDED sample of the headline workload.  The reference's Gram
    build is single-threaded (SciPy cdist + ~7 full-size NumPy temporaries: 1.6 s at n = 4096, i.e. ~100 s at n = 32768,
    twice per step), so the full step cannot be run inside a benchmark; what is run, ~20-30 s of CPU work:
      * Cholesky at the FULL n (LAPACK dpotrf through numpy.linalg.cholesky, all BLAS threads) -- also the host potrf figure;
      * Gram(xi, xi), Gram(xi, xt_s), the two triangular solves + reductions for m_sample points and the NLL solves at
        n_sample = min(n, 8192) observations of the same synthetic set.
    The step time at the full size is then assembled from these with their exact complexities -- Gram(xi,xi) ~ n^2,
    Gram(xi,xt) ~ n m, solves ~ n^2 m, NLL solves ~ n^2:
        T = 2 (Gram_ii (n/n_s)^2 + Cholesky(n)) + NLL_tail (n/n_s)^2 + m/m_s (Gram_it (n/n_s) + Solve (n/n_s)^2)
    and value = m / T.  Every term is a measured oracle call; only the scaling is arithmetic, and it is stated in `sample`."""
    from scipy.linalg import solve_triangular

    from oracle import gp_oracle as orc

    ns = min(n, n_sample)
    xi, zi, xt, theta = synth(n, m, d, 0)
    xi_s, zi_s, xs = xi[:ns], zi[:ns], xt[:m_sample]
    t = {}
    np.linalg.cholesky(orc.maternp_covariance(xi[:512], None, 2, theta))    # BLAS thread pool / page-in warm-up, untimed

    def tick(name, fn):
        t0 = time.perf_counter()
        out = fn()
        t[name] = time.perf_counter() - t0
        return out

    K = tick("gram_ii", lambda: orc.maternp_covariance(xi_s, None, 2, theta))               # kriging.py:59, likelihood.py:43
    L = tick("cholesky_ns", lambda: np.linalg.cholesky(K))                                   # numpy_backend.py:466
    del K
    Kit = tick("gram_it", lambda: orc.maternp_covariance(xi_s, xs, 2, theta))               # kriging.py:60

    def solves():                                                                            # numpy_backend.py:467-468, kriging.py:193-194, model.py:298
        y = solve_triangular(L, Kit, lower=True)
        lam = solve_triangular(L.T, y, lower=False)
        var = orc.maternp_covariance(xs, None, 2, theta, True) - np.einsum("i..., i...", lam, Kit)
        return np.einsum("i..., i...", lam, zi_s), var

    tick("solve_sample", solves)

    def nll_tail():                                                                          # likelihood.py:46-51
        a = solve_triangular(L.T, solve_triangular(L, zi_s, lower=True), lower=False)
        return 0.5 * (ns * math.log(2 * math.pi) + 2.0 * np.sum(np.log(np.diag(L))) + zi_s @ a)

    tick("nll_tail", nll_tail)
    del L, Kit
    if ns < n:
        # dpotrf at the full n: its time does not depend on the entries, so a cheap SPD matrix stands in for K (8n^2 bytes)
        S = np.full((n, n), 0.5)
        S[np.diag_indices(n)] = float(n)
        tick("cholesky", lambda: np.linalg.cholesky(S))
        del S
    else:
        t["cholesky"] = t["cholesky_ns"]
    r = n / ns
    per_point = (t["gram_it"] * r + t["solve_sample"] * r * r) / m_sample
    step = 2.0 * (t["gram_ii"] * r * r + t["cholesky"]) + t["nll_tail"] * r * r + m * per_point
    return {"value": m / step, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": f"oracle (SciPy cdist + Matern ufuncs + LAPACK), d={d}: Cholesky at the full n={n} ({t['cholesky']:.1f} s); Gram(xi,xi) "
                      f"({t['gram_ii']:.1f} s), Gram(xi,xt), 2 triangular solves + reductions for {m_sample} of the {m} points and the NLL "
                      f"solves at n_s={ns}; step time assembled with the exact complexities (Gram_ii, solves, NLL ~ (n/n_s)^2; Gram_it ~ n/n_s; "
                      f"per-point part x m/{m_sample}): {step:.0f} s per predict+NLL step; CPU work done {sum(t.values()):.0f} s; "
                      f"BLAS threads={threads} (cdist and the ufuncs are single-threaded)",
            "host_potrf": {"n": n, "s": t["cholesky"], "tflops": n ** 3 / 3.0 / t["cholesky"] / 1e12},
            "phases_s": {k_: round(v_, 3) for k_, v_ in t.items()}}


def config2_extra(model, d, threads, with_cpu):
    """BASELINE.json configs[1]: d = 8, n = 4096 / m = 10000 -- one predict + NLL step on the GPU beside the oracle at the
    SAME size on the host cores (full run, a few seconds)."""
    import torch

    import gpmp_amd.num as gnp

    n, m = 4096, 10000
    xi_h, zi_h, xt_h, theta = synth(n, m, d, 0)
    xi, zi, xt = gnp.asarray(xi_h), gnp.asarray(zi_h), gnp.asarray(xt_h)

    def step():
        model.predict(xi, zi, xt, convert_in=False, convert_out=False)
        return model.negative_log_likelihood_zero_mean(theta, xi, zi)

    step()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(5):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    flops = 2.0 * n ** 3 / 3.0 + float(n) * n * m      # two factorisations + the one solve the prediction needs
    out = {"n": n, "m": m, "d": d, "ms_per_step": 1e3 * best, "points_per_s": m / best, "mfma_flops": flops,
           "frac_of_fp64_mfma_peak": flops / best / 1e12 / FP64_MFMA_PEAK_TFLOPS}
    if with_cpu:
        from oracle import gp_oracle as orc

        kern = lambda x, y, t, pairwise=False: orc.maternp_covariance(x, y, 2, t, pairwise)  # noqa: E731
        om = orc.OracleModel(None, kern, None, theta, "zero")
        t0 = time.perf_counter()
        orc.predict(om, xi_h, zi_h, xt_h)
        orc.negative_log_likelihood_zero_mean(om, theta, xi_h, zi_h)
        dt = time.perf_counter() - t0
        out["cpu_same_size"] = {"s_per_step": dt, "points_per_s": m / dt, "cores": threads, "kind": "port (oracle, full run)"}
    return out


def config4_extra(threads, with_cpu):
    """BASELINE.json configs[3]: REML fit at n = 16384, d = 20 -- one criterion value + analytic gradient evaluation on the
    GPU (what each of the 50 L-BFGS evaluations costs), beside the two CPU routes SURVEY 8(d) names at n = 4096:
    (i) the NumPy backend's finite-difference route = (d + 2) criterion values per value + gradient
    (numpy_backend.py:333, parameter_selection.py:248-260), (ii) an analytic-gradient CPU route (potrf + inverse + traces)."""
    import torch

    import gpmp_amd as gp
    import gpmp_amd.num as gnp
    from gpmp_amd.core.gradients import REMLAnalytic
    from gpmp_amd.kernel import MaternCovariance

    n, d = 16384, 20
    rng = np.random.default_rng(1234)
    x = rng.random((n, d))
    z = np.sin(2 * np.pi * x[:, 0]) + x[:, 1:].sum(axis=1)
    theta = np.concatenate(([0.0], -np.log(0.5 + np.arange(d) / (d - 1.0))))      # rho_j in [0.5, 1.5]
    xd, zd = gnp.asarray(x), gnp.asarray(z)
    crit = REMLAnalytic(gp.Model(lambda a, p: gnp.ones((a.shape[0], 1)), MaternCovariance(2), None, theta))

    def evaluate():
        v, st = crit.value_and_state(theta, xd, zd)
        return v, crit.gradient_from_state(st)

    evaluate()
    torch.cuda.synchronize()
    best, best_v = float("inf"), float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        v, st = crit.value_and_state(theta, xd, zd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g = crit.gradient_from_state(st)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best, best_v = min(best, t2 - t0), min(best_v, t1 - t0)
        del st
    flops = float(n) ** 3                                 # potrf + trtri + lauum, n^3 / 3 each
    out = {"n": n, "d": d, "criterion": "REML, constant mean", "ms_per_value_and_gradient": 1e3 * best, "ms_value_only": 1e3 * best_v,
           "s_per_50_evaluations": 50 * best, "mfma_flops": flops, "frac_of_fp64_mfma_peak": flops / best / 1e12 / FP64_MFMA_PEAK_TFLOPS,
           "value": float(v), "grad_norm": float(np.linalg.norm(g))}
    del xd, zd
    torch.cuda.empty_cache()
    if with_cpu:
        from oracle import gp_oracle as orc

        nc = 4096
        xc, zc = x[:nc], z[:nc]
        P = np.ones((nc, 1))
        kern = lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise)  # noqa: E731
        om = orc.OracleModel(lambda a, p: np.ones((a.shape[0], 1)), kern, None, theta, "linear_predictor")
        t0 = time.perf_counter()
        orc.negative_log_restricted_likelihood(om, theta, xc, zc)
        t_val = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.reml_value_and_grad(xc, zc, P, 2, theta)
        t_ana = time.perf_counter() - t0
        # the GPU at the same size, for a like-for-like ratio
        xg, zg = gnp.asarray(xc), gnp.asarray(zc)
        crit.value_and_state(theta, xg, zg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, st = crit.value_and_state(theta, xg, zg)
        crit.gradient_from_state(st)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        out["cpu_n4096"] = {"cores": threads, "kind": "port (oracle)",
                            "reml_value_s": t_val,
                            "fd_route_s_per_value_and_gradient": (d + 2) * t_val,
                            "fd_route_note": f"(d + 2) = {d + 2} criterion values per value + gradient (SciPy finite differences of the NumPy backend)",
                            "analytic_route_s_per_value_and_gradient": t_ana,
                            "gpu_same_size_ms": 1e3 * t_gpu}
    return out


DIST_N = {2: 65536, 4: 90112, 8: 131072}     # about 17 GB of local matrix per GPU; 8 GPUs = BASELINE.json configs[4]


def dist_potrf_extra(world, rank, res):
    """OUTSIDE the timed region, N > 1 only: the 2-D block-cyclic Cholesky (+ NLL) of BASELINE.json configs[4]
    (n = 131072 on the 2 x 4 grid of 8 GPUs; n scaled to the same memory per GPU on 2 / 4 GPUs) with RCCL panel
    broadcasts, once per transport.  Fills ``res`` progressively so that a watchdog can still report what finished."""
    import torch
    import torch.distributed as dist

    import gpmp_amd.num as gnp
    from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
    from gpmp_amd.kernel import MaternCovariance

    pr, pc = ProcessGrid.default_shape(world)
    n = DIST_N.get(world, max(1024, int(46000 * math.sqrt(world)) // 1024 * 1024))
    d, nb = 8, 1024
    res.update({"n": n, "d": d, "grid": f"{pr}x{pc}", "block": nb, "noise_variance": 1e-4,
                "note": "outside the timed region; flops = n^3/3 over wall time of factor() (max over ranks)"})
    rng = np.random.default_rng(1234)
    x = rng.random((n, d))
    z = np.sin(2 * np.pi * x[:, 0]) + x[:, 1:].sum(axis=1)
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    xd = gnp.asarray(x)
    grid = ProcessGrid(pr, pc)
    torch.cuda.empty_cache()

    def tmax(v):
        t = torch.tensor([v], dtype=torch.float64, device=gnp._dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- values first: a small problem through the same grid / collectives against the single-GPU path on rank 0
    n_chk = 8192
    res["phase"] = "check_n8192: build + factor (bcast)"
    chk = BlockCyclicCholesky(grid, n_chk, nb=nb, transport="bcast")
    chk.build_local_gram(MaternCovariance(2), xd[:n_chk].contiguous(), theta, 1e-4)
    info_chk = chk.factor()
    res["phase"] = "check_n8192: nll (world broadcast + all-reduce per block column)"
    nll_dist = chk.negative_log_likelihood(z[:n_chk])
    del chk
    if rank == 0:
        import gpmp_amd as gp

        th2 = np.concatenate(([theta[0], math.log(1e-4)], theta[1:]))
        ref = float(gp.Model(None, MaternCovariance(2, noise=True), None, th2, "zero").negative_log_likelihood_zero_mean(th2, x[:n_chk], z[:n_chk]))
        res["check_n8192"] = {"info": info_chk, "nll_block_cyclic": nll_dist, "nll_single_gpu": ref,
                              "rel_diff": abs(nll_dist - ref) / abs(ref)}

    for transport in ("bcast", "p2p"):
        for rep in ("warm", "timed"):
            res["phase"] = f"{transport}_{rep}: gram + factor at n={n}"
            ch = BlockCyclicCholesky(grid, n, nb=nb, transport=transport, profile=(rep == "timed"))
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            ch.build_local_gram(MaternCovariance(2), xd, theta, 1e-4)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            info = ch.factor()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            gram_s, potrf_s = tmax(t1 - t0), tmax(t2 - t1)
            entry = {"gram_s": gram_s, "potrf_s": potrf_s, "info": info,
                     "potrf_tflops_aggregate": (n ** 3 / 3.0) / potrf_s / 1e12,
                     "frac_of_aggregate_fp64_mfma_peak": (n ** 3 / 3.0) / potrf_s / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
                     "GB_received_per_gpu_max": tmax(ch.bytes_received / 1e9)}
            if rep == "timed":
                # summed HIP-event spans per phase on rank 0: side stream = diag / trsm / row_bcast / col_exchange /
                # lookahead_update (the panel chain and every collective), caller's stream = update
                entry["phases_ms_rank0"] = {k_: round(v_, 2) for k_, v_ in ch.phase_times().items()}
                t3 = time.perf_counter()
                res["phase"] = f"{transport}_{rep}: nll solve at n={n}"
                entry["nll"] = ch.negative_log_likelihood(z)
                torch.cuda.synchronize()
                entry["nll_solve_s"] = tmax(time.perf_counter() - t3)
            res[f"{transport}_{rep}"] = entry
            del ch
    res["phase"] = "done"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-n", dest="n", type=int, default=32768)
    ap.add_argument("--size-m", dest="m", type=int, default=50000)
    ap.add_argument("--dim-d", dest="d", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: do not record per-kernel HIP events in the timed region")
    ap.add_argument("--cpu-m-sample", type=int, default=512, help="prediction points of the CPU baseline's bounded sample")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1] / configs[3] extras")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GPMP_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (RCCL refuses two ranks on one
    # GPU): the ranks share the GPUs there are.  The measured run is one rank per GPU over nccl (= RCCL).
    backend = os.environ.get("GPMP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
        os.environ["LOCAL_RANK"] = str(local_rank)       # (gpmp_amd.num picks its device from it)
    torch.cuda.set_device(local_rank)
    dist_on = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # launched by torch.distributed.run
    if dist_on:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    import gpmp_amd as gp
    import gpmp_amd.num as gnp
    from gpmp_amd import _lib
    from gpmp_amd.kernel import MaternCovariance

    lib = _lib.load()
    n, m, d = args.n, args.m, args.d
    xi_h, zi_h, xt_h, theta = synth(n, m, d, rank)
    xi, zi, xt = gnp.asarray(xi_h), gnp.asarray(zi_h), gnp.asarray(xt_h)   # resident in HBM before timing
    model = gp.Model(None, MaternCovariance(2), None, theta, "zero")

    def step():
        zpm, zpv = model.predict(xi, zi, xt, convert_in=False, convert_out=False)
        nll = model.negative_log_likelihood_zero_mean(theta, xi, zi)
        return zpm, zpv, nll

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    import ctypes

    # HIP events on the launch stream around every launch of the DOMINANT kernel only (kind 9: the LDS-direct NN GEMM
    # of the n x m solve, 63 launches per step) inside the timed region; recording every kind (9000 events per step
    # around the small launches of the factorisations) costs 0.8 % of the step, so the other kinds are collected from
    # one extra, untimed step below.
    DOMINANT = 1 << 9
    if not args.no_kernel_events:
        lib.gpmp_profile_begin_kinds(DOMINANT)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    table = (ctypes.c_double * 36)()
    lib.gpmp_profile_end(table)
    prof_timed = np.array(list(table)).reshape(12, 3)
    # diagnostics of the other kernels: one untimed step with every kind recorded, scaled to the timed step count
    lib.gpmp_profile_begin_kinds(0xFFFFFFFF & ~DOMINANT)
    step()
    torch.cuda.synchronize()
    lib.gpmp_profile_end(table)
    prof = np.array(list(table)).reshape(12, 3) * args.steps
    prof[9] = prof_timed[9]

    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=gnp._dev())
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- outside the timed region: the Cholesky alone (BASELINE metric also quotes potrf TFLOP/s)
    potrf_ms = None
    if rank == 0:
        cov = model.covariance

        def _timed(fn, reps=2):
            best = float("inf")
            for _ in range(reps):
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t_)
            return best

        t_gram = _timed(lambda: cov.gram_lower(xi, theta))
        t_both = _timed(lambda: gnp.cholesky_factor(cov.gram_lower(xi, theta), overwrite=True))
        potrf_ms = 1e3 * (t_both - t_gram)

    zpm, zpv, nll = out
    assert bool(torch.isfinite(zpm).all()) and bool((zpv >= 0).all()) and math.isfinite(float(nll))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * m / (elapsed / args.steps)
        # ---- roofline of the dominant kernel: the NN fp64 MFMA GEMM behind the n x m triangular solve.
        # Algorithmic flops routed through it per step: n^2 m for V = L^-1 K(xi, xt)
        # (+ n^2 for the NLL's single right-hand side); launches and time measured with HIP events
        # on the launch stream over the timed region.
        steps = args.steps
        nn_cnt, nn_ms, nn_exec = prof[9]            # gemm_f64_kernel_v2<AKC=1,BKC=0>: the large trsm updates
        nn1_cnt, nn1_ms, nn1_exec = prof[1]         # register-staged kernel: K < 512 updates, diagonal-block products
        nt_cnt, nt_ms, nt_exec = prof[0] + prof[8]
        alg_solve = (float(n) * n * m + float(n) * n) * steps
        # Every flop of these launches is algorithmic: they are the plain rectangular products B2 -= L21 X1 of the
        # recursive solve (no triangular waste), so executed == algorithmic for THIS kernel.
        roof = {
            "bound": "mfma", "kernel": "gemm_f64_kernel_v2<true, false, true> (trsm updates B2 -= L21 X1, K >= 512)",
            "achieved": nn_exec / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else None,
            "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": (nn_exec / (nn_ms * 1e-3) / 1e12) / FP64_MFMA_PEAK_TFLOPS if nn_ms > 0 else None,
            "traffic": pmc_traffic_per_launch("gemm_f64_kernel_v2<true, false, true>") if (n, m) == (32768, 50000) else None,
            "traffic_note": "NOT measured in this run: bytes per launch at the fabric side of L2 (Infinity-Cache hits included) from the "
                            "latest COMMITTED rocprofv3 --pmc passes of this same command (separate FETCH_SIZE / WRITE_SIZE passes, "
                            "FETCH_SIZE x2 on gfx950 + WRITE_SIZE): " + str(getattr(pmc_traffic_per_launch, "source", None)),
            "launches_per_step": nn_cnt / steps, "avg_launch_ms": nn_ms / max(nn_cnt, 1),
            "algorithmic_flops_per_launch": nn_exec / max(nn_cnt, 1),
            "share_of_solve_flops": nn_exec / alg_solve,
            "whole_solve": {"algorithmic_tflops": alg_solve / ((nn_ms + nn1_ms) * 1e-3) / 1e12 if nn_ms > 0 else None,
                            "note": "n^2 m flops of V = L^-1 K(xi,xt) over ALL NN GEMM launches (both kernels)"},
        }
        gram_cnt, gram_ms, gram_bytes = prof[5]
        extra = {
            "potrf": {"n": n, "ms": potrf_ms, "tflops": (float(n) ** 3 / 3.0) / (potrf_ms * 1e-3) / 1e12,
                      "frac_of_fp64_mfma_peak": (float(n) ** 3 / 3.0) / (potrf_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                      "note": "whole factorisation (diagonal blocks + panels + trailing updates), wall time outside the timed region"},
            "potrf_gemm_nt_events": {"sum_of_spans_ms_per_step": nt_ms / steps, "launches_per_step": nt_cnt / steps,
                                     "note": "look-ahead overlaps panel and trailing kernels on two streams: spans are not additive"},
            "potf2_diag_blocks": {"sum_of_spans_ms_per_step": prof[4][1] / steps, "launches_per_step": prof[4][0] / steps},
            "gram": {"ms_per_step": gram_ms / steps, "GBps_written": gram_bytes / (gram_ms * 1e-3) / 1e9 if gram_ms > 0 else None,
                     "frac_of_hbm_peak": (gram_bytes / (gram_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if gram_ms > 0 else None},
            "coldots": {"ms_per_step": prof[6][1] / steps},
            "nll": float(nll),
        }
        line = {
            "metric": "fp64 predict+NLL throughput (points/s) and potrf TFLOP/s vs roofline, n=32k",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"d={d} Matern-5/2 anisotropic, n={n} train, m={m} test points per GPU, fp64 "
                                   f"predict (mean+variance) + one zero-mean NLL eval per step",
                       "n": n, "m_per_gpu": m, "d": d, "parallelism": f"xt-sharded x{world}, K replicated"},
            "roofline": roof,
            "extra": extra,
        }
        threads = _host_threads()
        if world == 1 and not args.no_extras:
            # configs[1] and configs[3] at their stated sizes, outside the timed region, each beside a same-size CPU figure
            del out, zpm, zpv
            torch.cuda.empty_cache()
            extra["config2"] = config2_extra(model, d, threads, not args.no_cpu_baseline)
            extra["config4"] = config4_extra(threads, not args.no_cpu_baseline)
            out = zpm = zpv = None
        if world == 1 and not args.no_cpu_baseline:      # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(n, m, d, threads, m_sample=args.cpu_m_sample)
            extra["potrf"]["host_potrf"] = line["cpu_baseline"].pop("host_potrf")

    # ---- N > 1, OPT-IN (GPMP_BENCH_DIST=1; =force also runs it on a 1 x 1 grid): the distributed Cholesky of configs[4]
    # as an extra after the headline measurement.  Off by default: its RCCL path with more than one rank has not run on
    # hardware yet (this pool gives one GPU per box), and an unproven collective schedule must not be able to turn the
    # headline run into a failure.  When it is on, a watchdog bounds it: on a timeout or an error rank 0 still prints
    # the headline line (with what finished and the phase that was in flight), and then EVERY rank leaves with a
    # non-zero exit code -- 3 for a timeout, 4 for an error -- so the run is recorded as failed, never as rc 0.
    dist_env = os.environ.get("GPMP_BENCH_DIST", "0")
    run_dist = dist_on and dist_env not in ("0", "") and (world > 1 or dist_env == "force")
    if run_dist:
        import threading

        res = {"status": "started", "phase": "setup"}
        done = threading.Lock()
        EXIT = {"ok": 0, "timeout": 3, "error": 4}

        def finish(status):
            if not done.acquire(blocking=False):
                return
            res["status"] = status
            if rank == 0:
                line["extra"]["dist_potrf"] = res
                print(json.dumps(line), flush=True)
            if status != "ok":
                sys.stderr.write(f"[bench rank {rank}] distributed extra: {status} in phase {res.get('phase')!r}"
                                 f" {res.get('error', '')}\n")
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(EXIT[status])   # collectives may be wedged: no orderly teardown, and never exit code 0

        wd = threading.Timer(float(os.environ.get("GPMP_BENCH_DIST_TIMEOUT", "240")), finish, args=("timeout",))
        wd.daemon = True
        wd.start()
        out = zpm = zpv = None
        try:
            dist_potrf_extra(world, rank, res)
            wd.cancel()
            finish("ok")
        except BaseException as e:     # noqa: BLE001 -- report, then leave with exit code 4 (peers: their own watchdogs, 3)
            wd.cancel()
            res["error"] = f"{type(e).__name__}: {e}"[:400]
            finish("error")
    elif rank == 0:
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
