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
are replicated and EVERY rank factors K itself -- no collective on the data path.  `extra.strong_scaling`
is the same step with a FIXED total of m points split over the ranks.

value = (N * m) / (max over ranks of the time of K steps / K)   [points/s].

Processes.  `--gpus N` with N > 1 and no RANK in the environment: this process is a LAUNCHER that makes no GPU call
(it does not even import torch): it starts N worker processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
set, each in its own session), relays rank 0's JSON line as its own last stdout line and leaves with a non-zero code if
any worker failed.  Launched under `torch.distributed.run` the N ranks are the workers; a WORLD_SIZE that differs from
--gpus is an error (exit 2), never a silent one-rank run.
At N > 1 the 2-D block-cyclic Cholesky of BASELINE.json configs[4] runs BY DEFAULT after the headline, in a separate
group of N fresh processes under a time limit (GPMP_BENCH_DIST_TIMEOUT): a wedged or failing collective there costs
`extra.dist_potrf` (status + the phase in flight are recorded) and the exit code (3 = timeout, 4 = error), not the
headline line.
CPU baseline (rank 0 at N = 1): at the headline shape ONE full-size step of the oracle is MEASURED in this run, in a child process
(`--role cpu-full`) under a time limit; the bounded-sample model is the fall-back.  Environment: GPMP_BENCH_CPU_FULL (1 at the
headline shape; 0: model only), GPMP_BENCH_CPU_FULL_TIMEOUT (360 s), GPMP_BENCH_CPU_MODEL (1: the model beside the measurement),
GPMP_BENCH_LIVE_PMC (1: two rocprofv3 counter passes of one step each measure roofline.traffic in the run); N > 1: GPMP_BENCH_BACKEND
(nccl; gloo = rehearsal on fewer GPUs than ranks), GPMP_BENCH_DIST (1), GPMP_BENCH_DIST_TIMEOUT (420 s), GPMP_BENCH_DIST_N,
GPMP_BENCH_DIST_STRONG / GPMP_BENCH_DIST_MORE (1: the strong-scaling / predict + gradient parts of the distributed extra),
GPMP_BENCH_STRONG_NM, GPMP_BENCH_HEADLINE_TIMEOUT (1500 s); tests: GPMP_BENCH_STUB_MODULE.
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
        kb, disp = 0.0, 0          # the kernel may appear as several template instances (tile widths): all of its launches together
        for row in csv.DictReader(open(path)):
            if kernel_substr in row["kernel"]:
                kb += float(row["per_dispatch_KB_raw"]) * int(row["dispatches"])
                disp += int(row["dispatches"])
        if disp:
            tot += factor * 1024.0 * kb / disp
            n_disp = disp
    return tot if n_disp else None


def pmc_traffic_live(kernel_substr, args, timeout_s=300.0):
    """HBM-side bytes per launch of a kernel measured IN THIS RUN: two child processes of this same benchmark (one step, no
    extras) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only (no tracing domain
    beside them), the program itself after `--`, as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes; FETCH_SIZE and
    WRITE_SIZE count KB, and on gfx950 FETCH_SIZE reports half of a wide streaming read (x2).  Children are STARTED (nothing is
    exec'ed from this GPU-initialised process), each in its own session under a time limit.  Returns (bytes per launch,
    note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    tot = 0.0
    launches = None
    for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        out = tempfile.mkdtemp(prefix="gpmp_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--role", "headline",
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--no-kernel-events", "--no-live-pmc",
               "--size-n", str(args.n), "--size-m", str(args.m), "--dim-d", str(args.d)]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GPMP_BENCH_CHILD")}
        env["TMPDIR"] = "/tmp"
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                return None, f"{counter} pass exceeded {timeout_s:.0f} s and was killed"
            if rc != 0:
                return None, f"{counter} pass ended with code {rc}"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"{counter} pass wrote no counter_collection.csv"
            val, disp = 0.0, set()
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                    val += float(row["Counter_Value"])
                    disp.add(row["Dispatch_Id"])
            if not disp:
                return None, f"{counter} pass: no dispatch of the kernel"
            launches = len(disp)
            tot += factor * 1024.0 * val / launches
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return tot, (f"measured in this run: two rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE; KB -> bytes) over one step of "
                 f"this command each, {launches} launches of the kernel per pass; bytes at the fabric side of L2, Infinity-Cache hits included")


def _host_threads():
    """BLAS threads the CPU baseline uses: the cores this process may actually run on -- the smallest of the affinity mask, the
    cgroup CPU quota and the PHYSICAL core count (round 3 let the BLAS pool default to every logical CPU of the host: 128
    oversubscribed threads ran dpotrf at 0.32 TFLOP/s where 64 reach 0.9)."""
    cands = [os.cpu_count() or 1]
    try:
        cands.append(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        import psutil

        phys = psutil.cpu_count(logical=False)
        if phys:
            cands.append(int(phys))
    except Exception:  # noqa: BLE001
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cands.append(max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cands))


def _latest_cpu_fullsize_log(n, m, d):
    """The most recent COMMITTED full-size CPU run of the headline step (tools/cpu_fullsize_step.py -> profiles/r*/cpu_fullsize_step.log,
    last JSON line) for this (n, m, d), or None: what the assembled figure should be read against.  Read at run time, never a literal."""
    import glob

    import re

    def round_no(path):                     # numerically: r10 after r9
        mt = re.search(r"profiles[/\\]r(\d+)[/\\]", path)
        return (int(mt.group(1)) if mt else 0, path)

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cpu_fullsize_step*.log")), key=round_no):
        try:
            with open(path) as f:
                rec = _last_json_line(f.read())
        except OSError:
            continue
        if rec and rec.get("tool") == "cpu_fullsize_step" and (rec.get("n"), rec.get("m"), rec.get("d")) == (n, m, d):
            best = {"s_per_step": rec["full_step_s"], "points_per_s": rec["points_per_s"], "threads": rec.get("threads"),
                    "log": os.path.relpath(path, ROOT), "tool": "tools/cpu_fullsize_step.py",
                    "note": "NOT measured in this run: the latest committed full-size run (possibly another box of the pool)"}
    return best


def cpu_baseline(n, m, d, threads, m_sample=2048, n_sample=8192):
    """The oracle (NumPy / SciPy restatement of the reference's NumPy backend: cdist -> Matern ufuncs -> cholesky ->
    2 x solve_triangular -> einsum) on the host cores, on a BOUNDED sample of the headline workload (~45-60 s of CPU work).
    The full step takes 3.5 minutes on this host (ONE full-size run: tools/cpu_fullsize_step.py, profiles/r3/cpu_fullsize_step.log:
    209.7 s = 238 points/s on 64 threads), so `value` is ASSEMBLED, and round 3 changed what is sampled after that run showed
    the previous model 2.9x too slow: LAPACK / BLAS-3 rates depend on n, so everything that goes through BLAS is now timed at
    the FULL n and only the number of prediction points is sampled:
      * Cholesky at the full n (numpy.linalg.cholesky on a stand-in SPD matrix: dpotrf's time does not depend on the entries);
      * with that full-size factor: the two triangular solves (their TFLOP/s is printed), the reductions and the NLL's
        single-vector solves, for m_sample = 2048 of the m prediction points (8192 columns run the host trsm no faster -- 0.55 vs 0.52 TFLOP/s -- and take 88 s of CPU work instead of 48) (x m / m_sample: linear in m at fixed n);
      * Gram(xi, xt_s) at the full n x m_sample (x m / m_sample) and Gram(xi, xi) at n_sample = 8192 observations
        (x (n / n_sample)^2): SciPy cdist + the Matern ufuncs are single-threaded elementwise passes, linear in the entries.
        T = 2 (Gram_ii (n/n_s)^2 + Cholesky(n)) + NLL_tail(n) + m/m_s (Gram_it(n) + Solve(n) + Reductions(n))
    `assembled` says that value = m / T is a model figure; `measured_s` holds every measured piece, `extrapolated_s` the terms."""
    from scipy.linalg import solve_triangular

    from oracle import gp_oracle as orc

    ns = min(n, n_sample)
    ms = min(m, m_sample)
    xi, zi, xt, theta = synth(n, m, d, 0)
    xs = xt[:ms]
    t = {}
    try:                                     # one BLAS thread per usable core (see _host_threads), for everything below
        from threadpoolctl import threadpool_limits

        _limit = threadpool_limits(limits=threads)
    except Exception:  # noqa: BLE001
        _limit = None
    try:
        return _cpu_baseline_sampled(orc, solve_triangular, xi, zi, xt, xs, theta, n, m, d, ns, ms, threads, t)
    finally:
        if _limit is not None:
            _limit.restore_original_limits()


def _cpu_baseline_sampled(orc, solve_triangular, xi, zi, xt, xs, theta, n, m, d, ns, ms, threads, t):
    np.linalg.cholesky(orc.maternp_covariance(xi[:512], None, 2, theta))    # BLAS thread pool / page-in warm-up, untimed

    def tick(name, fn):
        t0 = time.perf_counter()
        out = fn()
        t[name] = time.perf_counter() - t0
        return out

    K = tick("gram_ii_ns", lambda: orc.maternp_covariance(xi[:ns], None, 2, theta))          # kriging.py:59, likelihood.py:43
    if ns < n:
        del K
        # dpotrf at the full n: its time does not depend on the entries, so a cheap SPD matrix stands in for K (8n^2 bytes);
        # its factor is a well-conditioned lower-triangular matrix of the right size for timing the solves
        K = np.full((n, n), 0.5)
        K[np.diag_indices(n)] = float(n)
    L = tick("cholesky", lambda: np.linalg.cholesky(K))                                      # numpy_backend.py:466
    del K
    Kit = tick("gram_it", lambda: orc.maternp_covariance(xi, xs, 2, theta))                 # kriging.py:60

    def trsm_pair():                                                                         # numpy_backend.py:467-468
        y = solve_triangular(L, Kit, lower=True)
        return solve_triangular(L.T, y, lower=False)

    lam = tick("trsm_pair", trsm_pair)

    def reductions():                                                                        # kriging.py:193-194, model.py:298
        var = orc.maternp_covariance(xs, None, 2, theta, True) - np.einsum("i..., i...", lam, Kit)
        return np.einsum("i..., i...", lam, zi), var

    tick("reductions", reductions)
    del lam

    def nll_tail():                                                                          # likelihood.py:46-51
        a = solve_triangular(L.T, solve_triangular(L, zi, lower=True), lower=False)
        return 0.5 * (n * math.log(2 * math.pi) + 2.0 * np.sum(np.log(np.diag(L))) + zi @ a)

    tick("nll_tail", nll_tail)
    del L, Kit
    r = n / ns
    trsm_tflops = 2.0 * n * n * ms / t["trsm_pair"] / 1e12
    ext = {"gram_ii_x2": 2.0 * t["gram_ii_ns"] * r * r, "cholesky_x2": 2.0 * t["cholesky"], "nll_tail": t["nll_tail"],
           "gram_it": t["gram_it"] * m / ms, "trsm_pair": t["trsm_pair"] * m / ms, "reductions": t["reductions"] * m / ms}
    step = sum(ext.values())
    out = {"value": m / step, "unit": "points/s", "cores": threads, "kind": "port", "assembled": True,
           "sample": f"oracle (SciPy cdist + Matern ufuncs + LAPACK), d={d}, everything that goes through BLAS at the FULL n={n}: Cholesky "
                     f"({t['cholesky']:.1f} s), 2 triangular solves ({trsm_tflops:.2f} TFLOP/s measured) + reductions + Gram(xi,xt) for {ms} of the "
                     f"{m} points (x m/{ms}), NLL solves; Gram(xi,xi) at n_s={ns} (x (n/n_s)^2: single-threaded elementwise passes); NOT a "
                     f"measured step: {step:.0f} s per predict+NLL step ASSEMBLED from these; CPU work done {sum(t.values()):.0f} s; BLAS threads={threads}",
           "cpu_trsm_tflops": trsm_tflops,
           "host_potrf": {"n": n, "s": t["cholesky"], "tflops": n ** 3 / 3.0 / t["cholesky"] / 1e12},
           "measured_s": {k_: round(v_, 3) for k_, v_ in t.items()},
           "extrapolated_s": {k_: round(v_, 2) for k_, v_ in ext.items()}}
    # the assembled figure beside a MEASURED full-size step: the latest committed run (read from its log)
    full = _latest_cpu_fullsize_log(n, m, d)
    out["full_size_run"] = full
    out["model_over_measured"] = (step / full["s_per_step"]) if full else None
    return out


def cpu_full_child(args):
    """`--role cpu-full`: ONE full-size predict + NLL step of the oracle on the host cores, in a process of its own (no torch, no
    GPU): what gpmp/core/kriging.py:35-67 + gpmp/num/numpy_backend.py:465-469 + gpmp/core/likelihood.py:18-52 cost on this host.
    The LAPACK calls inside the step are timed by wrapping them (the step itself is untouched).  Prints one JSON line."""
    from oracle import gp_oracle as orc

    n, m, d = args.n, args.m, args.d
    threads = _host_threads()
    limit = None
    try:
        from threadpoolctl import threadpool_limits

        limit = threadpool_limits(limits=threads)
    except Exception:  # noqa: BLE001
        pass
    spans = {"cholesky": [], "solve_triangular": []}
    chol0, trs0 = np.linalg.cholesky, orc._sp_solve_triangular

    def timed(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            spans[name].append((time.perf_counter() - t0, tuple(np.shape(a[1])) if len(a) > 1 else tuple(np.shape(a[0]))))
            return r
        return w

    try:
        xi, zi, xt, theta = synth(n, m, d, 0)
        np.linalg.cholesky(orc.maternp_covariance(xi[:512], None, 2, theta))    # BLAS thread pool / page-in warm-up, untimed
        np.linalg.cholesky = timed("cholesky", chol0)
        orc._sp_solve_triangular = timed("solve_triangular", trs0)
        om = orc.OracleModel(None, lambda x, y, th, pairwise=False: orc.maternp_covariance(x, y, 2, th, pairwise), None, theta, "zero")
        t0 = time.perf_counter()
        zpm, zpv = orc.predict(om, xi, zi, xt)
        t1 = time.perf_counter()
        nll = float(orc.negative_log_likelihood_zero_mean(om, theta, xi, zi))
        t2 = time.perf_counter()
    finally:
        np.linalg.cholesky, orc._sp_solve_triangular = chol0, trs0
        if limit is not None:
            limit.restore_original_limits()
    chol_s = [c[0] for c in spans["cholesky"]]
    trsm_s = sum(c[0] for c in spans["solve_triangular"] if len(c[1]) == 2 and c[1][1] == m)
    print(json.dumps({"tool": "bench.py --role cpu-full", "n": n, "m": m, "d": d, "threads": threads, "predict_s": t1 - t0, "nll_s": t2 - t1,
                      "full_step_s": t2 - t0, "points_per_s": m / (t2 - t0), "nll": nll, "finite": bool(np.isfinite(zpm).all() and np.isfinite(zpv).all()),
                      "cholesky_s": chol_s, "host_potrf_tflops": n ** 3 / 3.0 / min(chol_s) / 1e12 if chol_s else None,
                      "trsm_pair_s": trsm_s, "host_trsm_tflops": 2.0 * n * n * m / trsm_s / 1e12 if trsm_s > 0 else None}), flush=True)
    return 0


def cpu_baseline_measured(n, m, d, timeout_s):
    """`cpu_baseline` as a MEASUREMENT of this run: one full-size step of the oracle (cpu_full_child) in a child process of its own
    session under a time limit -- a slow host costs the child, never the GPU line.  Returns (record, None) or (None, reason)."""
    import signal
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--role", "cpu-full", "--size-n", str(n), "--size-m", str(m), "--dim-d", str(d)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GPMP_BENCH_CHILD")}
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    except OSError as e:
        return None, f"could not start the child: {e}"
    try:
        so, se = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        p.communicate()
        return None, f"the full-size CPU step exceeded its limit of {timeout_s:.0f} s and was killed"
    if p.returncode != 0:
        return None, f"the full-size CPU step ended with code {p.returncode}: {se[-300:]}"
    rec = _last_json_line(so)
    if not rec or "full_step_s" not in rec:
        return None, "the full-size CPU step printed no record"
    rec["child_wall_s"] = time.perf_counter() - t0
    return rec, None


def cpu_baseline_for_line(n, m, d, threads, m_sample):
    """The `cpu_baseline` object of the JSON line.  Headline shape: a same-run MEASURED full-size step (north_star: "the NumPy/SciPy
    CPU path timed on the node's host cores (count stated) in the same run"), ~3 minutes on 16 cores -- longer than a bounded
    sample on purpose: round 4's assembled figure was 1.35 x off, and the whole default run still ends within ~5 minutes.  The
    assembled model is the fall-back (child failed or timed out; GPMP_BENCH_CPU_FULL=0; other shapes) and says so."""
    want_full = os.environ.get("GPMP_BENCH_CPU_FULL", "1" if (n, m, d) == (32768, 50000, 8) else "0") == "1"
    reason = None
    if want_full:
        rec, reason = cpu_baseline_measured(n, m, d, float(os.environ.get("GPMP_BENCH_CPU_FULL_TIMEOUT", "360")))
        if rec is not None:
            out = {"value": rec["points_per_s"], "unit": "points/s", "cores": rec["threads"], "kind": "port", "assembled": False,
                   "sample": f"ONE full-size predict + NLL step of the oracle (SciPy cdist + Matern ufuncs + LAPACK: the reference's NumPy-backend "
                             f"call sequence), d={d}, n={n}, m={m}, MEASURED in this run in a child process of its own: {rec['full_step_s']:.1f} s "
                             f"(predict {rec['predict_s']:.1f} s + NLL {rec['nll_s']:.1f} s) on {rec['threads']} BLAS threads = the host cores this "
                             f"process may use; no extrapolation",
                   "full_size_run": {"s_per_step": rec["full_step_s"], "points_per_s": rec["points_per_s"], "threads": rec["threads"],
                                     "predict_s": rec["predict_s"], "nll_s": rec["nll_s"], "nll": rec["nll"], "note": "measured in THIS run"},
                   "cpu_trsm_tflops": rec.get("host_trsm_tflops"),
                   "host_potrf": {"n": n, "s": min(rec["cholesky_s"]) if rec.get("cholesky_s") else None, "tflops": rec.get("host_potrf_tflops"),
                                  "note": "numpy.linalg.cholesky inside the measured step (best of its two calls)"}}
            if os.environ.get("GPMP_BENCH_CPU_MODEL", "0") == "1":       # the bounded-sample model beside it, on request
                model = cpu_baseline(n, m, d, threads, m_sample=m_sample)
                out["assembled_value"] = model["value"]
                out["model_over_measured"] = (m / model["value"]) / rec["full_step_s"]
                out["assembled_model"] = {k_: model[k_] for k_ in ("measured_s", "extrapolated_s", "sample")}
            return out
    out = cpu_baseline(n, m, d, threads, m_sample=m_sample)
    if reason is not None:
        out["sample"] = f"FALL-BACK ({reason}): " + out["sample"]
        out["measured_step_failed"] = reason
    return out


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


# ------------------------------------------------------------------------------------------------------------------
# process groups: launcher side (NO GPU call in this section -- it runs in processes that never import torch)
# ------------------------------------------------------------------------------------------------------------------
EXIT_TIMEOUT, EXIT_ERROR = 3, 4


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, timeout_s, env_extra=None, grace_s=15.0):
    """Start ``n`` fresh worker processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their
    environment, one session each so that a worker and anything it started can be killed as a group), wait for them under a
    time limit and return {"status": "ok" | "timeout" | "error", "rc": [...], "pids": [...], "stdout0": rank 0's stdout,
    "seconds": wall time}.  A worker that leaves with a non-zero code ends the group: its peers (possibly waiting for it
    inside a collective) get ``grace_s`` seconds, then SIGKILL.  On a timeout every worker's process group is killed.
    stderr of the workers is inherited; stdout of ranks > 0 goes to stderr."""
    import signal
    import subprocess
    import threading

    port = _free_port()
    procs, chunks = [], []
    for r in range(n):
        # (when rank 0 of a torch.distributed.run launch starts this group, the agent's variables must not leak into it:
        #  TORCHELASTIC_USE_AGENT_STORE would make the children look for the agent's store on OUR port)
        env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC", "GROUP_", "ROLE_"))}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GPMP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))

    def drain(stream):            # a reader thread: rank 0 can never block on a full pipe
        for raw in iter(stream.readline, b""):
            chunks.append(raw.decode(errors="replace"))

    reader = threading.Thread(target=drain, args=(procs[0].stdout,), daemon=True)
    reader.start()

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)        # p is its session's leader: pgid == pid
                except (ProcessLookupError, PermissionError):
                    pass
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass

    t0 = time.monotonic()
    status, first_bad, first_rc = "ok", None, None
    while True:
        rcs = [p.poll() for p in procs]
        now = time.monotonic()
        if all(rc is not None for rc in rcs):
            if any(rc != 0 for rc in rcs):
                status = "error"
                if first_rc is None:
                    first_rc = next(rc for rc in rcs if rc != 0)
            break
        if first_bad is None and any(rc not in (None, 0) for rc in rcs):
            first_bad = now
            first_rc = next(rc for rc in rcs if rc not in (None, 0))     # the failure that came first (its peers fail after it)
        if first_bad is not None and now - first_bad > grace_s:
            status = "error"
            kill_all()
            break
        if now - t0 > timeout_s:
            status = "timeout"
            kill_all()
            break
        time.sleep(0.1)
    reader.join(timeout=5)
    return {"status": status, "rc": [p.poll() for p in procs], "first_bad_rc": first_rc, "pids": [p.pid for p in procs],
            "stdout0": "".join(chunks), "seconds": time.monotonic() - t0}


def _last_json_line(text):
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def run_dist_extra_group(n_ranks):
    """BASELINE.json configs[4] as an extra: N FRESH processes (never the ones that ran the headline), a progress file that
    rank 0 rewrites at every phase change, and a time limit enforced from outside -- so whatever a collective does, the
    caller gets {"status", "phase", everything that finished} back and decides the exit code."""
    import tempfile

    timeout_s = float(os.environ.get("GPMP_BENCH_DIST_TIMEOUT", "420"))
    with tempfile.TemporaryDirectory(prefix="gpmp_bench_") as tmp:
        path = os.path.join(tmp, "dist_progress.json")
        got = spawn_ranks(n_ranks, ["--role", "dist-extra", "--gpus", str(n_ranks), "--progress", path], timeout_s)
        res = {"status": "started", "phase": "no progress written: the workers did not get as far as setup"}
        try:
            with open(path) as f:
                res = json.load(f)
        except (OSError, ValueError):
            pass
    if got["status"] != "ok":
        res["status"] = got["status"]                    # "timeout" / "error": what was in flight is in res["phase"]
    elif res.get("status") != "ok":
        res["status"] = "error"
    res.update({"worker_rc": got["rc"], "worker_pids": got["pids"], "wall_s": round(got["seconds"], 2), "timeout_s": timeout_s,
                "isolation": f"{n_ranks} fresh processes in their own sessions, killed as groups on timeout"})
    return res


def dist_extra_enabled(world):
    env = os.environ.get("GPMP_BENCH_DIST", "1")
    return env not in ("0", "") and (world > 1 or env == "force")


def finish_with_dist_extra(line, world):
    """The coordinator's last act (the launcher, or rank 0 of a torch.distributed.run launch after it left its process
    group): run the distributed extra in its own process group, attach it, print THE line, return the exit code."""
    rc = 0
    if dist_extra_enabled(world):
        res = run_dist_extra_group(world)
        line.setdefault("extra", {})["dist_potrf"] = res
        if res["status"] != "ok":
            rc = EXIT_TIMEOUT if res["status"] == "timeout" else EXIT_ERROR
            sys.stderr.write(f"[bench] distributed extra: {res['status']} in phase {res.get('phase')!r} {res.get('error', '')}\n")
    print(json.dumps(line), flush=True)
    return rc


def launcher_main(args, argv):
    """`python bench.py --gpus N` (N > 1, no RANK in the environment).  Makes no GPU call."""
    # the launcher itself never imports torch; if the interpreter came with it preloaded, what matters is that no GPU call was made
    if "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized():
        sys.stderr.write("[bench] the launcher process has an initialised GPU context: refusing to start GPU workers from it\n")
        return EXIT_ERROR
    got = spawn_ranks(args.gpus, ["--role", "headline"] + argv, float(os.environ.get("GPMP_BENCH_HEADLINE_TIMEOUT", "1500")))
    line = _last_json_line(got["stdout0"])
    if got["status"] != "ok" or line is None:
        sys.stderr.write(f"[bench] headline workers: {got['status']}, exit codes {got['rc']}\n")
        print(json.dumps({"metric": METRIC, "value": None, "unit": "points/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "error": f"headline workers {got['status']}: exit codes {got['rc']}", "partial": line}), flush=True)
        bad = got["first_bad_rc"]
        return EXIT_TIMEOUT if got["status"] == "timeout" else (bad if bad is not None and 0 < bad < 256 else EXIT_ERROR)
    line.setdefault("extra", {})["launcher"] = {"pid": os.getpid(), "worker_pids": got["pids"], "wall_s": round(got["seconds"], 2),
                                                 "note": "GPU-free parent started one fresh process per rank"}
    return finish_with_dist_extra(line, args.gpus)


# ------------------------------------------------------------------------------------------------------------------
# who ran: one record per rank, gathered through the process group itself (so the record proves the group's membership)
# ------------------------------------------------------------------------------------------------------------------
def _device_identity(have_gpu):
    """(index, name, uuid, pci bus id) of this process's current GPU; Nones without one (stub runs)."""
    if not have_gpu:
        return {"device_index": None, "device_name": None, "uuid": None, "pci_bus_id": None}
    import torch

    idx = torch.cuda.current_device()
    out = {"device_index": idx, "device_name": torch.cuda.get_device_name(idx), "uuid": None, "pci_bus_id": None}
    try:
        props = torch.cuda.get_device_properties(idx)
        u = getattr(props, "uuid", None)
        out["uuid"] = str(u) if u is not None else None
        if all(hasattr(props, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            out["pci_bus_id"] = "%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    except Exception as e:  # noqa: BLE001 -- identity is evidence, never a reason to lose the measurement
        out["identity_error"] = f"{type(e).__name__}: {e}"[:120]
    return out


def gather_ranks_seen(dist_mod, rank, world, backend, have_gpu):
    """All-gather of (rank, pid, host, LOCAL_RANK, current device index / name / uuid / PCI bus id) over the process group
    that produced the numbers beside it, + the RCCL version torch was built with: the record then shows N distinct DEVICES
    under RCCL, not just N PIDs.  Every rank gets the full record; `distinct_devices` counts (host, uuid or PCI id or index)."""
    import socket

    me = {"rank": rank, "pid": os.getpid(), "host": socket.gethostname(), "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}
    me.update(_device_identity(have_gpu))
    recs = [me]
    if dist_mod is not None and world > 1:
        recs = [None] * world
        dist_mod.all_gather_object(recs, me)
    rccl = None
    if have_gpu:
        try:
            import torch

            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl = None
    ident = {(r["host"], r.get("uuid") or r.get("pci_bus_id") or r.get("device_index")) for r in recs}
    return {"backend": backend, "rccl_version": rccl, "world": world, "ranks": recs,
            "distinct_pids": len({(r["host"], r["pid"]) for r in recs}),
            "distinct_devices": len(ident) if have_gpu else 0,
            # one process AND one device per rank (false in a gloo rehearsal where ranks share a GPU, and in stub runs)
            "one_device_per_rank": bool(have_gpu and len(ident) == world and len({(r["host"], r["pid"]) for r in recs}) == world)}


def time_block_cyclic_strong_scaling(runner, dist_mod, world, rank, reps, tmax):
    """Timing harness of extra.strong_scaling_block_cyclic (shared with tests/bench_stub.py): the headline STEP -- predict at
    m points + one zero-mean NLL at the same n -- with K DISTRIBUTED over the grid instead of factored on every rank.
    `runner.step(shared_factor)`: one step; shared_factor False = K built and factored twice (what the single-GPU headline step
    does: no cache), True = once for both calls.  One warm-up step each, then `reps` steps between barrier + sync pairs, max over
    ranks.  `runner.check()` (every rank calls it; rank 0 compares with its own single-GPU result) -> dict of deviations."""
    out = {}
    for key, shared in (("two_factorisations", False), ("shared_factor", True)):
        runner.step(shared)
        runner.sync(); dist_mod.barrier(); runner.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            runner.step(shared)
        runner.sync(); dist_mod.barrier(); runner.sync()
        dt = tmax(time.perf_counter() - t0) / reps
        out[key] = {"steps": reps, "ms_per_step": 1e3 * dt, "points_per_s": runner.m / dt}
    out["values_check"] = runner.check()
    return out


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: the distributed extra (worker side)
# ------------------------------------------------------------------------------------------------------------------
DIST_N = {2: 65536, 4: 90112, 8: 131072}     # about 17 GB of local matrix per GPU; 8 GPUs = BASELINE.json configs[4]


class _Progress(dict):
    """dict that rank 0 mirrors into a JSON file at every assignment (atomic rename): what the launcher reads back."""

    def __init__(self, path):
        super().__init__()
        self._path = path

    def _flush(self):
        if self._path:
            tmp = self._path + ".tmp"
            with open(tmp, "w") as f:
                json.dump(self, f)
            os.replace(tmp, self._path)

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        self._flush()

    def update(self, *a, **kw):
        super().update(*a, **kw)
        self._flush()


def dist_potrf_extra(world, rank, res):
    """The 2-D block-cyclic Cholesky (+ NLL) of BASELINE.json configs[4] (n = 131072 on the 2 x 4 grid of 8 GPUs; n scaled
    to the same memory per GPU on 2 / 4 GPUs; GPMP_BENCH_DIST_N overrides) with RCCL panel broadcasts, once per transport,
    preceded by a values check at n = 8192 against the single-GPU path.  Fills ``res`` progressively."""
    import torch
    import torch.distributed as dist

    import gpmp_amd.num as gnp
    from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid
    from gpmp_amd.kernel import MaternCovariance

    pr, pc = ProcessGrid.default_shape(world)
    n = int(os.environ.get("GPMP_BENCH_DIST_N", "0")) or DIST_N.get(world, max(1024, int(46000 * math.sqrt(world)) // 1024 * 1024))
    d, nb = 8, 1024
    nccl = dist.get_backend() == "nccl"
    res.update({"n": n, "d": d, "grid": f"{pr}x{pc}", "block": nb, "noise_variance": 1e-4, "backend": dist.get_backend(),
                "note": "own process group, outside the timed region; flops = n^3/3 over wall time of factor() (max over ranks)"})
    rng = np.random.default_rng(1234)
    x = rng.random((n, d))
    z = np.sin(2 * np.pi * x[:, 0]) + x[:, 1:].sum(axis=1)
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    xd = gnp.asarray(x)
    grid = ProcessGrid(pr, pc)
    torch.cuda.empty_cache()

    def tmax(v):
        t = torch.tensor([v], dtype=torch.float64, device=gnp._dev() if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def agree(ok_here, what):
        """rank 0 holds the verdict of a values check; EVERY rank learns it through the process group and raises or goes on
        together (a failure on rank 0 alone would leave the others waiting in the next collective until the time limit)."""
        t = torch.tensor([1.0 if ok_here else 0.0], dtype=torch.float64, device=gnp._dev() if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if float(t.item()) == 0.0:
            raise RuntimeError(what)

    # ---- values first: a small problem through the same grid / collectives against the single-GPU path on rank 0
    n_chk = min(8192, n)
    for transport in ("bcast", "p2p"):
        res["phase"] = f"check_n{n_chk} ({transport}): build + factor"
        chk = BlockCyclicCholesky(grid, n_chk, nb=nb, transport=transport)
        chk.build_local_gram(MaternCovariance(2), xd[:n_chk].contiguous(), theta, 1e-4)
        info_chk = chk.factor()
        res["phase"] = f"check_n{n_chk} ({transport}): nll (world broadcast + all-reduce per block column)"
        nll_dist = chk.negative_log_likelihood(z[:n_chk])
        del chk
        if rank == 0:
            import gpmp_amd as gp

            th2 = np.concatenate(([theta[0], math.log(1e-4)], theta[1:]))
            ref = float(gp.Model(None, MaternCovariance(2, noise=True), None, th2, "zero").negative_log_likelihood_zero_mean(th2, x[:n_chk], z[:n_chk]))
            res[f"check_n{n_chk}_{transport}"] = {"info": info_chk, "nll_block_cyclic": nll_dist, "nll_single_gpu": ref,
                                                   "rel_diff": abs(nll_dist - ref) / abs(ref)}
            ok_chk = bool(info_chk == 0 and abs(nll_dist - ref) <= 1e-9 * abs(ref))
        else:
            ok_chk = True
        agree(ok_chk, f"values check failed ({transport}) at n={n_chk}: the block-cyclic NLL differs from rank 0's single-GPU value (see check_n{n_chk}_{transport})")

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
            if not (transport == "p2p" and rep == "timed"):
                del ch
    # ---- strong scaling on the path that can scale: the HEADLINE workload (n = 32768, m = 50000 in total) through
    #      DistributedModel.predict + NLL on this grid -- K distributed, nothing replicated
    if os.environ.get("GPMP_BENCH_DIST_STRONG", "1") != "0":
        res["phase"] = "strong_scaling_block_cyclic: headline workload on the block-cyclic factor"
        res["strong_scaling_block_cyclic"] = strong_block_cyclic_extra(world, rank, grid, tmax, res)
        vc = res["strong_scaling_block_cyclic"].get("values_check")
        agree(rank != 0 or bool(vc and vc.get("ok")), f"strong_scaling_block_cyclic: distributed results differ from the single-GPU ones: {vc}")
    # ---- on the last factor (still resident): what a parameter fit and a prediction at this n cost.  LAST on
    #      purpose: everything above is already in the progress file if one of these wedges.
    if os.environ.get("GPMP_BENCH_DIST_MORE", "1") != "0" and info == 0:
        m_pred = 50000
        res["phase"] = f"predict: zero-mean prediction at {m_pred} points on the block-cyclic factor (overlapped many-RHS solve)"
        xt = np.random.default_rng(4321).random((m_pred, d))
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        mean, var, _ = ch.predict_zero_mean(MaternCovariance(2), xd, z, xt, theta)
        torch.cuda.synchronize()
        dt = tmax(time.perf_counter() - t0)
        res["predict"] = {"m": m_pred, "s": dt, "points_per_s": m_pred / dt, "solve_tflops_aggregate": float(n) * n * m_pred / dt / 1e12,
                          "finite": bool(np.isfinite(mean).all() and np.isfinite(var).all())}
        res["phase"] = "value_and_grad: ML value + analytic gradient on the block-cyclic factor (inverse factor + ring of T^T T blocks)"
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        th_noisy = np.concatenate(([theta[0], math.log(1e-4)], theta[1:]))
        v, g = ch.value_and_grad(x, z, th_noisy, 2, noise=True)
        torch.cuda.synchronize()
        dt = tmax(time.perf_counter() - t0)
        res["value_and_grad"] = {"s": dt, "value": float(v), "grad_norm": float(np.linalg.norm(g)),
                                 "note": "one criterion + gradient evaluation of a parameter fit at this n (the factorisation above is the rest of it)"}
    del ch
    res["phase"] = "done"


class _BlockCyclicHeadline:
    """The headline step on the block-cyclic factor (one instance per rank; every rank makes the same calls)."""

    def __init__(self, grid, n, m, d, rank):
        import torch

        from gpmp_amd.dist import DistributedModel
        from gpmp_amd.kernel import MaternCovariance

        self.torch, self.rank, self.n, self.m, self.d = torch, rank, n, m, d
        self.xi, self.zi, self.xt, self.theta = synth(n, m, d, 0)
        self.model = DistributedModel(grid, None, MaternCovariance(2), None, self.theta, "zero")
        self.out = None

    def sync(self):
        self.torch.cuda.synchronize()

    def step(self, shared_factor):
        self.model._cache = None
        zpm, zpv = self.model.predict(self.xi, self.zi, self.xt)
        if not shared_factor:
            self.model._cache = None
        nll = self.model.negative_log_likelihood_zero_mean(self.theta, self.xi, self.zi)
        self.out = (zpm, zpv, float(nll))

    def check(self):
        """rank 0: the same step on ITS OWN GPU alone (the single-GPU product path), deviations of the distributed results"""
        zpm, zpv, nll = self.out
        res = None
        if self.rank == 0:
            import gpmp_amd as gp
            from gpmp_amd.kernel import MaternCovariance

            self.model._cache = None
            self.torch.cuda.empty_cache()
            ref = gp.Model(None, MaternCovariance(2), None, self.theta, "zero")
            rpm, rpv = ref.predict(self.xi, self.zi, self.xt)
            rnll = float(ref.negative_log_likelihood_zero_mean(self.theta, self.xi, self.zi))
            res = {"max_abs_dmean": float(np.max(np.abs(zpm - rpm))), "max_abs_dvar": float(np.max(np.abs(zpv - rpv))),
                   "nll_block_cyclic": nll, "nll_single_gpu": rnll, "nll_rel_diff": abs(nll - rnll) / abs(rnll),
                   "against": "rank 0's single-GPU Model.predict / NLL on the same inputs"}
            res["ok"] = bool(res["max_abs_dmean"] < 1e-6 and res["max_abs_dvar"] < 1e-6 and res["nll_rel_diff"] < 1e-9)
        return res


def strong_block_cyclic_extra(world, rank, grid, tmax, res):
    import torch.distributed as dist

    n, m, d = 32768, 50000, 8
    if os.environ.get("GPMP_BENCH_STRONG_NM"):
        n, m = (int(v) for v in os.environ["GPMP_BENCH_STRONG_NM"].split(","))
    runner = _BlockCyclicHeadline(grid, n, m, d, rank)
    out = {"n": n, "m_total": m, "d": d, "grid": f"{grid.pr}x{grid.pc}", "block": runner.model.nb,
           "note": "fixed total work: the single-GPU headline step (predict at m points + one NLL, inputs as host arrays, results as "
                   "host arrays on every rank) with K 2-D block-cyclic over the grid; compare with the N = 1 headline ms_per_step"}
    out.update(time_block_cyclic_strong_scaling(runner, dist, world, rank, 3, tmax))
    return out


def dist_extra_worker(args):
    """One rank of the distributed extra's own process group (started by spawn_ranks).  Exit code 0 / 4; a hang is the
    launcher's business (it kills the group at its time limit)."""
    world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    if world != args.gpus:
        sys.stderr.write(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}\n")
        return 2
    res = _Progress(args.progress if rank == 0 else None)
    res.update({"status": "started", "phase": "setup: process group"})
    try:
        stub = _stub_module()
        if stub is not None:
            import torch.distributed as dist

            dist.init_process_group(os.environ.get("GPMP_BENCH_BACKEND", "gloo"))
            res["ranks_seen"] = gather_ranks_seen(dist, rank, world, dist.get_backend(), False)
            stub.dist_extra(world, rank, res)
        else:
            import torch
            import torch.distributed as dist

            backend = os.environ.get("GPMP_BENCH_BACKEND", "nccl")
            if backend != "nccl":
                local_rank %= max(torch.cuda.device_count(), 1)
                os.environ["LOCAL_RANK"] = str(local_rank)
            torch.cuda.set_device(local_rank)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            res["phase"] = "setup: all-gather of the rank records"
            res["ranks_seen"] = gather_ranks_seen(dist, rank, world, backend, True)
            dist_potrf_extra(world, rank, res)
        res["status"] = "ok"
        dist.barrier()
        dist.destroy_process_group()
        return 0
    except BaseException as e:     # noqa: BLE001 -- recorded, then exit code 4 (the peers are ended by the launcher)
        res["error"] = f"{type(e).__name__}: {e}"[:400]
        res["status"] = "error"
        sys.stderr.write(f"[bench dist-extra rank {rank}] {type(e).__name__}: {e}\n")
        sys.stderr.flush()
        os._exit(EXIT_ERROR)       # collectives may be wedged: no orderly teardown


# ------------------------------------------------------------------------------------------------------------------
# the headline worker (one per GPU)
# ------------------------------------------------------------------------------------------------------------------
# the dominant kernel as rocprofv3 names it, up to the tile-width template argument (prefix match: every instance)
DOMINANT_KERNEL = "gemm_f64_kernel_v2<true, false, true"
METRIC = "fp64 predict+NLL throughput (points/s) and potrf TFLOP/s vs roofline, n=32k"


def _stub_module():
    """GPMP_BENCH_STUB_MODULE=<module>: launcher / process-group tests on machines without a GPU (tests/bench_stub.py).  The
    module supplies Workload and dist_extra; its line says "data": "stub".  Never set in a measurement."""
    name = os.environ.get("GPMP_BENCH_STUB_MODULE")
    if not name:
        return None
    import importlib

    return importlib.import_module(name)


class HipWorkload:
    """The measured workload: Model.predict + zero-mean NLL through libgpmp_hip.so, inputs resident in HBM."""

    data = "synthetic"

    def __init__(self, args, rank, world):
        import torch

        import gpmp_amd as gp
        import gpmp_amd.num as gnp
        from gpmp_amd import _lib
        from gpmp_amd.kernel import MaternCovariance

        self.torch, self.gnp, self.args, self.rank, self.world = torch, gnp, args, rank, world
        self.lib = _lib.load()
        n, m, d = args.n, args.m, args.d
        xi_h, zi_h, xt_h, self.theta = synth(n, m, d, rank)
        self.xi, self.zi, self.xt = gnp.asarray(xi_h), gnp.asarray(zi_h), gnp.asarray(xt_h)   # resident in HBM before timing
        self.model = gp.Model(None, MaternCovariance(2), None, self.theta, "zero")
        self.out = None

    def device(self):
        return self.gnp._dev()

    def sync(self):
        self.torch.cuda.synchronize()

    def step(self, m=None):
        xt = self.xt if m is None else self.xt[:m]
        zpm, zpv = self.model.predict(self.xi, self.zi, xt, convert_in=False, convert_out=False)
        nll = self.model.negative_log_likelihood_zero_mean(self.theta, self.xi, self.zi)
        self.out = (zpm, zpv, nll)
        return self.out

    # HIP events on the launch stream around every launch of the DOMINANT kernel only (kind 9: the LDS-direct NN GEMM
    # of the n x m solve, 63 launches per step) inside the timed region; recording every kind (9000 events per step
    # around the small launches of the factorisations) costs 0.8 % of the step, so the other kinds are collected from
    # one extra, untimed step afterwards.
    DOMINANT = 1 << 9

    def timed_begin(self):
        if not self.args.no_kernel_events:
            self.lib.gpmp_profile_begin_kinds(self.DOMINANT)

    def timed_end(self):
        import ctypes

        table = (ctypes.c_double * 36)()
        self.lib.gpmp_profile_end(table)
        prof_timed = np.array(list(table)).reshape(12, 3)
        # diagnostics of the other kernels: one untimed step with every kind recorded, scaled to the timed step count
        self.lib.gpmp_profile_begin_kinds(0xFFFFFFFF & ~self.DOMINANT)
        self.step()
        self.sync()
        self.lib.gpmp_profile_end(table)
        self.prof = np.array(list(table)).reshape(12, 3) * self.args.steps
        self.prof[9] = prof_timed[9]

    def check(self):
        zpm, zpv, nll = self.out
        assert bool(self.torch.isfinite(zpm).all()) and bool((zpv >= 0).all()) and math.isfinite(float(nll))

    def report(self, line):
        """rank 0: roofline of the dominant kernel + diagnostics, from the events of the timed region"""
        torch, gnp, args = self.torch, self.gnp, self.args
        n, m, d, steps = args.n, args.m, args.d, args.steps
        prof = self.prof
        # ---- outside the timed region: the Cholesky alone (BASELINE metric also quotes potrf TFLOP/s)
        cov = self.model.covariance

        def _timed(fn, reps=2):
            best = float("inf")
            for _ in range(reps):
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t_)
            return best

        t_gram = _timed(lambda: cov.gram_lower(self.xi, self.theta))
        t_both = _timed(lambda: gnp.cholesky_factor(cov.gram_lower(self.xi, self.theta), overwrite=True))
        potrf_ms = 1e3 * (t_both - t_gram)
        # ---- roofline of the dominant kernel: the NN fp64 MFMA GEMM behind the n x m triangular solve.
        # Algorithmic flops routed through it per step: n^2 m for V = L^-1 K(xi, xt)
        # (+ n^2 for the NLL's single right-hand side); launches and time measured with HIP events
        # on the launch stream over the timed region.
        nn_cnt, nn_ms, nn_exec = prof[9]            # gemm_f64_kernel_v2<AKC=1,BKC=0>: the large trsm updates
        nn1_cnt, nn1_ms, nn1_exec = prof[1]         # register-staged kernel: K < 512 updates, diagonal-block products
        nt_cnt, nt_ms, nt_exec = prof[0] + prof[8]
        alg_solve = (float(n) * n * m + float(n) * n) * steps
        # Every flop of these launches is algorithmic: they are the plain rectangular products B2 -= L21 X1 of the
        # recursive solve (no triangular waste), so executed == algorithmic for THIS kernel.
        line["roofline"] = {
            "bound": "mfma", "kernel": "gemm_f64_kernel_v2<true, false, true, W> (trsm updates B2 -= L21 X1, K >= 512; W = tile width 128 / 112 / 96: "
                                       "ONE kernel template, its instances appear as separate rows in a rocprofv3 summary and are counted together here)",
            "achieved": nn_exec / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else None,
            "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": (nn_exec / (nn_ms * 1e-3) / 1e12) / FP64_MFMA_PEAK_TFLOPS if nn_ms > 0 else None,
            "traffic": pmc_traffic_per_launch(DOMINANT_KERNEL) if (n, m) == (32768, 50000) else None,
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
        line["extra"].update({
            "potrf": {"n": n, "ms": potrf_ms, "tflops": (float(n) ** 3 / 3.0) / (potrf_ms * 1e-3) / 1e12,
                      "frac_of_fp64_mfma_peak": (float(n) ** 3 / 3.0) / (potrf_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                      "note": "whole factorisation (diagonal blocks + panels + trailing updates), wall time outside the timed region"},
            "potrf_gemm_nt_events": {"sum_of_spans_ms_per_step": nt_ms / steps, "launches_per_step": nt_cnt / steps,
                                     "note": "look-ahead overlaps panel and trailing kernels on two streams: spans are not additive"},
            "potf2_diag_blocks": {"sum_of_spans_ms_per_step": prof[4][1] / steps, "launches_per_step": prof[4][0] / steps},
            "gram": {"ms_per_step": gram_ms / steps, "GBps_written": gram_bytes / (gram_ms * 1e-3) / 1e9 if gram_ms > 0 else None,
                     "frac_of_hbm_peak": (gram_bytes / (gram_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if gram_ms > 0 else None},
            "coldots": {"ms_per_step": prof[6][1] / steps},
            "nll": float(self.out[2]),
        })
        threads = _host_threads()
        if self.world == 1 and (n, m) == (32768, 50000) and not args.no_live_pmc and os.environ.get("GPMP_BENCH_LIVE_PMC", "1") != "0":
            # `traffic` measured in THIS run (two counter passes of one step each, ~1 min); on any failure the figure of the
            # latest committed passes stays, and the note says which it is
            self.out = None
            torch.cuda.empty_cache()
            live, note = pmc_traffic_live(DOMINANT_KERNEL, args)
            if live is not None:
                line["roofline"]["traffic_committed_passes"] = line["roofline"]["traffic"]
                line["roofline"]["traffic"], line["roofline"]["traffic_note"] = live, note
            else:
                line["roofline"]["traffic_note"] = f"live PMC passes failed ({note}); " + line["roofline"]["traffic_note"]
        if self.world == 1 and not args.no_extras:
            # configs[1] and configs[3] at their stated sizes, outside the timed region, each beside a same-size CPU figure
            self.out = None
            torch.cuda.empty_cache()
            line["extra"]["config2"] = config2_extra(self.model, d, threads, not args.no_cpu_baseline)
            line["extra"]["config4"] = config4_extra(threads, not args.no_cpu_baseline)
        if self.world == 1 and not args.no_cpu_baseline:      # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline_for_line(n, m, d, threads, args.cpu_m_sample)
            line["extra"]["potrf"]["host_potrf"] = line["cpu_baseline"].pop("host_potrf")

    def release(self):
        """Drop every device buffer (the distributed extra's processes are about to use this GPU)."""
        self.out = self.xi = self.zi = self.xt = None
        self.torch.cuda.empty_cache()


def headline_worker(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never a silent one-rank run that prints "n_gpus": 1 for a --gpus N request
        sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (self-launching) "
                         f"or under torch.distributed.run --nproc-per-node {args.gpus}\n")
        return 2
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    stub = _stub_module()
    # GPMP_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (RCCL refuses two ranks on one
    # GPU): the ranks share the GPUs there are.  The measured run is one rank per GPU over nccl (= RCCL).
    backend = os.environ.get("GPMP_BENCH_BACKEND", "gloo" if stub is not None else "nccl")
    import torch

    have_gpu = stub is None
    if have_gpu:
        if backend != "nccl":
            local_rank %= max(torch.cuda.device_count(), 1)
            os.environ["LOCAL_RANK"] = str(local_rank)       # (gpmp_amd.num picks its device from it)
        torch.cuda.set_device(local_rank)
    dist = None
    if dist_on:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world

    wl = (stub.Workload if stub is not None else HipWorkload)(args, rank, world)
    comm_dev = wl.device() if backend == "nccl" else "cpu"

    def barrier():
        wl.sync()
        if dist_on:
            dist.barrier()
        wl.sync()

    def max_over_ranks(v):
        if not dist_on:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for _ in range(args.warmup):
        wl.step()
    barrier()
    wl.timed_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    elapsed = time.perf_counter() - t0
    wl.timed_end()
    elapsed = max_over_ranks(elapsed)
    wl.check()

    # ---- who ran: one PID per rank (evidence that N processes, not one, produced the line)
    pids = [os.getpid()]
    if dist_on:
        pt = torch.zeros(world, dtype=torch.int64, device=comm_dev)
        pt[rank] = os.getpid()
        dist.all_reduce(pt, op=dist.ReduceOp.SUM)
        pids = [int(v) for v in pt.tolist()]

    # ---- strong scaling, outside the timed region: the SAME step with a fixed total of m points split over the ranks
    #      (K still factored on every rank: 2 potrf + n^2 m / N solve flops per rank)
    m = args.m
    strong = None
    if world > 1:
        from gpmp_amd.dist.predict import shard_bounds

        lo, hi = shard_bounds(m, world, rank)
        wl.step(hi - lo)
        barrier()
        reps = max(1, min(args.steps, 3))
        t0 = time.perf_counter()
        for _ in range(reps):
            wl.step(hi - lo)
        barrier()
        t_strong = max_over_ranks(time.perf_counter() - t0) / reps
        strong = {"m_total": m, "m_per_gpu": [shard_bounds(m, world, r)[1] - shard_bounds(m, world, r)[0] for r in range(world)],
                  "steps": reps, "ms_per_step": 1e3 * t_strong, "points_per_s": m / t_strong,
                  "note": "fixed total work: the m points of ONE GPU's headline step split over the ranks, K factored on every rank "
                          "(Amdahl bound at N = 8: 2 potrf = 372 ms replicated + 763 / N ms of solve)"}

    ranks_seen = gather_ranks_seen(dist if dist_on else None, rank, world, backend if dist_on else None, have_gpu)

    rc = 0
    line = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * m / (elapsed / args.steps)
        if strong is None:
            strong = {"m_total": m, "ms_per_step": ms_per_step, "points_per_s": value, "note": "N = 1: the headline step itself"}
        n, d = args.n, args.d
        line = {
            "metric": METRIC,
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": wl.data,
            "config": {"workload": f"d={d} Matern-5/2 anisotropic, n={n} train, m={m} test points per GPU, fp64 "
                                   f"predict (mean+variance) + one zero-mean NLL eval per step",
                       "n": n, "m_per_gpu": m, "d": d,
                       "parallelism": f"xt-sharded x{world} (m points per rank); K built and factored on EVERY rank (replicated, "
                                      f"no data-path collective) -- weak scaling is ~N x by construction, see extra.strong_scaling"},
            "ranks_seen": ranks_seen,
            "extra": {"worker_pids": pids, "backend": backend if dist_on else None, "strong_scaling": strong},
        }
        wl.report(line)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if os.environ.get("GPMP_BENCH_CHILD") == "1":
            print(json.dumps(line), flush=True)         # under the launcher: it attaches the distributed extra itself
        else:
            # N = 1, or started by torch.distributed.run: rank 0 is the coordinator.  It has left its process group and holds no device
            # buffer any more; the extra runs in N fresh child processes (children are started, nothing is exec'ed).
            wl.release()
            rc = finish_with_dist_extra(line, world)
    return rc


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
    ap.add_argument("--cpu-m-sample", type=int, default=2048, help="prediction points of the CPU baseline's bounded sample")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1] / configs[3] extras")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--role", choices=("auto", "headline", "dist-extra", "cpu-full"), default="auto", help="internal: set by the launcher")
    ap.add_argument("--progress", default=None, help="internal: progress file of the distributed extra")
    args = ap.parse_args()
    if args.role == "dist-extra":
        return dist_extra_worker(args)
    if args.role == "cpu-full":
        return cpu_full_child(args)
    if args.role == "auto" and args.gpus > 1 and "RANK" not in os.environ:
        argv = [a for a in sys.argv[1:]]
        return launcher_main(args, argv)
    return headline_worker(args)


if __name__ == "__main__":
    sys.exit(main())
