#!/usr/bin/env python3
"""Many small problems: problems/s of the batched library call (gpmp_nll_grad_batch) against the one-problem-at-a-time
route (value + analytic gradient), zero-mean NLL, d = 4."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp
from gpmp_amd.core.gradients import MLZeroMeanAnalytic, batch_values_and_gradients

d = 4
th = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
model = gp.Model(None, gp.kernel.MaternCovariance(2), None, th, "zero")
ana = MLZeroMeanAnalytic(model)
print("n      B   batched ms   problems/s   one-at-a-time ms   problems/s   speed-up   | per-problem parameters (sampler pattern): ms   problems/s")
sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]
for n in sizes:
    for B in ((8, 64, 256) if n <= 1024 else (4, 16)):
        rng = np.random.default_rng(n + B)
        batches = []
        for b in range(B):
            x = rng.random((n, d))
            batches.append((gnp.asarray(x), gnp.asarray(np.sin(3 * x[:, 0]) + x.sum(axis=1))))
        batch_values_and_gradients(model, th, batches, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            vals, grads = batch_values_and_gradients(model, th, batches, True)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / 3
        nseq = min(B, 32)
        for xb, zb in batches[:2]:
            v, st = ana.value_and_state(th, xb, zb); ana.gradient_from_state(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for xb, zb in batches[:nseq]:
            v, st = ana.value_and_state(th, xb, zb)
            gseq = ana.gradient_from_state(st)
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / nseq * B
        tv, tg = (1e-10, 1e-8) if n <= 1024 else (1e-7, 1e-5)      # (above 1024 points the two routes block differently and K is ill-conditioned at d = 4)
        assert abs(vals[nseq - 1] - v) < tv * abs(v) and np.max(np.abs(grads[nseq - 1] - gseq)) < tg * np.max(np.abs(gseq))
        TH = th + 0.05 * rng.standard_normal((B, th.size))
        batch_values_and_gradients(model, TH, batches, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            vp, gp_ = batch_values_and_gradients(model, TH, batches, True)
        torch.cuda.synchronize()
        tp = (time.perf_counter() - t0) / 3
        print(f"{n:5d} {B:4d}   {1e3*tb:9.2f}   {B/tb:10.0f}   {1e3*ts:16.2f}   {B/ts:10.0f}   {ts/tb:7.1f}x   | {1e3*tp:9.2f}   {B/tp:10.0f}")
