#!/usr/bin/env python3
"""Time the individual phases of predict / NLL at a given size through the C ABI (diagnostic)."""
import argparse, os, sys, time, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd import _lib
from gpmp_amd.kernel import MaternCovariance

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32768)
ap.add_argument("--m", type=int, default=50000)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
lib = _lib.load()
rng = np.random.default_rng(1234)
n, m, d = a.n, a.m, a.d
xi = gnp.asarray(rng.random((n, d))); xt = gnp.asarray(rng.random((m, d)))
zi = gnp.asarray(np.sin(2 * np.pi * gnp.to_np(xi)[:, 0]))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
cov = MaternCovariance(2)

def timed(fn, reps=a.reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), r

t, K = timed(lambda: cov.gram_lower(xi, theta)); print(f"gram_lower n={n}: {t*1e3:8.2f} ms  {8*n*n/2/t/1e12:.2f} TB/s written")
t, Kf = timed(lambda: cov(xi, None, theta)); print(f"gram_full  n={n}: {t*1e3:8.2f} ms  {8*n*n/t/1e12:.2f} TB/s written")
t, Kit = timed(lambda: cov(xi, xt, theta)); print(f"gram_it n={n} m={m}: {t*1e3:8.2f} ms  {8*n*m/t/1e12:.2f} TB/s written")
t, Dm = timed(lambda: gnp.scaled_distance(theta[1:], xi, xt)); print(f"scaled_distance n={n} m={m}: {t*1e3:8.2f} ms  {8*n*m/t/1e12:.2f} TB/s written"); del Dm
def fac():
    Kc = cov.gram_lower(xi, theta)
    return gnp.cholesky_factor(Kc, overwrite=True)
tg, _ = timed(lambda: cov.gram_lower(xi, theta))
t, F = timed(fac); tp = t - tg
print(f"potrf n={n}: {tp*1e3:8.2f} ms  {n**3/3/tp/1e12:.2f} TFLOP/s ({n**3/3/tp/1e12/78.6*100:.1f}% of 78.6)")
z1 = zi.reshape(-1, 1)
t, w = timed(lambda: F.solve_lower(zi)); print(f"trsv (m=1): {t*1e3:8.2f} ms")
def trsm():
    B = cov(xi, xt, theta)
    return F.solve_lower(B, overwrite=True)
tk, _ = timed(lambda: cov(xi, xt, theta))
t, V = timed(trsm); tt = t - tk
print(f"trsm n={n} m={m}: {tt*1e3:8.2f} ms  {n*n*m/tt/1e12:.2f} TFLOP/s algorithmic ({n*n*m/tt/1e12/78.6*100:.1f}%)")
t, D = timed(lambda: gnp.coldots(V, w.reshape(-1, 1))); print(f"coldots: {t*1e3:8.2f} ms  {8*n*m/t/1e12:.2f} TB/s read")
t, _ = timed(lambda: F.logdet()); print(f"logdet: {t*1e3:8.2f} ms")
if n <= 16384:
    t, T = timed(lambda: F.inverse_factor()); print(f"trtri: {t*1e3:8.2f} ms  {n**3/3/t/1e12:.2f} TFLOP/s")
    t, Ki = timed(lambda: F.inverse_lower(T)); print(f"lauum: {t*1e3:8.2f} ms  {n**3/3/t/1e12:.2f} TFLOP/s")
