#!/usr/bin/env python3
"""Config 4 (BASELINE.json): REML hyper-parameter fit -- NLL/REML value + analytic gradient evaluations at
n = 16384, d = 20 anisotropic length-scales, fp64, one MI355X.  Times (i) 50 fixed-theta value+gradient
evaluations and (ii) a real SciPy L-BFGS-B run capped at 50 evaluations."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--d", type=int, default=20)
ap.add_argument("--evals", type=int, default=50)
ap.add_argument("--criterion", default="reml")
a = ap.parse_args()
n, d = a.n, a.d
rng = np.random.default_rng(1234)
xi = rng.random((n, d)); zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1) + 0.05 * rng.standard_normal(n)
theta0 = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
xi_t, zi_t = gnp.asarray(xi), gnp.asarray(zi)

def constant_mean(x, param):
    return gnp.ones((x.shape[0], 1))

if a.criterion == "reml":
    model = gp.Model(constant_mean, MaternCovariance(2))
    crit = gp.kernel.negative_log_restricted_likelihood
else:
    model = gp.Model(None, MaternCovariance(2), None, None, "zero")
    crit = gp.kernel.negative_log_likelihood_zero_mean
_, pre, nograd, grad = gp.kernel.make_selection_criterion_with_gradient(model, crit, xi_t, zi_t)
thetas = [theta0 + 0.1 * np.random.default_rng(1234 + k).standard_normal(d + 1) for k in range(a.evals)]
v = pre(thetas[0]); g = grad(thetas[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
tv = tg = 0.0
for th in thetas:
    torch.cuda.synchronize(); s = time.perf_counter(); v = pre(th); torch.cuda.synchronize(); tv += time.perf_counter() - s
    s = time.perf_counter(); g = grad(th); torch.cuda.synchronize(); tg += time.perf_counter() - s
tot = time.perf_counter() - t0
print(f"{a.criterion} n={n} d={d}: {a.evals} value+grad evals in {tot:.2f} s  ({1e3*tot/a.evals:.1f} ms/eval: value {1e3*tv/a.evals:.1f} ms, gradient {1e3*tg/a.evals:.1f} ms)")
print(f"  flops/eval ~ n^3/3 (potrf) + n^3/3 (trtri) + n^3/3 (lauum) = {n**3/1e12:.2f} TFLOP -> {n**3/1e12/(tot/a.evals):.1f} TFLOP/s")
print("  last value", v, "grad[:4]", g[:4])
# real optimiser run
from scipy.optimize import minimize
hist = []
def f(p):
    val = pre(p); hist.append(val); return val
t0 = time.perf_counter()
r = minimize(f, theta0, jac=grad, method="L-BFGS-B", options=dict(maxfun=a.evals, maxcor=20, ftol=1e-6, gtol=1e-5, maxls=40))
torch.cuda.synchronize()
print(f"  L-BFGS-B (maxfun={a.evals}): {len(hist)} evals in {time.perf_counter()-t0:.2f} s, criterion {hist[0]:.4f} -> {r.fun:.4f}, status: {r.message}")

# ---- breakdown of one gradient evaluation
from gpmp_amd.core import gradients as G
from gpmp_amd.core.linalg import covariance_factor, MeanSpace
def tm(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); s = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - s)
    return best * 1e3, r
th = thetas[0]
t, F = tm(lambda: covariance_factor(model, xi_t, th)); print(f"  gram_lower + potrf: {t:.1f} ms")
if a.criterion == "reml":
    P = constant_mean(xi_t, None)
    t, ms = tm(lambda: MeanSpace(F, zi_t, P)); print(f"  MeanSpace (L^-1 [z,P], Gram of W): {t:.1f} ms")
    t, X = tm(lambda: F.solve_lower(ms.W, trans=True)); print(f"  L^-T W (2 cols): {t:.1f} ms")
t, T = tm(lambda: F.inverse_factor()); print(f"  trtri: {t:.1f} ms")
t, Kinv = tm(lambda: F.inverse_lower(T)); print(f"  lauum: {t:.1f} ms")
alpha = F.solve(zi_t).reshape(-1, 1)
t, g = tm(lambda: G._grad_trace(model.covariance, Kinv, xi_t, th, alpha, alpha)); print(f"  grad_trace (r=1): {t:.1f} ms  ({4*n*n/t/1e9:.2f} TB/s of K^-1 lower read)")
t, _ = tm(lambda: F.logdet()); print(f"  logdet: {t:.2f} ms")
