"""Latency of predict / NLL / REML+gradient at small n through the Python layer (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
from gpmp_amd.kernel import MaternCovariance

def constant_mean(x, param):
    import gpmp_amd.num as gnp
    return gnp.ones((x.shape[0], 1))

for n, m, d in ((50, 200, 2), (128, 500, 4), (500, 1000, 4), (1000, 2000, 8), (2000, 2000, 8)):
    rng = np.random.default_rng(n)
    xi, xt = rng.random((n, d)), rng.random((m, d))
    zi = np.sin(3 * xi[:, 0]) + xi.sum(axis=1)
    th = np.concatenate(([0.0], -np.log(0.5 * np.ones(d))))
    mz = gp.Model(None, MaternCovariance(2), None, th, "zero")
    mc = gp.Model(constant_mean, MaternCovariance(2), None, th)
    _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(mc, gp.kernel.negative_log_restricted_likelihood, xi, zi)
    def timeit(f, reps=20):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    tp = timeit(lambda: mz.predict(xi, zi, xt))
    tn = timeit(lambda: mz.negative_log_likelihood_zero_mean(th, xi, zi))
    tg = timeit(lambda: (pre(th), grad(th)))
    tc = timeit(lambda: mc.predict(xi, zi, xt))
    print(f"n={n:5d} m={m:5d} d={d}: predict(zero) {tp:6.2f} ms  predict(const mean) {tc:6.2f} ms  NLL {tn:6.2f} ms  REML value+grad {tg:6.2f} ms")
