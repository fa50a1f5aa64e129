"""Prediction time by mean type at one size (diagnostic): zero mean, constant mean, linear mean (q = d + 1 columns)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
d = 8
rng = np.random.default_rng(1234)
xi = rng.random((n, d)); zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1); xt = rng.random((m, d))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
const = lambda x, prm: gnp.ones((x.shape[0], 1))  # noqa: E731
lin = lambda x, prm: gnp.hstack((gnp.ones((x.shape[0], 1)), gnp.asarray(x)))  # noqa: E731
xi, zi, xt = gnp.asarray(xi), gnp.asarray(zi), gnp.asarray(xt)
only = sys.argv[3] if len(sys.argv) > 3 else None
for name, model in (("zero", gp.Model(None, gp.kernel.MaternCovariance(2), None, theta, "zero")),
                    ("constant", gp.Model(const, gp.kernel.MaternCovariance(2), None, theta)),
                    ("linear", gp.Model(lin, gp.kernel.MaternCovariance(2), None, theta))):
    if only and name != only:
        continue
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        zpm, zpv = model.predict(xi, zi, xt, convert_in=False, convert_out=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-9s mean: predict n=%d m=%d: %.1f ms" % (name, n, m, 1e3 * dt))
