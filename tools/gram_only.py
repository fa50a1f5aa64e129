"""Gram build alone (diagnostic; run under rocprofv3)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(1234)
xi = gnp.asarray(rng.random((n, d))); xt = gnp.asarray(rng.random((m, d)))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
cov = MaternCovariance(2)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = cov(xi, xt, theta); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("gram n=%d m=%d d=%d: %.2f ms  %.2f TB/s written" % (n, m, d, dt * 1e3, 8.0 * n * m / dt / 1e12))
    del K
