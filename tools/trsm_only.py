"""Forward solve L^-1 K(xi, xt) and backward solve L^-T (.) alone (diagnostic; run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
rng = np.random.default_rng(1234)
xi = gnp.asarray(rng.random((n, 8))); xt = gnp.asarray(rng.random((m, 8)))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(8) / 8))))
cov = MaternCovariance(2)
F = gnp.cholesky_factor(cov.gram_lower(xi, theta), overwrite=True)
for rep in range(2):
    B = cov(xi, xt, theta); torch.cuda.synchronize()
    t0 = time.perf_counter(); V = F.solve_lower(B, overwrite=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"forward  trsm ms {1e3 * dt:8.2f}  {float(n) * n * m / dt / 1e12:5.1f} TFLOP/s")
    torch.cuda.synchronize(); t0 = time.perf_counter(); W = F.solve_lower(V, trans=True, overwrite=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"backward trsm ms {1e3 * dt:8.2f}  {float(n) * n * m / dt / 1e12:5.1f} TFLOP/s")
