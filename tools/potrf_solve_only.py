"""Factor + forward solve in one call (gpmp_potrf_trsm_lower_async), diagnostic timing."""
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
for rep in range(3):
    K = cov.gram_lower(xi, theta); B = gnp.as_matrix(cov(xi, xt, theta)); torch.cuda.synchronize()
    t0 = time.perf_counter(); F, V = gnp.cholesky_factor_solve(K, B); torch.cuda.synchronize()
    print("potrf + solve ms %.2f" % (1e3 * (time.perf_counter() - t0)))
    del K, B, F, V
