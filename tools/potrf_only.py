import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rng = np.random.default_rng(1234)
xi = gnp.asarray(rng.random((n, 8)))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(8) / 8))))
cov = MaternCovariance(2)
for rep in range(2):
    K = cov.gram_lower(xi, theta); torch.cuda.synchronize()
    t0 = time.perf_counter(); F = gnp.cholesky_factor(K, overwrite=True); torch.cuda.synchronize()
    print("potrf ms", 1e3 * (time.perf_counter() - t0))
