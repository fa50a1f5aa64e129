"""trtri + lauum alone (diagnostic; run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(1234)
xi = gnp.asarray(rng.random((n, 8)))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(8) / 8))))
F = gnp.cholesky_factor(MaternCovariance(2).gram_lower(xi, theta), overwrite=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); T = F.inverse_factor(); torch.cuda.synchronize(); t1 = time.perf_counter()
    Ki = F.inverse_lower(T); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("trtri ms %.2f (%.1f TF)  lauum ms %.2f (%.1f TF)" % (1e3 * (t1 - t0), n**3 / 3 / (t1 - t0) / 1e12, 1e3 * (t2 - t1), n**3 / 3 / (t2 - t1) / 1e12))
    del T, Ki
