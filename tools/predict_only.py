"""One prediction (diagnostic; run under rocprofv3 --kernel-trace and read with tools/step_timeline.py --predict)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
d = 8
rng = np.random.default_rng(1234)
xi = rng.random((n, d)); zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1); xt = rng.random((m, d))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
model = gp.Model(None, gp.kernel.MaternCovariance(2), None, theta, "zero")
xi, zi, xt = gnp.asarray(xi), gnp.asarray(zi), gnp.asarray(xt)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zpm, zpv = model.predict(xi, zi, xt, convert_in=False, convert_out=False)
    torch.cuda.synchronize()
    print("predict n=%d m=%d: %.3f ms" % (n, m, 1e3 * (time.perf_counter() - t0)))
