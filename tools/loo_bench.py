"""Leave-one-out (Model.loo) timing at large n: Gram + potrf + trtri (doubling) + column sums of squares."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
for n in [int(a) for a in sys.argv[1:]] or [16384, 32768]:
    d = 8
    rng = np.random.default_rng(1234)
    xi = gnp.asarray(rng.random((n, d)))
    zi = gnp.asarray(np.sin(2 * np.pi * gnp.to_np(xi)[:, 0]) + gnp.to_np(xi)[:, 1:].sum(axis=1))
    theta = np.concatenate(([0.0, np.log(1e-4)], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    model = gp.Model(None, MaternCovariance(2, noise=True), None, theta, "zero")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        zloo, sigma2loo, eloo = model.loo(xi, zi)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"loo n={n}: {dt * 1e3:.1f} ms  ({(2 * n ** 3 / 3) / dt / 1e12:.1f} TF over potrf + trtri);  rms loo error {float(np.sqrt(np.mean(np.asarray(gnp.to_np(eloo)) ** 2))):.3e}")
