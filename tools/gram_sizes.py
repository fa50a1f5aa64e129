"""Gram pass at several sizes: back-to-back launches timed with events (kernel time, no host latency).  n m d triples."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
cov = MaternCovariance(2)
for n, m, d in [(4096, 10000, 8), (4096, 4096, 8), (1024, 1024, 8), (16384, 16384, 8), (32768, 50000, 8)]:
    rng = np.random.default_rng(1)
    xi = gnp.asarray(rng.random((n, d))); xt = gnp.asarray(rng.random((m, d)))
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    for which in ("it", "ii_lower"):
        if which == "ii_lower" and n != m:
            continue
        f = (lambda: cov(xi, xt, theta)) if which == "it" else (lambda: cov.gram_lower(xi, theta))
        K = f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            K = f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        bytes_ = 8.0 * n * m * (0.5 if which == "ii_lower" else 1.0)
        print(f"gram {which:9s} n={n:6d} m={m:6d} d={d}: {ms:8.4f} ms per launch  {bytes_ / ms / 1e9:6.2f} TB/s written", flush=True)
        del K
