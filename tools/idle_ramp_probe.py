"""Does a kernel run slower right after the GPU has been idle?  One Gram launch (4096 x 10000, HBM-write bound, 0.10 ms back to back)
timed with events after a host-side pause of t microseconds behind a synchronize."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
cov = MaternCovariance(2)
n, m, d = 4096, 10000, 8
rng = np.random.default_rng(1)
xi = gnp.asarray(rng.random((n, d))); xt = gnp.asarray(rng.random((m, d)))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
for _ in range(5):
    K = cov(xi, xt, theta)
torch.cuda.synchronize()
for pause_us in (0, 20, 50, 100, 200, 500, 1000, 5000, 20000, 100000):
    ts = []
    for rep in range(7):
        torch.cuda.synchronize()
        t_end = time.perf_counter() + pause_us * 1e-6
        while time.perf_counter() < t_end:
            pass
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); K = cov(xi, xt, theta); e1.record(); K = cov(xi, xt, theta); e2.record()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    a = sorted(t[0] for t in ts)[len(ts) // 2]; b = sorted(t[1] for t in ts)[len(ts) // 2]
    print(f"idle {pause_us:7d} us: first launch {a:7.4f} ms, second {b:7.4f} ms (medians of 7)", flush=True)
