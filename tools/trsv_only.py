"""Single-vector triangular solves alone: one persistent launch vs the launch-per-block chain (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
for n in [int(a) for a in sys.argv[1:]] or [32768, 16384, 4096, 5000]:
    rng = np.random.default_rng(n)
    xi = gnp.asarray(rng.random((n, 8)))
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(8) / 8))))
    K = MaternCovariance(2).gram_lower(xi, theta)
    torch.diagonal(K).add_(1e-4)
    F = gnp.cholesky_factor(K, overwrite=True)
    for R in (1, 4, 8, 10, 16):
        B = gnp.asarray(rng.standard_normal((n, R)) if R > 1 else rng.standard_normal(n))
        out = {}
        for mode in ("0", "1"):
            os.environ["GPMP_TRSV_PERSIST"] = mode
            for trans in (False, True):
                X = F.solve_lower(B, trans=trans); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5): X = F.solve_lower(B, trans=trans)
                torch.cuda.synchronize()
                out[(mode, trans)] = ((time.perf_counter() - t0) / 5 * 1e3, X.clone())
        for trans in (False, True):
            a, b = out[("0", trans)], out[("1", trans)]
            err = float((a[1] - b[1]).abs().max() / a[1].abs().max())
            print(f"n={n:6d} R={R} trans={int(trans)}: chain {a[0]:7.3f} ms  persistent {b[0]:7.3f} ms  ({4.0 * n * n / b[0] / 1e9:6.2f} TB/s of L)  max rel diff {err:.1e}")
