"""A/B of one library switch that is read at every call, inside ONE process: potrf wall time (best and median of reps) for each
setting, interleaved.  usage: potrf_ab.py ENVVAR valueA valueB n1 n2 ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance

var, va, vb = sys.argv[1:4]
sizes = [int(a) for a in sys.argv[4:]]
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(8) / 8))))
cov = MaternCovariance(2)
for n in sizes:
    xi = gnp.asarray(np.random.default_rng(1234).random((n, 8)))
    res = {va: [], vb: []}
    for rep in range(9):
        for v in (va, vb):
            os.environ[var] = v
            K = cov.gram_lower(xi, theta); torch.cuda.synchronize()
            t0 = time.perf_counter(); F = gnp.cholesky_factor(K, overwrite=True); torch.cuda.synchronize()
            res[v].append(1e3 * (time.perf_counter() - t0))
    for v in (va, vb):
        r = sorted(res[v][1:])
        print(f"n={n:6d} {var}={v:>6s}: best {r[0]:8.3f} ms  median {r[len(r)//2]:8.3f} ms", flush=True)
