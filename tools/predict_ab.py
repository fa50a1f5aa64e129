"""A/B of one library switch that is read at every call, inside ONE process: wall time of one prediction (best / median of reps),
settings interleaved.  usage: predict_ab.py ENVVAR valueA valueB m n1 n2 ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp

var, va, vb = sys.argv[1:4]
m = int(sys.argv[4])
d = 8
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
model = gp.Model(None, gp.kernel.MaternCovariance(2), None, theta, "zero")
for n in [int(a) for a in sys.argv[5:]]:
    rng = np.random.default_rng(1234)
    xi = rng.random((n, d)); zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1); xt = rng.random((m, d))
    xi, zi, xt = gnp.asarray(xi), gnp.asarray(zi), gnp.asarray(xt)
    res = {va: [], vb: []}
    for rep in range(9):
        for v in (va, vb):
            os.environ[var] = v
            torch.cuda.synchronize(); t0 = time.perf_counter()
            zpm, zpv = model.predict(xi, zi, xt, convert_in=False, convert_out=False)
            torch.cuda.synchronize()
            res[v].append(1e3 * (time.perf_counter() - t0))
    for v in (va, vb):
        r = sorted(res[v][1:])
        print(f"predict n={n:6d} m={m} {var}={v:>6s}: best {r[0]:8.3f} ms  median {r[len(r)//2]:8.3f} ms", flush=True)
