import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import gpmp_amd.num as gnp
from gpmp_amd.kernel import MaternCovariance
for n in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(1234)
    xi = gnp.asarray(rng.random((n, 8)))
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(8) / 8))))
    F = gnp.cholesky_factor(MaternCovariance(2).gram_lower(xi, theta), overwrite=True)
    res = {"0": [], "1": []}
    Ts = {}
    for rep in range(7):
        for v in ("0", "1"):
            os.environ["GPMP_TRTRI_NN"] = v
            torch.cuda.synchronize(); t0 = time.perf_counter(); T = F.inverse_factor(); torch.cuda.synchronize()
            res[v].append(1e3 * (time.perf_counter() - t0)); Ts[v] = T
    d = float((Ts["0"] - Ts["1"]).abs().max() / Ts["0"].abs().max())
    for v in ("0", "1"):
        r = sorted(res[v][1:]); print(f"trtri n={n} GPMP_TRTRI_NN={v}: best {r[0]:.3f} ms median {r[len(r)//2]:.3f} ms ({n**3/3/r[0]/1e9:.1f} TF)")
    print("  max rel diff between the two forms:", d)
