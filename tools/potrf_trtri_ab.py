#!/usr/bin/env python3
"""Factor + inverse factor: gpmp_potrf_trtri_lower_async (the leading block's inverse along the factorisation's tail) against
potrf followed by trtri, interleaved in ONE process (best / median of 7), and one REML value + gradient (config 4) either way.

    python tools/potrf_trtri_ab.py [n ...]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp
from gpmp_amd.core import gradients as G
from gpmp_amd.kernel import MaternCovariance

for n in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768]:
    d = 20
    rng = np.random.default_rng(1234)
    x = rng.random((n, d))
    z = np.sin(2 * np.pi * x[:, 0]) + x[:, 1:].sum(axis=1)
    theta = np.concatenate(([0.0], -np.log(0.5 + np.arange(d) / (d - 1.0))))
    xd, zd = gnp.asarray(x), gnp.asarray(z)
    cov = MaternCovariance(2)
    res = {"sequence": [], "along": []}
    for rep in range(8):
        for mode in ("sequence", "along"):
            K = cov.gram_lower(xd, theta)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            F = gnp.cholesky_factor(K, overwrite=True, with_inverse=(mode == "along"))
            T = F.inverse_factor()
            torch.cuda.synchronize()
            res[mode].append(1e3 * (time.perf_counter() - t0))
            del F, T, K
    for mode, v in res.items():
        v = sorted(v[1:])
        print(f"n={n:6d} potrf+trtri {mode:8s}: best {v[0]:8.2f} ms  median {v[len(v) // 2]:8.2f} ms", flush=True)
    crit = G.REMLAnalytic(gp.Model(lambda a, p: gnp.ones((a.shape[0], 1)), cov, None, theta))
    ev = {"sequence": [], "along": []}
    for rep in range(6):
        for mode in ("sequence", "along"):
            G.INVERSE_ALONG_FROM = 8192 if mode == "along" else 1 << 30
            torch.cuda.synchronize(); t0 = time.perf_counter()
            v, st = crit.value_and_state(theta, xd, zd)
            g = crit.gradient_from_state(st)
            torch.cuda.synchronize()
            ev[mode].append((1e3 * (time.perf_counter() - t0), v, g))
            del st
    a, b = ev["sequence"][-1], ev["along"][-1]
    assert a[1] == b[1] and np.array_equal(a[2], b[2]), "the two routes must give identical values and gradients"
    for mode, v in ev.items():
        t = sorted(e[0] for e in v[1:])
        print(f"n={n:6d} REML value+gradient {mode:8s}: best {t[0]:8.2f} ms  median {t[len(t) // 2]:8.2f} ms", flush=True)
