"""Soak test of the one-launch single-vector solves: many repetitions, with and without a concurrent machine-filling GEMM,
every result compared bit for bit with the first one (diagnostic; exits non-zero on the first mismatch)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd import _lib
from gpmp_amd.kernel import MaternCovariance

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
bad = 0
for n in (8192, 5000, 32768):
    rng = np.random.default_rng(n)
    xi = gnp.asarray(rng.random((n, 8)))
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(8) / 8))))
    K = MaternCovariance(2).gram_lower(xi, theta)
    torch.diagonal(K).add_(1e-4)
    F = gnp.cholesky_factor(K, overwrite=True)
    z = gnp.asarray(rng.standard_normal((n, 2)))
    ref_f, ref_b = F.solve_lower(z).clone(), F.solve_lower(z, trans=True).clone()
    A = gnp.alloc_matrix(16384, 1024, zero=True)
    C = gnp.alloc_matrix(16384, 16384, zero=True)
    side = torch.cuda.Stream()
    t0 = time.perf_counter()
    for rep in range(reps if n < 30000 else reps // 10):
        if rep % 3 == 0:
            with torch.cuda.stream(side):
                lib.gpmp_dgemm(0, 1, 16384, 16384, 1024, -1.0, gnp._ptr(A), gnp._ld(A), gnp._ptr(A), gnp._ld(A), 1.0, gnp._ptr(C), gnp._ld(C), 1, gnp._stream())
        w = z.clone()
        if rep % 2:
            _ = float(w.sum())
        gf, gb = F.solve_lower(w), F.solve_lower(w, trans=True)
        if not (torch.equal(gf, ref_f) and torch.equal(gb, ref_b)):
            bad += 1
            print(f"MISMATCH n={n} rep={rep}: fwd {float((gf - ref_f).abs().max()):.3e} bwd {float((gb - ref_b).abs().max()):.3e}", flush=True)
            if bad > 5:
                sys.exit(1)
    torch.cuda.synchronize()
    print(f"n={n}: {reps if n < 30000 else reps // 10} x (forward + transposed) ok in {time.perf_counter() - t0:.1f} s, mismatches so far {bad}", flush=True)
sys.exit(1 if bad else 0)
