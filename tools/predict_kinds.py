"""Per-kind kernel time (the library's own HIP-event records) of `reps` predictions + NLL evaluations at one size.
usage: predict_kinds.py n m [reps]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd as gp
import gpmp_amd.num as gnp
from gpmp_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]); m = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
d = 8
rng = np.random.default_rng(1234)
xi = rng.random((n, d)); zi = np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1); xt = rng.random((m, d))
theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
model = gp.Model(None, gp.kernel.MaternCovariance(2), None, theta, "zero")
xi, zi, xt = gnp.asarray(xi), gnp.asarray(zi), gnp.asarray(xt)
names = ["gemm NT", "gemm NN (staged)", "gemm TN", "gemm TT", "potf2", "gram", "coldots", "trsv", "gemm NT v2", "gemm NN v2", "10", "11"]
for what in ("predict", "nll"):
    f = (lambda: model.predict(xi, zi, xt, convert_in=False, convert_out=False)) if what == "predict" else \
        (lambda: model.negative_log_likelihood_zero_mean(theta, xi, zi))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    lib.gpmp_profile_begin_kinds(0xFFFFFFFF)
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    table = (ctypes.c_double * 36)()
    lib.gpmp_profile_end(table)
    prof = np.array(list(table)).reshape(12, 3)
    print(f"{what} n={n} m={m}: {wall * 1e3:.3f} ms per call (un-profiled); per call by kind (launches, sum of kernel spans in ms):")
    for k in range(12):
        if prof[k][0] > 0:
            print(f"   {names[k]:18s} {prof[k][0] / reps:7.1f} launches  {prof[k][1] / reps:8.4f} ms")
