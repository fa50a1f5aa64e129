"""Does a HIP graph shorten a launch-bound evaluation?  The fused zero-mean NLL (gpmp_nll_zero_mean: Gram, blocked Cholesky,
sweep, two reductions -- enqueue-only, one stream at these sizes) captured once with torch.cuda.CUDAGraph and replayed, against
the same call issued eagerly (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd import _lib

lib = _lib.load()
dev = gnp._dev()
for n, d in ((128, 4), (500, 4), (1000, 8)):
    rng = np.random.default_rng(n)
    x = torch.as_tensor(rng.random((n, d)), device=dev)
    z = torch.as_tensor(np.sin(3 * rng.random(n)), device=dev)
    th = _lib.host_vec(np.concatenate(([0.0], -np.log(0.5 * np.ones(d)))))
    ws = torch.empty(int(lib.gpmp_nll_ws_elems(n)), dtype=torch.float64, device=dev)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)

    def call():
        _lib.check(lib.gpmp_nll_zero_mean(gnp._ptr(x), gnp._ptr(z), n, d, 2, th, 0, gnp._ptr(ws), gnp._ptr(out), gnp._ptr(info),
                                          gnp._stream()), "gpmp_nll_zero_mean")

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ref = float(out.item())
        t0 = time.perf_counter()
        for _ in range(200):
            call()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 200 * 1e3
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            call()
        g.replay(); torch.cuda.synchronize()
        ok = abs(float(out.item()) - ref) <= 1e-12 * abs(ref)
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 200 * 1e3
        print(f"n={n:5d}: eager {eager:.3f} ms per evaluation (back to back, no host sync)   graph replay {graph:.3f} ms   same value: {ok}")
    except Exception as exc:     # capture refused (e.g. an allocation or a sync inside the call)
        print(f"n={n:5d}: eager {eager:.3f} ms   graph capture failed: {type(exc).__name__}: {str(exc)[:200]}")
