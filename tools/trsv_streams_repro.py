"""Eight one-launch solves in flight on eight streams, each behind a machine-filling GEMM (the scenario of
tests/test_hip_parity.py::test_single_vector_solves_eight_streams_in_flight), repeated; prints per-stream status."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpmp_amd.num as gnp
from gpmp_amd import _lib
from oracle import gp_oracle as orc
lib = _lib.load()
rng = np.random.default_rng(8)
facs, refs = [], []
for k, n in enumerate((2048, 3000, 4101, 1500)):
    x = rng.random((n, 3))
    K = orc.maternp_covariance(x, None, 2, np.array([0.0, 1.0, 0.7, 1.3])) + 1e-5 * np.eye(n)
    F = gnp.cholesky_factor(gnp.asarray(K))
    z = gnp.asarray(rng.standard_normal((n, 1 + k)))
    facs.append((F, z))
    refs.append((F.solve_lower(z).clone(), F.solve_lower(z, trans=True).clone()))
A = gnp.alloc_matrix(8192, 1024, zero=True)
Cs = [gnp.alloc_matrix(8192, 8192, zero=True) for _ in range(8)]
streams = [torch.cuda.Stream() for _ in range(8)]
print("stream handles", [hex(s.cuda_stream) for s in streams])
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    torch.cuda.synchronize()
    outs = []
    t0 = time.perf_counter()
    for s, st in enumerate(streams):
        F, z = facs[s % 4]
        with torch.cuda.stream(st):
            if os.environ.get("NO_GEMM") != "1":
                _lib.check(lib.gpmp_dgemm(0, 1, 8192, 8192, 1024, -1.0, gnp._ptr(A), gnp._ld(A), gnp._ptr(A), gnp._ld(A), 1.0, gnp._ptr(Cs[s]),
                                          gnp._ld(Cs[s]), 1, gnp._stream()), "gpmp_dgemm")
            outs.append(F.solve_lower(z, trans=(s >= 4)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = [bool(torch.equal(o, refs[s % 4][1 if s >= 4 else 0])) for s, o in enumerate(outs)]
    nan = [int(torch.isnan(o).sum()) for o in outs]
    stat = []
    for st in streams:
        v = ctypes.c_int(-1)
        lib.gpmp_solve_status(ctypes.c_void_p(st.cuda_stream), ctypes.byref(v)); stat.append(v.value)
    print(f"rep {rep}: {1e3*dt:.1f} ms  equal {ok}  nan-count {nan}  status {stat}")
