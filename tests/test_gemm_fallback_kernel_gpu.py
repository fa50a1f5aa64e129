"""GPMP_GEMM_V2=0 -- the one kernel-level switch left (DESIGN.md section 4 "Switches"): every product on the register-staged
kernel, the fall-back should the LDS-direct kernel misbehave on some driver / firmware.  Read once per process, so it runs in a
child process: Cholesky, many-right-hand-side solve, single-vector solves and the four GEMM transposition cases against
NumPy / LAPACK."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np, scipy.linalg as sla
sys.path.insert(0, %r)
import torch
import gpmp_amd.num as gnp
from gpmp_amd import _lib
from oracle import gp_oracle as orc
lib = _lib.load()
def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
n, m = 3100, 700
rng = np.random.default_rng(5)
x = rng.random((n, 4))
K = orc.maternp_covariance(x, None, 2, np.array([0.0, 1.2, 1.0, 0.8, 1.1])) + 1e-6 * np.eye(n)
B = rng.standard_normal((n, m)); z = rng.standard_normal(n)
Lref = np.linalg.cholesky(K)
F, V = gnp.cholesky_factor_solve(gnp.asarray(K), gnp.asarray(B), overwrite=False)
L = np.tril(gnp.to_np(F.L))
assert rel(L, Lref) < 1e-10 and rel(L @ L.T, K) < 1e-14
assert rel(gnp.to_np(V), sla.solve_triangular(Lref, B, lower=True)) < 1e-9
w = gnp.to_np(F.solve_lower(gnp.asarray(z)))
assert rel(w, sla.solve_triangular(Lref, z, lower=True)) < 1e-9
a = gnp.to_np(F.solve(gnp.asarray(z)))
assert rel(a, np.linalg.solve(K, z)) < 1e-7
Ki = gnp.to_np(gnp.cholesky_inv(gnp.asarray(K)))
assert rel(Ki @ K, np.eye(n)) < 1e-7
for ta in (0, 1):
    for tb in (0, 1):
        for (M, N, Kk) in ((700, 900, 640), (260, 130, 1024), (1500, 128, 128)):
            A = rng.standard_normal((Kk, M) if ta else (M, Kk)); Bm = rng.standard_normal((N, Kk) if tb else (Kk, N)); C0 = rng.standard_normal((M, N))
            At, Bt, Ct = (gnp.as_matrix(gnp.asarray(v), copy=True) for v in (A, Bm, C0))
            _lib.check(lib.gpmp_dgemm(ta, tb, M, N, Kk, -1.5, gnp._ptr(At), gnp._ld(At), gnp._ptr(Bt), gnp._ld(Bt), 0.5, gnp._ptr(Ct), gnp._ld(Ct), 0, gnp._stream()), "gpmp_dgemm")
            ref = -1.5 * (A.T if ta else A) @ (Bm.T if tb else Bm) + 0.5 * C0
            assert rel(gnp.to_np(Ct), ref) < 1e-13, (ta, tb, M, N, Kk)
print("CHILD_OK")
""" % ROOT

SETTINGS = [{"GPMP_GEMM_V2": "0"}]


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: ",".join(f"{k[5:]}={v}" for k, v in s.items()))
def test_once_read_switches_in_a_child_process(setting):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env.update(setting)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (setting, r.stdout[-500:], r.stderr[-1500:])
