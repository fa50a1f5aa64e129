"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/gpmp_hip.h declares (no compute calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpmp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpmp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = _declared_symbols()
    for must in ("gpmp_matern_gram", "gpmp_potrf_lower_async", "gpmp_trsm_lower", "gpmp_coldots", "gpmp_logdet_chol",
                 "gpmp_trtri_lower", "gpmp_lauum_lower", "gpmp_matern_grad_trace", "gpmp_dgemm"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from gpmp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in gpmp_hip.h but not exported by libgpmp_hip.so"
    assert lib.gpmp_hip_abi_version() == 1
    # every declared symbol has a ctypes signature in the binding and vice versa
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_pure_host_queries():
    from gpmp_amd import _lib

    lib = _lib.load()
    assert lib.gpmp_dinv_elems(0) == 0
    assert lib.gpmp_dinv_elems(1) == 128 * 128
    assert lib.gpmp_dinv_elems(129) == 2 * 128 * 128
    assert lib.gpmp_coldots_ws_rows(10) >= 9
    assert lib.gpmp_grad_ws_elems(1000, 8) > 0


def test_bad_arguments_are_reported_not_executed():
    from gpmp_amd import _lib

    lib = _lib.load()
    rc = lib.gpmp_matern_gram(None, None, 4, 4, 3, 2, None, 0, 0.0, 0, None, 4, None)
    assert rc < 0 and b"argument" in lib.gpmp_last_error()
    rc = lib.gpmp_potrf_lower_async(None, 4, 4, None, None, None)
    assert rc == -1


def test_missing_library_fails_loudly(monkeypatch):
    from gpmp_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgpmp_hip.so")
    with pytest.raises(ImportError):
        _lib.load()


def test_config_rules():
    from gpmp_amd import config

    with pytest.raises(ValueError):
        config.set_dtype("float32")
    with pytest.raises(ValueError):
        config.set_backend("numpy")
    config.set_dtype("float64")
    assert config.get_backend() == "hip"


def test_backend_namespace_covers_the_reference_contract():
    """every public name of the reference's NumPy backend (tests/golden/ref_gnp_names.txt, written by make_fixtures.py)
    exists in gpmp_amd.num, except typing aliases and module objects"""
    import os

    import gpmp_amd.num as gnp

    here = os.path.dirname(os.path.abspath(__file__))
    names = [l.strip() for l in open(os.path.join(here, "golden", "ref_gnp_names.txt")) if l.strip()]
    not_api = {"Any", "ArrayLike", "Callable", "CriterionCallable", "Iterable", "LoaderLike", "NDArray", "Scalar", "Tuple", "Union",
               "numpy_backend", "os", "shared", "warnings"}
    missing = [n for n in names if n not in not_api and not hasattr(gnp, n)]
    assert len(names) > 100 and missing == []


def test_dist_layout_queries_match_the_block_cyclic_definition():
    """gpmp_dist_local_shape / gpmp_dist_step_shape / gpmp_dist_exchange_rows are host arithmetic (no HIP call): checked here
    against the definition -- global block (I, J) on rank (I mod Pr, J mod Pc) -- over ragged sizes and non-square grids"""
    import ctypes

    from gpmp_amd import _lib

    lib = _lib.load()
    L = ctypes.c_long
    for n, nb in ((1900, 128), (2000, 256), (1024, 1024), (131072, 1024), (700, 128), (200, 128)):
        nblk = (n + nb - 1) // nb
        bs = lambda I: min(nb, n - I * nb)  # noqa: E731
        for pr, pc in ((1, 1), (1, 2), (2, 1), (2, 2), (2, 4), (3, 2), (2, 3)):
            for r in range(pr):
                for c in range(pc):
                    rows, cols = L(-1), L(-1)
                    assert lib.gpmp_dist_local_shape(n, nb, pr, pc, r, c, ctypes.byref(rows), ctypes.byref(cols)) == 0
                    rb, cb = list(range(r, nblk, pr)), list(range(c, nblk, pc))
                    assert rows.value == sum(bs(I) for I in rb) and cols.value == sum(bs(J) for J in cb)
                    for k in sorted({0, 1, nblk // 2, nblk - 2, nblk - 1} & set(range(nblk))):
                        pr_rows, co_rows, r0, c0 = L(-1), L(-1), L(-1), L(-1)
                        assert lib.gpmp_dist_step_shape(n, nb, pr, pc, r, c, k, ctypes.byref(pr_rows), ctypes.byref(co_rows), ctypes.byref(r0),
                                                        ctypes.byref(c0)) == 0
                        assert pr_rows.value == sum(bs(I) for I in rb if I > k) and co_rows.value == sum(bs(J) for J in cb if J > k)
                        assert r0.value == sum(bs(I) for I in rb if I <= k) and c0.value == sum(bs(J) for J in cb if J <= k)
                        for rp in range(pr):
                            want = sum(bs(J) for J in cb if J > k and J % pr == rp)
                            assert lib.gpmp_dist_exchange_rows(n, nb, pr, pc, rp, c, k) == want, (n, nb, pr, pc, rp, c, k)
    # argument checks come back negative with a message, without touching a device
    assert lib.gpmp_dist_local_shape(100, 100, 1, 1, 0, 0, None, None) < 0 and lib.gpmp_last_error()
    assert lib.gpmp_dist_exchange_rows(1000, 128, 2, 2, 0, 5, 0) < 0
    assert lib.gpmp_dist_diag_msg_elems(1024) == 1024 * 1024 + 8 * 128 * 128 + 1 and lib.gpmp_dist_diag_msg_elems(200) == 200 * 208 + 2 * 128 * 128 + 1


_NULL_PROBE = r'''
import ctypes, sys
sys.path.insert(0, sys.argv[1])
from gpmp_amd import _lib
lib = _lib.load()
deep = sys.argv[2] == "deep"
ints = sys.argv[2] == "ints"
not_compute = {"gpmp_hip_abi_version", "gpmp_last_error", "gpmp_profile_begin", "gpmp_profile_begin_kinds", "gpmp_profile_end",
               "gpmp_stream_create_reserving_cus", "gpmp_stream_destroy", "gpmp_stream_release", "gpmp_hint_machine_busy",
               "gpmp_device_release", "gpmp_device_state_count", "gpmp_debug_device_table_selftest", "gpmp_coldots_ws_rows",
               "gpmp_dist_exchange_rows"}
dummies = []
calls = 0
for name, (res, args) in _lib.SIGNATURES.items():
    if name in not_compute or res is not ctypes.c_int:
        continue
    ptr_pos = [i for i, a in enumerate(args) if a is ctypes.c_void_p]
    # all: every pointer NULL; deep: one NULL at a time (the last pointer is the stream); ints: every size / leading dimension /
    # index in turn negative, zero and huge, all pointers at host scratch
    if ints:
        cases = [(set(), i, v) for i, a in enumerate(args) if a in (ctypes.c_int, ctypes.c_long)
                 for v in ((-1, 0, 2 ** 31 - 1) if a is ctypes.c_int else (-1, 0, 2 ** 62))]
    else:
        cases = [(set(ptr_pos), -1, 0)] if not deep else [({i}, -1, 0) for i in ptr_pos[:-1]]
    for nulls, ipos, ival in cases:
        vals = []
        for i, a in enumerate(args):
            if a is ctypes.c_void_p:
                if i in nulls:
                    vals.append(None)
                else:
                    buf = (ctypes.c_double * 8192)()
                    dummies.append(buf)
                    vals.append(ctypes.cast(buf, ctypes.c_void_p))
            elif a is ctypes.c_double:
                vals.append(1.0)
            elif i == ipos:
                vals.append(ival)
            elif a is ctypes.c_long:
                vals.append(256)
            else:
                vals.append(2)
        rc = getattr(lib, name)(*vals)
        calls += 1
        if rc > 0 or (not ints and (rc == 0 or not lib.gpmp_last_error())):     # (ints: an empty problem may be a no-op, status 0)
            print("NOT REJECTED", name, sorted(nulls), ipos, ival, rc)
            sys.exit(1)
if sys.argv[2] == "lds":
    # every leading dimension in turn SMALLER than the extent it strides over (4 against 12 rows / columns; the others 4096): an
    # ARGUMENT error, i.e. caught by the entry point's own checks before anything reaches HIP
    calls = 0
    for name, (res, args) in _lib.SIGNATURES.items():
        if name in not_compute or res is not ctypes.c_int or name == "gpmp_maternp_kernel":      # (its long is an element count)
            continue
        for pos, a0 in enumerate(args):
            if a0 is not ctypes.c_long:
                continue
            vals = []
            for i, a in enumerate(args):
                if a is ctypes.c_void_p:
                    buf = (ctypes.c_double * 8192)()
                    dummies.append(buf)
                    vals.append(ctypes.cast(buf, ctypes.c_void_p))
                elif a is ctypes.c_double:
                    vals.append(1.0)
                elif a is ctypes.c_long:
                    vals.append(4 if i == pos else 4096)
                else:
                    vals.append(128 if name.startswith("gpmp_dist") and i in (3, 4) else 12)
            rc = getattr(lib, name)(*vals)
            calls += 1
            if rc >= 0 or b"argument" not in lib.gpmp_last_error():
                print("NOT REJECTED", name, pos, rc, lib.gpmp_last_error())
                sys.exit(1)
print("REJECTED", calls)
'''


def _run_null_probe(mode):
    import subprocess
    import sys

    r = subprocess.run([sys.executable, "-c", _NULL_PROBE, ROOT, mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "REJECTED" in r.stdout and "NOT REJECTED" not in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return int(r.stdout.split()[-1])


def test_every_compute_entry_point_rejects_null_operands():
    """all pointers NULL, positive sizes: every compute entry point of the header returns a negative status with a message -- no
    dereference, no launch (run in a child process: a crash would be a finding, not the end of the test session)"""
    assert _run_null_probe("all") >= 35


def test_one_null_operand_at_a_time_is_rejected_without_a_device():
    """every pointer position NULL in turn, the others pointing at host scratch: still a negative status (an argument error, or
    'no device' from the first HIP call).  Only meaningful -- and only safe -- without a GPU: on a GPU box host scratch passed as a
    device operand could be launched on."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: host scratch must not be handed to kernels")
    assert _run_null_probe("deep") >= 140
    # ... and every integer argument in turn negative, zero and huge: a status <= 0, never a crash or a host loop over a bad count
    assert _run_null_probe("ints") >= 600
    # ... and every leading dimension in turn too small for its matrix: an argument error from the entry point's own checks
    assert _run_null_probe("lds") >= 45
