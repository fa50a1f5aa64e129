"""Host-side sanitizer run: the AddressSanitizer + UBSan build of the library
(`make -C gpmp_amd/csrc asan`, host halves instrumented, device code untouched) is loaded in a CHILD process with the
ASan runtime preloaded and driven through everything that runs without a GPU: symbol table, workspace-size queries over
a sweep of sizes (the layout arithmetic of every fused driver), argument validation of every entry point (error-string
plumbing, thread-local buffers), profiling begin / end.  Any ASan / UBSan report makes the child exit non-zero."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, os.environ["GPMP_ROOT"])
from gpmp_amd import _lib
lib = _lib.load()
assert _lib.LIB_PATH.endswith("libgpmp_hip_asan.so"), _lib.LIB_PATH
assert lib.gpmp_hip_abi_version() == 1
# workspace / layout queries
for n in (0, 1, 127, 128, 129, 1000, 1024, 1025, 4096, 32768, 131072):
    lib.gpmp_dinv_elems(n); lib.gpmp_coldots_ws_rows(n); lib.gpmp_nll_ws_elems(n)
    for m in (1, 77, 50000):
        lib.gpmp_predict_ws_elems(n, m)
    for q in (0, 1, 21, 71, 72):
        lib.gpmp_reml_ws_elems(n, q); lib.gpmp_loo_ws_elems(n, q)
        for d in (1, 8, 20, 64):
            lib.gpmp_nll_grad_ws_elems(n, d, q)
for nmax in (1, 128, 300, 1024, 1025):
    for B in (1, 7, 256):
        for q in (0, 3, 4, 7, 8):
            lib.gpmp_batch_ws_elems(nmax, 5, q, B, 1); lib.gpmp_batch_ws_elems(nmax, 5, q, B, 0)
for d in (1, 4, 8, 16, 20, 33, 64):
    lib.gpmp_grad_ws_elems(1000, d)
# argument validation of every entry point: NULL pointers / bad sizes must come back as negative codes with a message
N = None
th = _lib.host_vec([0.0, 0.1, 0.2])
calls = [
    lambda: lib.gpmp_matern_gram(N, N, 4, 4, 3, 2, N, 0, 0.0, 0, N, 4, N),
    lambda: lib.gpmp_matern_pairwise(N, N, 4, 3, 2, N, 0, N, N),
    lambda: lib.gpmp_scaled_distance(N, N, 4, 4, 3, N, N, 4, N),
    lambda: lib.gpmp_maternp_kernel(N, 10, 2, N, N),
    lambda: lib.gpmp_matern_gram_deriv(N, 4, 3, 2, N, 0, 0, N, 4, N),
    lambda: lib.gpmp_potrf_lower_async(N, 4, 4, N, N, N),
    lambda: lib.gpmp_potrf_trsm_lower_async(N, 4, 4, N, N, N, 2, 2, N),
    lambda: lib.gpmp_trsm_lower(N, 4, 4, N, N, 2, 2, 0, N, N),
    lambda: lib.gpmp_trsm_right_lower(N, 4, 4, N, N, 2, 4, N),
    lambda: lib.gpmp_trtri_diag_blocks(N, 4, 4, N, N),
    lambda: lib.gpmp_trtri_lower(N, 4, 4, N, N, 4, N),
    lambda: lib.gpmp_lauum_lower(N, 4, 4, N, 4, N),
    lambda: lib.gpmp_tril(N, 4, 4, N),
    lambda: lib.gpmp_symmetrize_from_lower(N, 4, 4, N),
    lambda: lib.gpmp_dgemm(0, 0, 4, 4, 4, 1.0, N, 4, N, 4, 0.0, N, 4, 0, N),
    lambda: lib.gpmp_coldots(N, 4, 4, 4, N, 0, 1, N, 4, N, N),
    lambda: lib.gpmp_logdet_chol(N, 4, 4, N, N),
    lambda: lib.gpmp_matern_grad_trace(N, 4, N, 4, 3, 2, N, 0, N, N, 0, 1, N, N, N),
    lambda: lib.gpmp_nll_zero_mean(N, N, 4, 3, 2, N, 0, N, N, N, N),
    lambda: lib.gpmp_predict_zero_mean(N, N, N, 4, 4, 3, 2, N, 0, 1, N, N, N, N, N),
    lambda: lib.gpmp_reml(N, N, N, 1, 10, 2, 0, 2, th, 0, N, N, N, N),
    lambda: lib.gpmp_nll_grad(N, N, N, 1, 10, 2, 0, 2, th, 0, N, N, N, N, N),
    lambda: lib.gpmp_loo(N, N, N, 1, 10, 2, 0, 2, th, 0, N, N, N, N, N, N),
    lambda: lib.gpmp_nll_grad_batch(N, 0, N, 0, N, 1, 0, 0, N, 10, 2, 4, 2, th, 0, 0, N, N, N, N, N),
    lambda: lib.gpmp_dist_diag_factor(N, 256, 256, N, N),
    lambda: lib.gpmp_dist_diag_factor(ctypes.c_void_p(8), 2048, 2048, ctypes.c_void_p(8), N),
    lambda: lib.gpmp_dist_panel_solve(N, 256, N, 10, 256, N, 256, N, N),
    lambda: lib.gpmp_dist_exchange_pack(N, 256, N, 256, 4096, 256, 2, 2, 0, 0, 0, 256, N),
    lambda: lib.gpmp_dist_exchange_unpack(N, 256, N, 256, 4096, 256, 2, 2, 0, 0, 0, 256, N),
    lambda: lib.gpmp_dist_trailing_update(N, 4096, 4096, 256, 2, 2, 0, 0, 0, N, 256, N, 256, 0, -1, -1, N),
    lambda: lib.gpmp_dist_trailing_update(N, 4096, 4096, 100, 2, 2, 0, 0, 0, N, 256, N, 256, 0, -1, -1, N),
    lambda: lib.gpmp_dist_step_shape(4096, 256, 2, 2, 0, 0, 99, N, N, N, N),
    lambda: lib.gpmp_profile_end(N),
    lambda: lib.gpmp_stream_create_reserving_cus(0, N),
]
for k, call in enumerate(calls):
    rc = call()
    assert rc < 0, (k, rc)
    msg = lib.gpmp_last_error()
    assert msg, k
# degenerate sizes return before touching any pointer
assert lib.gpmp_potrf_lower_async(ctypes.c_void_p(8), 0, 0, ctypes.c_void_p(8), ctypes.c_void_p(8), N) == 0
assert lib.gpmp_profile_begin() == 0
tab = (ctypes.c_double * 36)()
assert lib.gpmp_profile_end(tab) == 0 and sum(tab) == 0.0
assert lib.gpmp_hint_machine_busy(1) == 0 and lib.gpmp_hint_machine_busy(0) == 1
# the distributed step's layout arithmetic over a sweep (host only)
L_ = ctypes.c_long
for n in (1, 127, 1000, 4097, 131072):
    for nb in (128, 256, 1024):
        for (pr, pc) in ((1, 1), (2, 4), (3, 2)):
            for r in range(pr):
                for c in range(pc):
                    a, b = L_(0), L_(0)
                    assert lib.gpmp_dist_local_shape(n, nb, pr, pc, r, c, ctypes.byref(a), ctypes.byref(b)) == 0
                    for k in range(0, (n + nb - 1) // nb, 7):
                        assert lib.gpmp_dist_step_shape(n, nb, pr, pc, r, c, k, ctypes.byref(a), ctypes.byref(b), None, None) == 0
                        for rp in range(pr):
                            assert lib.gpmp_dist_exchange_rows(n, nb, pr, pc, rp, c, k) >= 0
# nothing to exchange / update: returns before any pointer or device is touched
assert lib.gpmp_dist_exchange_pack(N, 256, N, 256, 256, 256, 1, 1, 0, 0, 0, 256, N) == 0
assert lib.gpmp_dist_trailing_update(N, 256, 256, 256, 1, 1, 0, 0, 0, N, 256, N, 256, 0, -1, -1, N) == 0
# stream bookkeeping: releasing / destroying a stream the library holds nothing for is a no-op that makes no HIP call
for h in (None, ctypes.c_void_p(0x1000), ctypes.c_void_p(0x1000)):
    assert lib.gpmp_stream_release(h) == 0
assert lib.gpmp_stream_destroy(None) == 0
# per-device state table (round 4): 8 host threads x 8 made-up device ordinals, every thread sees one state object per
# ordinal, the table grows by 8 and shrinks back; no HIP call is made (the entries hold no stream)
assert lib.gpmp_device_state_count() == 0
assert lib.gpmp_debug_device_table_selftest(8, 8, 200) == 0
assert lib.gpmp_device_state_count() == 0
assert lib.gpmp_debug_device_table_selftest(0, 8, 1) < 0 and lib.gpmp_last_error()
# new reduction: argument validation + empty input
assert lib.gpmp_coldots_pair(N, 4, N, 4, 4, 4, N, N, N) < 0
assert lib.gpmp_coldots_pair(ctypes.c_void_p(8), 4, ctypes.c_void_p(8), 4, 4, 0, ctypes.c_void_p(8), ctypes.c_void_p(8), N) == 0
# a block size above 1024 is refused when the layout is made, not at step 0
assert lib.gpmp_dist_local_shape(4096, 2048, 1, 1, 0, 0, None, None) < 0
print("ASAN-CHILD-OK")
'''


def test_host_side_sanitizer_build_runs_clean():
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm clang here")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("ASan runtime not found")
    subprocess.run(["make", "-C", os.path.join(ROOT, "gpmp_amd", "csrc"), "-j8", "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "gpmp_amd", "libgpmp_hip_asan.so")
    env = dict(os.environ, LD_PRELOAD=rt, GPMP_HIP_LIB=lib, GPMP_ROOT=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ASAN-CHILD-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize("mode", ["all", "deep", "ints", "lds"])
def test_argument_probes_run_clean_under_the_host_sanitizers(mode):
    """tests/test_abi_cpu.py's probes (every pointer NULL; one NULL at a time; every integer negative / zero / huge; every leading
    dimension too small) against the ASan + UBSan build of the library: an argument check that reads through a bad pointer, or a
    size computation that overflows a signed integer before the check rejects it, is reported here."""
    import torch

    from tests.test_abi_cpu import _NULL_PROBE

    if torch.cuda.device_count() > 0 and mode != "all":
        pytest.skip("a GPU is present: host scratch must not be handed to kernels")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm clang here")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("ASan runtime not found")
    subprocess.run(["make", "-C", os.path.join(ROOT, "gpmp_amd", "csrc"), "-j8", "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=rt, GPMP_HIP_LIB=os.path.join(ROOT, "gpmp_amd", "libgpmp_hip_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, "-c", _NULL_PROBE, ROOT, mode], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REJECTED" in r.stdout and "NOT REJECTED" not in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
