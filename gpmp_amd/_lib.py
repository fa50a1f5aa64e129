"""ctypes binding of libgpmp_hip.so (include/gpmp_hip.h).

The library is the product path: there is NO CPU fallback.  Importing this module never touches the
GPU; the first call that needs the library loads it and raises if it is missing.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPMP_HIP_LIB: another build of the same library (the host-sanitizer build `make -C gpmp_amd/csrc asan`)
LIB_PATH = os.environ.get("GPMP_HIP_LIB") or os.path.join(_HERE, "libgpmp_hip.so")

_lib = None

# name -> (restype, argtypes); mirrors include/gpmp_hip.h line by line
_P = c_void_p
SIGNATURES = {
    "gpmp_hip_abi_version": (c_int, []),
    "gpmp_last_error": (c_char_p, []),
    "gpmp_profile_begin": (c_int, []),
    "gpmp_profile_begin_kinds": (c_int, [ctypes.c_uint]),
    "gpmp_profile_end": (c_int, [_P]),
    "gpmp_stream_create_reserving_cus": (c_int, [c_int, _P]),
    "gpmp_stream_destroy": (c_int, [_P]),
    "gpmp_stream_release": (c_int, [_P]),
    "gpmp_hint_machine_busy": (c_int, [c_int]),
    "gpmp_matern_gram": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, c_double, c_int, _P, c_long, _P]),
    "gpmp_matern_pairwise": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "gpmp_scaled_distance": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_long, _P]),
    "gpmp_maternp_kernel": (c_int, [_P, c_long, c_int, _P, _P]),
    "gpmp_matern_gram_deriv": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _P, c_long, _P]),
    "gpmp_dinv_elems": (c_size_t, [c_int]),
    "gpmp_potrf_lower_async": (c_int, [_P, c_int, c_long, _P, _P, _P]),
    "gpmp_potrf_trsm_lower_async": (c_int, [_P, c_int, c_long, _P, _P, _P, c_int, c_long, _P]),
    "gpmp_trsm_lower": (c_int, [_P, c_int, c_long, _P, _P, c_int, c_long, c_int, _P, _P]),
    "gpmp_solve_status": (c_int, [_P, _P]),
    "gpmp_trsm_right_lower": (c_int, [_P, c_int, c_long, _P, _P, c_int, c_long, _P]),
    "gpmp_trtri_diag_blocks": (c_int, [_P, c_int, c_long, _P, _P]),
    "gpmp_trtri_lower": (c_int, [_P, c_int, c_long, _P, _P, c_long, _P]),
    "gpmp_lauum_lower": (c_int, [_P, c_int, c_long, _P, c_long, _P]),
    "gpmp_tril": (c_int, [_P, c_int, c_long, _P]),
    "gpmp_symmetrize_from_lower": (c_int, [_P, c_int, c_long, _P]),
    "gpmp_dgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_double, _P, c_long, _P, c_long, c_double, _P, c_long, c_int, _P]),
    "gpmp_coldots": (c_int, [_P, c_int, c_int, c_long, _P, c_int, c_long, _P, c_long, _P, _P]),
    "gpmp_coldots_ws_rows": (c_int, [c_int]),
    "gpmp_device_release": (c_int, []),
    "gpmp_device_state_count": (c_int, []),
    "gpmp_debug_device_table_selftest": (c_int, [c_int, c_int, c_int]),
    "gpmp_jacobi_sweep": (c_int, [_P, c_long, _P, c_long, c_int, c_double, _P, _P]),
    "gpmp_coldots_pair": (c_int, [_P, c_long, _P, c_long, c_int, c_int, _P, _P, _P]),
    "gpmp_logdet_chol": (c_int, [_P, c_int, c_long, _P, _P]),
    "gpmp_matern_grad_trace": (c_int, [_P, c_long, _P, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, c_long, _P, _P, _P]),
    "gpmp_matern_grad_trace_cross": (c_int, [_P, c_long, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, c_long, _P, _P, _P]),
    "gpmp_grad_ws_elems": (c_size_t, [c_int, c_int]),
    "gpmp_nll_ws_elems": (c_size_t, [c_int]),
    "gpmp_nll_zero_mean": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "gpmp_predict_ws_elems": (c_size_t, [c_int, c_int]),
    "gpmp_predict_zero_mean": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "gpmp_predict_mean_ws_elems": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "gpmp_predict_mean": (c_int, [_P, _P, _P, c_long, _P, _P, c_long, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "gpmp_reml_ws_elems": (c_size_t, [c_int, c_int]),
    "gpmp_reml": (c_int, [_P, _P, _P, c_long, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "gpmp_nll_grad_ws_elems": (c_size_t, [c_int, c_int, c_int]),
    "gpmp_nll_grad": (c_int, [_P, _P, _P, c_long, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P]),
    "gpmp_batch_ws_elems": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "gpmp_nll_grad_batch": (c_int, [_P, c_long, _P, c_long, _P, c_long, c_long, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int,
                                    _P, _P, _P, _P, _P]),
    "gpmp_loo_ws_elems": (c_size_t, [c_int, c_int]),
    "gpmp_dist_diag_msg_elems": (c_size_t, [c_int]),
    "gpmp_dist_diag_factor": (c_int, [_P, c_int, c_long, _P, _P]),
    "gpmp_dist_panel_ws_elems": (c_size_t, [c_int]),
    "gpmp_dist_panel_solve": (c_int, [_P, c_int, _P, c_int, c_long, _P, c_long, _P, _P]),
    "gpmp_dist_local_shape": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "gpmp_dist_step_shape": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "gpmp_dist_exchange_rows": (c_long, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "gpmp_dist_exchange_pack": (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "gpmp_dist_exchange_unpack": (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "gpmp_dist_trailing_update": (c_int, [_P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_long, _P, c_long, c_int, c_int,
                                          c_int, _P]),
    "gpmp_dist_inverse_gram": (c_int, [_P, c_long, _P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "gpmp_loo": (c_int, [_P, _P, _P, c_long, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, _P]),
}


class GpmpHipError(RuntimeError):
    """Raised when a libgpmp_hip entry point returns a non-zero status."""


def load():
    """Load libgpmp_hip.so (once) and declare every signature.  Fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gpmp_amd/csrc`). gpmp_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gpmp_last_error()
        raise GpmpHipError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def host_vec(values):
    """Small host parameter vector -> ctypes double array (kept alive by the caller)."""
    vals = [float(v) for v in values]
    return (c_double * len(vals))(*vals)
