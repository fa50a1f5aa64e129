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
