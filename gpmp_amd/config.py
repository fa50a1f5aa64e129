"""Process-global configuration: counterpart of gpmp/config.py:94-236 for the "hip" backend.

Same rules as the reference: float64 only (float32 is rejected, gpmp/config.py:59-78), backend
name fixed before gpmp_amd.num is imported, logger named after the package with the level taken
from GPMP_LOG_LEVEL (gpmp/config.py:111-117).  The only backend here is "hip".
"""
import logging
import os
import re


_WIDTH = re.compile(r"(?:float|f|fp)?(16|32|64|128)$|f(2|4|8|16)$")


def _normalize_dtype_spec(dtype) -> str:
    """Only IEEE double is a legal working precision (the reference's rule, gpmp/config.py:59-78: single
    precision is refused with its own message, anything that is not recognisably double is refused too).
    Accepts the spellings users pass around: "float64", "torch.float64", numpy / torch dtype objects, "double",
    "f8", the Python ``float`` type, or None (= default)."""
    if dtype is None or dtype is float:
        return "float64"
    name = (dtype if isinstance(dtype, str) else str(dtype)).strip().lower().rsplit(".", 1)[-1]
    name = name.strip("<>'\" ")
    bits = None
    if name == "double":
        bits = 64
    elif name in ("single", "half"):
        bits = 32 if name == "single" else 16
    else:
        m = _WIDTH.search(name)
        if m:
            bits = int(m.group(1)) if m.group(1) else 8 * int(m.group(2))
    if bits == 32:
        raise ValueError("GPmp supports float64 only (float32 is not supported).")
    if bits != 64:
        raise ValueError(f"dtype {dtype!r} does not resolve to float64")
    return "float64"


def _normalize_backend_spec(backend):
    if backend is None:
        return None
    if not isinstance(backend, str):
        raise ValueError("backend must be a string")
    b = backend.lower()
    if b != "hip":
        raise ValueError("backend must be 'hip' (the numpy/torch backends live in the reference package)")
    return b


def _package_logger():
    """One stream handler per process, level from GPMP_LOG_LEVEL (the knob of gpmp/config.py:111-117; this
    package stays quiet by default and reports library loading / fallbacks of the fast paths at DEBUG)."""
    log = logging.getLogger("gpmp_amd")
    wanted = os.environ.get("GPMP_LOG_LEVEL", "WARNING").strip().upper()
    level = logging.getLevelName(wanted)
    log.setLevel(level if isinstance(level, int) else logging.WARNING)
    if not any(isinstance(h, logging.StreamHandler) for h in log.handlers):
        handler = logging.StreamHandler()
        handler.setFormatter(logging.Formatter("gpmp_amd %(levelname)s: %(message)s"))
        log.addHandler(handler)
    return log


class _GPMPConfig:
    def __init__(self):
        self.backend = "hip"
        self.dtype = _normalize_dtype_spec(os.environ.get("GPMP_DTYPE", "float64"))
        self.dtype_resolved = None
        self.device = None  # resolved lazily: cuda:<LOCAL_RANK or 0>
        self.seed = 1234
        self.caches = {}
        # device bytes one prediction chunk (n x m_chunk cross-covariance) may take
        self.predict_chunk_bytes = int(float(os.environ.get("GPMP_HIP_CHUNK_GB", "24")) * (1 << 30))
        self.logger = _package_logger()

    def clear_caches(self, name=None):
        """Drop one named cache (e.g. "gammaln"), or all of them."""
        doomed = list(self.caches) if name is None else [name]
        for key in doomed:
            self.caches.pop(key, None)


_config = _GPMPConfig()


def get_config():
    return _config


def init_backend():
    return _config.backend


def set_backend(backend: str):
    _config.backend = _normalize_backend_spec(backend)


def get_backend():
    return _config.backend


def set_dtype(dtype):
    _config.dtype = _normalize_dtype_spec(dtype)


def set_device(device):
    _config.device = device


def get_device():
    """torch device of this process: explicit set_device(), else cuda:$LOCAL_RANK (one process per GPU)."""
    import torch

    if _config.device is None:
        _config.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device(_config.device)


def clear_caches(name=None):
    _config.clear_caches(name)


def get_logger():
    return _config.logger


def set_log_level(level):
    _config.logger.setLevel(level)
