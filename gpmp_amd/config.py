"""Process-global configuration: counterpart of gpmp/config.py:94-236 for the "hip" backend.

Same rules as the reference: float64 only (float32 is rejected, gpmp/config.py:59-78), backend
name fixed before gpmp_amd.num is imported, logger named after the package with the level taken
from GPMP_LOG_LEVEL (gpmp/config.py:111-117).  The only backend here is "hip".
"""
import logging
import os


def _normalize_dtype_spec(dtype) -> str:
    """gpmp/config.py:59-78 -- anything that is not float64 is an error."""
    if dtype is None or dtype is float:
        return "float64"
    s = dtype.lower() if isinstance(dtype, str) else str(dtype).lower()
    if "float32" in s or s.endswith("f4") or s.endswith("32"):
        raise ValueError("GPmp supports float64 only (float32 is not supported).")
    if "float64" in s or "double" in s or s.endswith("f8") or s.endswith("64"):
        return "float64"
    raise ValueError("dtype must resolve to float64")


def _normalize_backend_spec(backend):
    if backend is None:
        return None
    if not isinstance(backend, str):
        raise ValueError("backend must be a string")
    b = backend.lower()
    if b != "hip":
        raise ValueError("backend must be 'hip' (the numpy/torch backends live in the reference package)")
    return b


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
        self.logger = logging.getLogger("gpmp_amd")
        if not self.logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter("[%(levelname)s] %(message)s"))
            self.logger.addHandler(h)
        self.logger.setLevel(getattr(logging, os.environ.get("GPMP_LOG_LEVEL", "WARNING").upper(), logging.WARNING))

    def clear_caches(self, name=None):
        if name is None:
            self.caches.clear()
        else:
            self.caches.pop(name, None)


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
