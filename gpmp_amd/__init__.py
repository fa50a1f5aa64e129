"""gpmp_amd -- MI355X-native exact-GP inner loop behind GPmp's backend / Model surface.

Hot path only: Matern Gram build, fp64 Cholesky, triangular solves,
posterior mean / variance, LOO, ML / REML criteria and their analytic gradient, as hand-written
HIP kernels for gfx950 (gpmp_amd/csrc -> libgpmp_hip.so, C ABI in include/gpmp_hip.h).

Layout mirrors the reference package for the pieces on the path:
    gpmp_amd.config   <- gpmp/config.py      (backend name "hip", fp64-only rule)
    gpmp_amd.num      <- gpmp/num            (backend namespace on torch-ROCm tensors)
    gpmp_amd.kernel   <- gpmp/kernel         (maternp_covariance, REML selection driver)
    gpmp_amd.core     <- gpmp/core           (Model.predict / loo / negative_log_*)
"""
from . import config as config
from .core import Model

__version__ = "0.1.0"
__all__ = ["Model", "config", "num", "kernel", "core", "__version__"]


def __getattr__(name):
    if name in ("num", "kernel", "core", "dist", "dataloader"):
        import importlib

        mod = importlib.import_module(f"{__name__}.{name}")
        globals()[name] = mod
        return mod
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
