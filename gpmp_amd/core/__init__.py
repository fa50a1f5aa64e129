"""gpmp_amd.core -- GP model facade and its numerical routines (gpmp/core counterpart)."""
from .model import Model

__all__ = ["Model"]
