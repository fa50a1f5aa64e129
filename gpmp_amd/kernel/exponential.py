"""Exponential kernel -- counterpart of gpmp/kernel/exponential.py (not on the hot path: elementwise exp on the device)."""
from .. import num as gnp


def exponential_kernel(h):
    """exponential.py:9-24: exp(-h) for distances h >= 0."""
    return gnp.exp(-gnp.asarray(h))
