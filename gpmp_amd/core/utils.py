"""Input contract of the Model entry points -- counterpart of gpmp/core/utils.py.

Same observable behaviour as the reference (gpmp/core/utils.py:19-118): shape violations are
``AssertionError``s, an inconsistent mean specification is a ``ValueError`` / ``TypeError``; the checks
themselves are written as a small rule table instead of a chain of statements.
"""
from .. import num as gnp

_MEAN_TYPES = ("zero", "parameterized", "linear_predictor")


def _rank(a):
    return len(a.shape)


def _require(ok, what):
    if not ok:
        raise AssertionError(what)


def ensure_shapes_and_type(*, xi=None, zi=None, xt=None, convert: bool = True):
    """(xi, zi, xt) -> the same triple, checked and (optionally) moved to backend arrays.

    Rules (gpmp/core/utils.py:19-81): points are (n, d) / (m, d) matrices sharing d; observations are a
    vector of length n, a single-column matrix being flattened first."""
    have = {name: a is not None for name, a in (("xi", xi), ("zi", zi), ("xt", xt))}
    for name, pts in (("xi", xi), ("xt", xt)):
        if have[name]:
            _require(_rank(pts) == 2, f"{name} must be a 2-D array of points (got {_rank(pts)}-D)")
    if have["zi"]:
        r = _rank(zi)
        _require(r in (1, 2), "zi must be a vector or a one-column matrix")
        if r == 2:
            _require(zi.shape[1] == 1, f"zi has {zi.shape[1]} columns; only one is allowed")
            zi = zi.reshape(-1)
    if have["xi"] and have["zi"]:
        _require(xi.shape[0] == zi.shape[0], f"xi has {xi.shape[0]} rows but zi has {zi.shape[0]} values")
    if have["xi"] and have["xt"]:
        _require(xi.shape[1] == xt.shape[1], f"xi is {xi.shape[1]}-dimensional but xt is {xt.shape[1]}-dimensional")
    if not convert:
        return xi, zi, xt
    return tuple(gnp.asarray(a) if a is not None else None for a in (xi, zi, xt))


def validate_model_mean(meantype: str, mean, meanparam):
    """Constructor-time check of the mean specification (gpmp/core/utils.py:84-118): a "zero" model carries
    no mean function, the two other kinds need a callable; ``meanparam`` is not inspected here."""
    if meantype not in _MEAN_TYPES:
        raise ValueError(f"unknown meantype {meantype!r}: expected one of {', '.join(_MEAN_TYPES)}")
    if meantype == "zero":
        if mean is not None:
            raise ValueError("a 'zero' mean model takes mean=None")
    elif not callable(mean):
        raise TypeError(f"meantype {meantype!r} needs a callable mean(x, meanparam)")


def mean_values(model, x, param):
    """``model.mean(x, param)`` as a device array.  Mean PARAMETERS reach this backend as host vectors (SciPy's iterate, a
    slice of it, ``model.meanparam``); user mean functions combine them with device arrays (``param * gnp.ones(...)``,
    examples/gpmp_example22_1d_interpolation_variation_ml.py:38-39), so they are moved to the device first."""
    import torch

    if param is not None and not isinstance(param, torch.Tensor):
        param = gnp.asarray(param)
    return gnp.asarray(model.mean(x, param))
