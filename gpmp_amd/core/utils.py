"""Input contract of the Model entry points -- counterpart of gpmp/core/utils.py."""
from .. import num as gnp


def ensure_shapes_and_type(*, xi=None, zi=None, xt=None, convert: bool = True):
    """gpmp/core/utils.py:19-81: 2-D xi / xt, zi (n,) or (n,1) -> (n,), matching row / column counts."""
    if xi is not None:
        assert len(xi.shape) == 2, "xi should be a 2D array"
    if zi is not None:
        if len(zi.shape) == 2:
            assert zi.shape[1] == 1, "zi should only have one column if it's a 2D array"
            zi = zi.reshape(-1)
        else:
            assert len(zi.shape) == 1, "zi should be 1D or a 2D column array"
    if xt is not None:
        assert len(xt.shape) == 2, "xt should be a 2D array"
    if xi is not None and zi is not None:
        assert xi.shape[0] == zi.shape[0], "xi and zi must have the same number of rows"
    if xi is not None and xt is not None:
        assert xi.shape[1] == xt.shape[1], "xi and xt must have the same number of columns"
    if convert:
        if xi is not None:
            xi = gnp.asarray(xi)
        if zi is not None:
            zi = gnp.asarray(zi)
        if xt is not None:
            xt = gnp.asarray(xt)
    return xi, zi, xt


def validate_model_mean(meantype: str, mean, meanparam):
    """gpmp/core/utils.py:84-118."""
    if meantype not in {"zero", "parameterized", "linear_predictor"}:
        raise ValueError("meantype must be one of 'zero', 'parameterized', or 'linear_predictor'")
    if meantype == "zero" and mean is not None:
        raise ValueError("For meantype 'zero', mean must be None")
    if meantype in ["parameterized", "linear_predictor"] and not callable(mean):
        raise TypeError(
            "For meantype 'parameterized' or 'linear_predictor', mean must be a callable function"
        )


def mean_values(model, x, param):
    """``model.mean(x, param)`` as a device array.  Mean PARAMETERS reach this backend as host vectors (SciPy's iterate, a
    slice of it, ``model.meanparam``); user mean functions combine them with device arrays (``param * gnp.ones(...)``,
    examples/gpmp_example22_1d_interpolation_variation_ml.py:38-39), so they are moved to the device first."""
    import torch

    from .. import num as gnp

    if param is not None and not isinstance(param, torch.Tensor):
        param = gnp.asarray(param)
    return gnp.asarray(model.mean(x, param))
