"""Remaining names of the backend contract (every public name of gpmp/num/numpy_backend.py): thin device-tensor
wrappers, none of them on the hot path.  Kept apart from gpmp_amd/num/__init__.py, which holds the names the reference's
core and kernel modules call; imported into the same namespace at the end of that module."""
import builtins

import numpy
import torch
from scipy.stats import multivariate_normal as scipy_mvnormal
from scipy.stats import norm as normal  # noqa: F401  (host-side distribution objects, as in the reference's NumPy backend)

from . import _dev, _host_params, asarray, derivative_finite_diff, scaled_distance, to_np
from . import svd as _svd

ndarray = torch.Tensor


def _t(x):
    return x if isinstance(x, torch.Tensor) else asarray(x)


# ---- shape / construction
def transpose(x, dim0, dim1):
    """numpy_backend.py:207-209 (torch-style: swap two dimensions)."""
    return torch.transpose(_t(x), dim0, dim1)


def expand_dims(x, axis):
    return torch.unsqueeze(_t(x), axis)


def tile(x, reps):
    reps = (reps,) if isinstance(reps, int) else tuple(reps)
    return _t(x).repeat(*reps) if _t(x).dim() <= len(reps) else _t(x).repeat(*((1,) * (_t(x).dim() - len(reps)) + reps))


def split(x, indices_or_sections, axis=0):
    x = _t(x)
    if isinstance(indices_or_sections, int):
        return list(torch.chunk(x, indices_or_sections, dim=axis))
    idx = [0] + [int(i) for i in indices_or_sections] + [x.shape[axis]]
    return [x.narrow(axis, a, b - a) for a, b in zip(idx[:-1], idx[1:])]


def meshgrid(*xs, indexing="xy"):
    return torch.meshgrid(*[_t(v) for v in xs], indexing=indexing)


def logspace(start, stop, num=50, endpoint=True, base=10.0, dtype=None, axis=0):
    return asarray(numpy.logspace(start, stop, num=num, endpoint=endpoint, base=base, dtype=numpy.float64, axis=axis))


def empty_like(x, dtype=None):
    return torch.empty_like(_t(x), dtype=dtype)


def zeros_like(x, dtype=None):
    return torch.zeros_like(_t(x), dtype=dtype)


def ones_like(x, dtype=None):
    return torch.ones_like(_t(x), dtype=dtype)


def full_like(x, fill_value, dtype=None):
    return torch.full_like(_t(x), fill_value, dtype=dtype)


def asint(x):
    return _t(x).to(torch.int64)


def isscalar(x):
    return numpy.isscalar(x) or (isinstance(x, torch.Tensor) and x.dim() == 0)


def get_dtype():
    return torch.float64


def init_backend():
    """gpmp/config.py:190-196: idempotent; this package has exactly one backend."""
    return "hip"


# ---- elementwise / logical
def _ew(torch_fn, numpy_fn):
    def f(x):
        if isinstance(x, torch.Tensor):
            return torch_fn(x)
        y = numpy_fn(numpy.asarray(x, dtype=numpy.float64))
        return float(y) if y.ndim == 0 else y
    return f


log10, log1p, tan = _ew(torch.log10, numpy.log10), _ew(torch.log1p, numpy.log1p), _ew(torch.tan, numpy.tan)
floor, ceil = _ew(torch.floor, numpy.floor), _ew(torch.ceil, numpy.ceil)


def clip(x, a_min=None, a_max=None):
    return torch.clamp(_t(x), min=a_min, max=a_max)


def nan_to_num(x, nan=0.0, posinf=None, neginf=None):
    return torch.nan_to_num(_t(x), nan=nan, posinf=posinf, neginf=neginf)


def logical_and(a, b):
    return torch.logical_and(_t(a), _t(b))


def logical_or(a, b):
    return torch.logical_or(_t(a), _t(b))


def logical_not(a):
    return torch.logical_not(_t(a))


def isclose(a, b, rtol=1e-05, atol=1e-08, equal_nan=False):
    a, b = _t(a), _t(b)
    return torch.isclose(a, b.to(a.dtype), rtol=rtol, atol=atol, equal_nan=equal_nan)


def allclose(a, b, rtol=1e-05, atol=1e-08, equal_nan=False):
    return bool(isclose(a, b, rtol, atol, equal_nan).all())


def array_equal(a, b):
    a, b = _t(a), _t(b)
    return a.shape == b.shape and bool((a == b).all())


# ---- reductions / statistics
def std(x, axis=None):
    return torch.std(_t(x), unbiased=False) if axis is None else torch.std(_t(x), dim=axis, unbiased=False)


def prod(x, axis=None):
    return torch.prod(_t(x)) if axis is None else torch.prod(_t(x), dim=axis)


def cumsum(x, axis=None):
    return torch.cumsum(_t(x).reshape(-1), 0) if axis is None else torch.cumsum(_t(x), dim=axis)


def argmax(x, axis=None):
    return torch.argmax(_t(x)) if axis is None else torch.argmax(_t(x), dim=axis)


def argmin(x, axis=None):
    return torch.argmin(_t(x)) if axis is None else torch.argmin(_t(x), dim=axis)


def percentile(x, q, axis=None):
    qq = asarray(numpy.asarray(q, dtype=numpy.float64) / 100.0).reshape(-1)
    out = torch.quantile(_t(x) if axis is not None else _t(x).reshape(-1), qq, dim=0 if axis is None else axis)
    return out[0] if numpy.isscalar(q) else out


def unique(x):
    return torch.unique(_t(x))


def cov(m, rowvar=True):
    m = _t(m)
    return torch.cov(m if rowvar else m.T)


def inner(a, b):
    return torch.inner(_t(a), _t(b))


def norm(x, ord=None, axis=None):  # noqa: A002
    """numpy.linalg.norm.  Vector norms and the Frobenius norm are elementwise reductions on the device; a matrix 2-norm / nuclear
    norm comes from the library's Jacobi SVD (square input); every other matrix norm is host LAPACK (never on the GP path)."""
    x = _t(x)
    vector = x.dim() == 1 or isinstance(axis, int)
    if (vector and ord in (None, 2)) or (not vector and ord in (None, "fro")):
        return torch.sqrt(torch.sum(x * x, dim=axis) if axis is not None else torch.sum(x * x))
    if vector and ord in (1, numpy.inf, float("inf")):
        a = torch.abs(x)
        red = torch.sum if ord == 1 else torch.amax
        return red(a, dim=axis) if axis is not None else red(a)
    if not vector and x.dim() == 2 and axis is None and ord in (2, -2, "nuc") and x.shape[0] == x.shape[1] and x.shape[0] > 0:
        sv = _svd(x)[1]
        return sv[0] if ord == 2 else (sv[-1] if ord == -2 else torch.sum(sv))
    return asarray(numpy.linalg.norm(to_np(x), ord=ord, axis=axis))


def cond(x, p=None):
    """numpy.linalg.cond: 2-norm condition number from the library's Jacobi SVD (square input); other norms: host LAPACK."""
    x = _t(x)
    if p in (None, 2) and x.dim() == 2 and x.shape[0] == x.shape[1] and x.shape[0] > 0:
        sv = _svd(x)[1]
        return sv[0] / sv[-1]
    return asarray(numpy.linalg.cond(to_np(x), p=p))


def cdist(x, y):
    """scipy.spatial.distance.cdist (Euclidean) on the HIP distance kernel: unit scales."""
    x = _t(x)
    return scaled_distance(numpy.zeros(x.shape[1]), x, _t(y))


# ---- Cholesky pairs (scipy.linalg.cho_factor / cho_solve signatures)
def cho_factor(A, lower=True, **_):
    """Returns (factor object, lower) -- opaque first element, to be passed to cho_solve."""
    from . import cholesky_factor

    return cholesky_factor(_t(A)), True


def cho_solve(c_and_lower, b, **_):
    F, _lower = c_and_lower
    return F.solve(_t(b))


# ---- finite-difference derivatives (numpy_backend.py:256-326)
def grad(f):
    """Gradient of a scalar function by 5-point central differences, h = 1e-5."""
    def grad_f(x, h=1e-5):
        x0 = numpy.array(_host_params(x), dtype=numpy.float64)
        g = numpy.zeros_like(x0)
        for i in range(x0.size):
            def f_i(v, i=i):
                xx = x0.copy()
                xx[i] = v
                y = f(xx)
                return float(y.item() if hasattr(y, "item") else y)
            g[i] = derivative_finite_diff(f_i, float(x0[i]), h)
        return g
    return grad_f


def value_and_grad(f, x, *, h=1e-5):
    y = f(x)
    return y, grad(f)(x, h)


# ---- random helpers on the host generator of this backend
_rng = numpy.random.default_rng(1234)


def choice(a, size=None, replace=True, p=None):
    a_ = to_np(a) if isinstance(a, torch.Tensor) else a
    out = _rng.choice(a_, size=size, replace=replace, p=None if p is None else to_np(p))
    return asarray(out) if isinstance(a, torch.Tensor) else out


def permutation(x):
    x_ = to_np(x) if isinstance(x, torch.Tensor) else x
    out = _rng.permutation(x_)
    return asarray(out) if isinstance(x, torch.Tensor) else out


class multivariate_normal:
    """numpy_backend.py:501-580: host-side Gaussian helpers (scalar or full covariance) returning backend arrays."""

    @staticmethod
    def _np(v):
        return to_np(v) if isinstance(v, torch.Tensor) else numpy.asarray(v, dtype=numpy.float64)

    @staticmethod
    def _mean(mean, d):
        m = multivariate_normal._np(mean)
        return numpy.full((d,), float(m)) if m.ndim == 0 else m.reshape(-1)

    @staticmethod
    def rvs(mean=0.0, cov=1.0, n=1):
        c = multivariate_normal._np(cov)
        if c.size == 1:
            return asarray(normal.rvs(float(multivariate_normal._np(mean).reshape(-1)[0]), float(numpy.sqrt(c.reshape(-1)[0])), size=n,
                                      random_state=_rng))
        if c.ndim != 2 or c.shape[0] != c.shape[1]:
            raise ValueError("cov must be a scalar or a square 2D matrix.")
        return asarray(numpy.asarray(scipy_mvnormal.rvs(mean=multivariate_normal._mean(mean, c.shape[0]), cov=c, size=n, random_state=_rng)))

    @staticmethod
    def logpdf(x, mean=0.0, cov=1.0):
        c, xx = multivariate_normal._np(cov), multivariate_normal._np(x)
        if c.size == 1:
            return asarray(numpy.asarray(normal.logpdf(xx, multivariate_normal._np(mean), numpy.sqrt(c.reshape(-1)[0]))))
        return asarray(numpy.asarray(scipy_mvnormal.logpdf(xx, mean=multivariate_normal._mean(mean, c.shape[0]), cov=c)))

    @staticmethod
    def cdf(x, mean=0.0, cov=1.0):
        c, xx = multivariate_normal._np(cov), multivariate_normal._np(x)
        if c.size == 1:
            return asarray(numpy.asarray(normal.cdf(xx, multivariate_normal._np(mean), numpy.sqrt(c.reshape(-1)[0]))))
        return asarray(numpy.asarray(scipy_mvnormal.cdf(xx, mean=multivariate_normal._mean(mean, c.shape[0]), cov=c)))


def try_with_postmortem(func, *args, **kwargs):
    """gpmp/num/shared.py:58-70: run, and open pdb at the failure point."""
    try:
        return func(*args, **kwargs)
    except Exception:
        import pdb
        import sys
        import traceback

        traceback.print_exc()
        pdb.post_mortem(sys.exc_info()[2])
        raise


__all__ = [n for n in dir() if not n.startswith("_") and n not in ("builtins", "numpy", "torch", "asarray", "to_np", "scaled_distance",
                                                                    "derivative_finite_diff")]
