"""gpmp_amd.num -- the "hip" numerical backend: GPmp's backend contract on MI355X.

Counterpart of gpmp/num/numpy_backend.py (and of the dispatcher gpmp/num/__init__.py:25-46) for
the names gpmp.core / gpmp.kernel use on the hot path.  Arrays are ``torch.float64`` tensors in HBM
(PyTorch-ROCm is only the container: allocator, streams); every O(n^2) / O(n^3) operation is a
call into libgpmp_hip.so through ctypes -- there is no CPU or torch fallback for them.

Conventions kept from the reference: float64 only; ``cholesky`` returns the LOWER factor;
``cholesky_solve(A, b) -> (x, L)`` with b 1-D or 2-D (numpy_backend.py:465-469);
``scaled_distance`` takes LOG INVERSE length-scales (numpy_backend.py:432-436); a failed Cholesky
raises a ``numpy.linalg.LinAlgError`` whose text contains "not positive definite" / "cholesky", so
``_is_linalg_exception`` (numpy_backend.py:158-162) maps it to +inf in the selection criteria.
"""
from __future__ import annotations

import builtins
import ctypes
import math
from typing import Optional

import numpy
import torch

from .. import _lib
from ..config import get_config, get_device, get_logger

_gpmp_backend_ = "hip"
_config = get_config()
_config.dtype_resolved = torch.float64
_logger = get_logger()

float64 = torch.float64
pi = math.pi
inf = math.inf
nan = math.nan
eps = float(numpy.finfo(numpy.float64).eps)
fmax = float(numpy.finfo(numpy.float64).max)
finfo = torch.finfo

_LINALG_ERROR_KEYWORDS = (  # gpmp/num/numpy_backend.py:30-46
    "singular", "not positive definite", "not positive-definite", "cholesky", "decomposition",
    "factorization", "matrix is not invertible", "matrix inversion", "inverse", "svd did not converge",
    "ill-conditioned", "linalg", "lapack", "cusolver", "array must not contain infs or nans",
)

LD_ALIGN = 16  # leading dimensions are padded to 16 doubles (128 B): 16-byte vector loads everywhere


class HipLinAlgError(numpy.linalg.LinAlgError):
    """Non positive-definite pivot reported by gpmp_potrf_lower_async (LAPACK-style info > 0)."""


def safe_inf():
    return inf


def safe_neginf():
    return -inf


def _is_linalg_exception(exc: Exception) -> bool:
    """gpmp/num/numpy_backend.py:158-162."""
    if isinstance(exc, numpy.linalg.LinAlgError):
        return True
    msg = str(exc).lower()
    return builtins.any(k in msg for k in _LINALG_ERROR_KEYWORDS)


# ---------------------------------------------------------------------------------------------
# array plumbing (torch is the container)
# ---------------------------------------------------------------------------------------------
_device_bound = False


def _dev():
    """Device of this process; the first call makes it the current HIP device (the library's helper stream,
    events and kernel attributes are created on the current device)."""
    global _device_bound
    d = get_device()
    if not _device_bound and d.type == "cuda":
        torch.cuda.set_device(d)
        _device_bound = True
    return d


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream(_dev()).cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def alloc_matrix(n: int, m: int, zero: bool = False) -> torch.Tensor:
    """n x m fp64 matrix in HBM whose row stride is padded to LD_ALIGN doubles (a strided view)."""
    ld = builtins.max(LD_ALIGN, (m + LD_ALIGN - 1) // LD_ALIGN * LD_ALIGN)
    buf = (torch.zeros if zero else torch.empty)((builtins.max(n, 1), ld), dtype=torch.float64, device=_dev())
    return buf[:n, :m]


def as_matrix(a: torch.Tensor, copy: bool = False) -> torch.Tensor:
    """2-D device tensor with unit column stride (row stride = leading dimension)."""
    if a.dim() != 2:
        raise ValueError("expected a 2-D array")
    if copy or a.stride(1) != 1 or a.stride(0) < a.shape[1] or a.dtype != torch.float64 or a.device != _dev():
        out = alloc_matrix(a.shape[0], a.shape[1])
        out.copy_(a)
        return out
    return a


def _ld(a: torch.Tensor) -> int:
    return int(a.stride(0)) if a.shape[0] > 1 else builtins.max(int(a.stride(0)), int(a.shape[1]), 1)


def asarray(x, dtype=None):
    """gpmp/num/numpy_backend.py:174-188 -- anything -> fp64 device tensor (ints stay ints)."""
    if isinstance(x, torch.Tensor):
        if x.is_floating_point():
            return x.to(device=_dev(), dtype=dtype or torch.float64)
        return x.to(device=_dev())
    if isinstance(x, (int, float)):
        return torch.tensor([x], dtype=torch.float64 if isinstance(x, float) else None, device=_dev())
    if isinstance(x, (list, tuple)) and builtins.any(isinstance(v, torch.Tensor) for v in x):
        # e.g. gnp.array([gnp.log(gnp.var(z))]): a sequence holding device scalars (torch backend: torch_backend.py asarray)
        return torch.stack([asarray(v).reshape(()) if isinstance(v, torch.Tensor) and v.numel() == 1 and v.dim() <= 1
                            else asarray(v) for v in x]).to(dtype=dtype or torch.float64)
    arr = numpy.asarray(x)
    if numpy.issubdtype(arr.dtype, numpy.floating):
        return torch.as_tensor(numpy.ascontiguousarray(arr, dtype=numpy.float64), device=_dev())
    return torch.as_tensor(numpy.ascontiguousarray(arr), device=_dev())


array = asarray


def asdouble(x):
    return asarray(x).to(torch.float64)


def to_np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return numpy.asarray(x)


def small_spd_inverse(S, what="matrix"):
    """Inverse of a q x q symmetric positive definite matrix of the mean-space algebra (q = number of mean columns): the
    q^2 numbers are taken to the host, checked for numerical rank (lambda_min > q eps lambda_max, the test the reference's
    np.linalg.solve would fail) and inverted there -- as the reference does in NumPy (gpmp/core/kriging.py:141-159,
    loo.py:118-124); a device-side eigensolver / LU for a 9 x 9 matrix is a dozen library kernels on the critical path."""
    Sh = to_np(S).astype(numpy.float64)
    Sh = 0.5 * (Sh + Sh.T)
    ev = numpy.linalg.eigvalsh(Sh)
    if not ev[0] > Sh.shape[0] * eps * ev[-1]:
        raise numpy.linalg.LinAlgError(f"{what} is singular to working precision (rank-deficient mean design)")
    return asarray(numpy.linalg.inv(Sh))


def to_scalar(x):
    if isinstance(x, (int, float, bool)):
        return x
    return x.item()


def isarray(x):
    return isinstance(x, torch.Tensor)


def zeros(shape, dtype=None):
    return torch.zeros(shape, dtype=dtype or torch.float64, device=_dev())


def ones(shape, dtype=None):
    return torch.ones(shape, dtype=dtype or torch.float64, device=_dev())


def empty(shape, dtype=None):
    return torch.empty(shape, dtype=dtype or torch.float64, device=_dev())


def full(shape, fill_value, dtype=None):
    return torch.full(shape, fill_value, dtype=dtype or torch.float64, device=_dev())


def eye(n, m=None, k=0, dtype=None):
    out = torch.eye(n, n if m is None else m, dtype=dtype or torch.float64, device=_dev())
    return out if k == 0 else torch.roll(out, k, 1)


def linspace(start, stop, num=50, **_):
    return torch.linspace(start, stop, num, dtype=torch.float64, device=_dev())


def arange(*args):
    return torch.arange(*args, device=_dev())


# cheap O(n) / O(m) vector helpers used by mean functions and post-processing
def _elementwise(torch_fn, numpy_fn):
    """Device tensors go to torch; host values (the covariance / mean PARAMETERS, which reach user kernels as NumPy
    vectors from SciPy) are evaluated on the host as with the reference's NumPy backend.  Host scalars come back as
    Python floats so that ``gnp.exp(param[0]) * K`` multiplies a device matrix from the left without NumPy trying to
    convert it."""
    def f(x):
        if isinstance(x, torch.Tensor):
            return torch_fn(x)
        y = numpy_fn(numpy.asarray(x, dtype=numpy.float64))
        return float(y) if y.ndim == 0 else y
    f.__name__ = numpy_fn.__name__
    return f


exp, log, sqrt, abs = (_elementwise(torch.exp, numpy.exp), _elementwise(torch.log, numpy.log),  # noqa: A001
                       _elementwise(torch.sqrt, numpy.sqrt), _elementwise(torch.abs, numpy.abs))
sin, cos, tanh = _elementwise(torch.sin, numpy.sin), _elementwise(torch.cos, numpy.cos), _elementwise(torch.tanh, numpy.tanh)
maximum = lambda a, b: torch.clamp(a, min=b) if not isinstance(b, torch.Tensor) else torch.maximum(a, b)  # noqa: E731
minimum = lambda a, b: torch.clamp(a, max=b) if not isinstance(b, torch.Tensor) else torch.minimum(a, b)  # noqa: E731
where, isnan, isinf, isfinite = torch.where, torch.isnan, torch.isinf, torch.isfinite
def _joiner(torch_fn, numpy_fn):
    """Sequences of device tensors are joined on the device; sequences made only of host values (parameter vectors, which
    this backend keeps on the host as the NumPy backend does) are joined by NumPy; mixtures are moved to the device."""
    def f(seq, *args, **kwargs):
        seq = list(seq)
        if not builtins.any(isinstance(v, torch.Tensor) for v in seq):
            if "dim" in kwargs:
                kwargs["axis"] = kwargs.pop("dim")
            return numpy_fn([numpy.asarray(v) for v in seq], *args, **kwargs)
        if "axis" in kwargs:
            kwargs["dim"] = kwargs.pop("axis")
        return torch_fn([asarray(v) for v in seq], *args, **kwargs)
    f.__name__ = numpy_fn.__name__
    return f


hstack, vstack = _joiner(torch.hstack, numpy.hstack), _joiner(torch.vstack, numpy.vstack)
stack, concatenate = _joiner(torch.stack, numpy.stack), _joiner(torch.cat, numpy.concatenate)
diag, trace, copy = torch.diag, torch.trace, torch.clone
any, all = torch.any, torch.all  # noqa: A001


def sum(x, axis=None):  # noqa: A001
    return torch.sum(x) if axis is None else torch.sum(x, dim=axis)


def max(x, axis=None):  # noqa: A001
    return torch.max(x) if axis is None else torch.max(x, dim=axis).values


def min(x, axis=None):  # noqa: A001
    return torch.min(x) if axis is None else torch.min(x, dim=axis).values


def reshape(x, shape):
    return torch.reshape(x, shape)


def sort(x, axis=-1):
    return torch.sort(x, dim=axis).values


def diff(x, n=1, axis=-1):
    return torch.diff(x, n=n, dim=axis)


def var(x, axis=None):
    return torch.var(x, unbiased=False) if axis is None else torch.var(x, dim=axis, unbiased=False)


def mean(x, axis=None):
    return torch.mean(x) if axis is None else torch.mean(x, dim=axis)


def gammaln(x):
    return torch.lgamma(asarray(x).to(torch.float64))


def inftobigf(a, bigf=fmax / 1000.0):
    """gpmp/num/numpy_backend.py:250-252."""
    return torch.where(torch.isinf(a), torch.full_like(a, bigf), a)


def compute_gammaln(up_to_p: int):
    """gpmp/num/shared.py:21-41 (host table; the device kernels rebuild the coefficients themselves)."""
    cache = _config.caches.setdefault("gammaln", {})
    n = 2 * up_to_p + 2
    table = cache.get("table")
    if table is None or table.shape[0] < n:
        with numpy.errstate(divide="ignore"):
            table = numpy.array([math.lgamma(k) if k > 0 else math.inf for k in range(n)])
        cache["table"] = table
    return table[:n]


def derivative_finite_diff(f, x, h):
    """gpmp/num/shared.py:44-55 -- 5-point central difference."""
    return (-f(x + 2 * h) + 8 * f(x + h) - 8 * f(x - h) + f(x - 2 * h)) / (12.0 * h)


# ---------------------------------------------------------------------------------------------
# distances and Matern evaluation (HIP)
# ---------------------------------------------------------------------------------------------
def _host_params(p):
    if isinstance(p, torch.Tensor):
        p = p.detach().cpu().numpy()
    return numpy.ascontiguousarray(numpy.asarray(p, dtype=numpy.float64).reshape(-1))


def _points(x) -> torch.Tensor:
    x = asarray(x)
    if x.dim() != 2:
        raise ValueError("points must be a 2-D array (n, d)")
    return x.contiguous()


def scaled_distance(loginvrho, x, y):
    """gpmp/num/numpy_backend.py:432-436 on the GPU: direct sum of squared scaled differences."""
    lib = _lib.load()
    x, y = _points(x), _points(y)
    n, d = x.shape
    m = y.shape[0]
    lir = _host_params(loginvrho)
    if lir.shape[0] != d or y.shape[1] != d:
        raise ValueError("dimension mismatch between loginvrho, x and y")
    D = alloc_matrix(n, m)
    hv = _lib.host_vec(lir)
    _lib.check(lib.gpmp_scaled_distance(_ptr(x), _ptr(y), n, m, d, hv, _ptr(D), _ld(D), _stream()), "gpmp_scaled_distance")
    return D


def scaled_distance_elementwise(loginvrho, x, y):
    """gpmp/num/numpy_backend.py:438-446."""
    x = _points(x)
    if x is y or y is None:
        return zeros((x.shape[0],))
    y = _points(y)
    invrho = torch.exp(asarray(_host_params(loginvrho)))
    return torch.sqrt(torch.sum((invrho * (x - y)) ** 2, dim=1))


# ---------------------------------------------------------------------------------------------
# Cholesky factor object and triangular algebra (HIP)
# ---------------------------------------------------------------------------------------------
class CholFactor:
    """Lower Cholesky factor in HBM + the inverses of its 128 x 128 diagonal blocks."""

    __slots__ = ("L", "dinv", "n", "_logdet")

    def __init__(self, L, dinv):
        self.L, self.dinv, self.n = L, dinv, L.shape[0]
        self._logdet = None

    def logdet(self):
        """2 sum log L_ii (likelihood.py:50) -> python float."""
        if self._logdet is None:
            lib = _lib.load()
            out = torch.empty(1, dtype=torch.float64, device=_dev())
            _lib.check(lib.gpmp_logdet_chol(_ptr(self.L), self.n, _ld(self.L), _ptr(out), _stream()), "gpmp_logdet_chol")
            self._logdet = float(out.item())
        return self._logdet

    def solve_lower(self, B, trans=False, overwrite=False):
        """op(L)^-1 B; B is (n,) or (n, m)."""
        lib = _lib.load()
        vec = B.dim() == 1
        Bm = B.reshape(-1, 1) if vec else B
        if Bm.shape[0] != self.n:
            raise ValueError("right-hand side has the wrong number of rows")
        X = as_matrix(Bm, copy=not overwrite)
        # the factor's workspace doubles as scratch: its tail (n > 1024) feeds the fused forward-solve leaves
        scratch = _ptr(self.dinv) if self.dinv.numel() >= int(lib.gpmp_dinv_elems(self.n)) else None
        _lib.check(
            lib.gpmp_trsm_lower(_ptr(self.L), self.n, _ld(self.L), _ptr(self.dinv), _ptr(X), X.shape[1], _ld(X),
                                1 if trans else 0, scratch, _stream()),
            "gpmp_trsm_lower",
        )
        return X.reshape(-1) if vec else X

    def solve(self, B):
        """K^-1 B = L^-T L^-1 B."""
        Y = self.solve_lower(B, trans=False)
        return self.solve_lower(Y, trans=True, overwrite=True)

    def inverse_factor(self):
        """T = L^-1 (lower triangular, strict upper part zero)."""
        lib = _lib.load()
        T = alloc_matrix(self.n, self.n)
        _lib.check(lib.gpmp_trtri_lower(_ptr(self.L), self.n, _ld(self.L), _ptr(self.dinv), _ptr(T), _ld(T), _stream()),
                   "gpmp_trtri_lower")
        return T

    def inverse_lower(self, T=None):
        """Lower triangle of K^-1 = T^T T (tiles on/below the diagonal are written)."""
        lib = _lib.load()
        if T is None:
            T = self.inverse_factor()
        Kinv = alloc_matrix(self.n, self.n)
        _lib.check(lib.gpmp_lauum_lower(_ptr(T), self.n, _ld(T), _ptr(Kinv), _ld(Kinv), _stream()), "gpmp_lauum_lower")
        return Kinv


def cholesky_factor(A, overwrite=False, check=True) -> CholFactor:
    """Factor a symmetric positive definite matrix (only its lower triangle is read)."""
    lib = _lib.load()
    A = asarray(A)
    if A.dim() != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("cholesky needs a square 2-D array")
    n = A.shape[0]
    L = as_matrix(A, copy=not overwrite)
    dinv = torch.empty(builtins.max(int(lib.gpmp_dinv_elems(n)), 1), dtype=torch.float64, device=_dev())
    info = torch.zeros(1, dtype=torch.int32, device=_dev())
    _lib.check(lib.gpmp_potrf_lower_async(_ptr(L), n, _ld(L), _ptr(dinv), _ptr(info), _stream()), "gpmp_potrf_lower_async")
    if check:
        k = int(info.item())
        if k != 0:
            raise HipLinAlgError(
                f"Matrix is not positive definite: Cholesky factorization failed at leading minor {k} (potrf info={k})"
            )
    return CholFactor(L, dinv)


def cholesky_factor_solve(A, B, overwrite=True, check=True):
    """Factor A and solve L X = B in one library call (gpmp_potrf_trsm_lower_async: the solve of the leading rows
    overlaps the trailing part of the factorisation).  Returns (CholFactor, X); A and B are overwritten by default."""
    lib = _lib.load()
    A, B = asarray(A), asarray(B)
    n = A.shape[0]
    if A.dim() != 2 or A.shape[1] != n or B.dim() != 2 or B.shape[0] != n:
        raise ValueError("cholesky_factor_solve needs a square A (n x n) and a 2-D B (n x m)")
    L = as_matrix(A, copy=not overwrite)
    X = as_matrix(B, copy=not overwrite)
    dinv = torch.empty(builtins.max(int(lib.gpmp_dinv_elems(n)), 1), dtype=torch.float64, device=_dev())
    info = torch.zeros(1, dtype=torch.int32, device=_dev())
    _lib.check(lib.gpmp_potrf_trsm_lower_async(_ptr(L), n, _ld(L), _ptr(dinv), _ptr(info), _ptr(X), X.shape[1], _ld(X), _stream()),
               "gpmp_potrf_trsm_lower_async")
    if check:
        k = int(info.item())
        if k != 0:
            raise HipLinAlgError(
                f"Matrix is not positive definite: Cholesky factorization failed at leading minor {k} (potrf info={k})"
            )
    return CholFactor(L, dinv), X


def cholesky(A):
    """numpy.linalg.cholesky (numpy_backend.py:136): lower factor, zeros above the diagonal."""
    lib = _lib.load()
    F = cholesky_factor(A)
    _lib.check(lib.gpmp_tril(_ptr(F.L), F.n, _ld(F.L), _stream()), "gpmp_tril")
    F.L._gpmp_dinv = F.dinv  # lets solve_triangular reuse the diagonal-block inverses
    return F.L


def cholesky_solve(A, b):
    """gpmp/num/numpy_backend.py:465-469 -> (A^-1 b, L)."""
    lib = _lib.load()
    F = cholesky_factor(A)
    b = asarray(b)
    x = F.solve(b)
    _lib.check(lib.gpmp_tril(_ptr(F.L), F.n, _ld(F.L), _stream()), "gpmp_tril")
    F.L._gpmp_dinv = F.dinv
    return x, F.L


def solve_triangular(A, B, trans=0, lower=False, unit_diagonal=False, overwrite_b=False, check_finite=False):
    """scipy.linalg.solve_triangular for the cases on the path (numpy_backend.py:140, linalg.py:41)."""
    lib = _lib.load()
    if unit_diagonal:
        raise NotImplementedError("unit_diagonal is not used on the GP path")
    A = asarray(A)
    t = trans in (1, 2, "T", "t", "C", "c")
    if not lower:
        # upper triangular U = L^T as produced by `L.T`: solve with the transposed lower factor
        A = A.T
        t = not t
    dinv = getattr(A, "_gpmp_dinv", None)
    L = as_matrix(A)
    if dinv is None or L is not A:
        dinv = torch.empty(builtins.max(int(lib.gpmp_dinv_elems(L.shape[0])), 1), dtype=torch.float64, device=_dev())
        _lib.check(lib.gpmp_trtri_diag_blocks(_ptr(L), L.shape[0], _ld(L), _ptr(dinv), _stream()), "gpmp_trtri_diag_blocks")
    return CholFactor(L, dinv).solve_lower(asarray(B), trans=t)


def cholesky_inv(A):
    """K^-1 through potrf + trtri + T^T T (numpy_backend.py:458-463 uses numpy.linalg.inv)."""
    lib = _lib.load()
    F = cholesky_factor(A)
    Kinv = F.inverse_lower()
    _lib.check(lib.gpmp_symmetrize_from_lower(_ptr(Kinv), F.n, _ld(Kinv), _stream()), "gpmp_symmetrize_from_lower")
    return Kinv


inv = cholesky_inv


def logdet(A):
    """numpy_backend.py:449-456 for symmetric positive definite A: 2 sum log L_ii (raises if not PD)."""
    try:
        return cholesky_factor(A).logdet()
    except HipLinAlgError as exc:
        raise ValueError("Matrix is not positive definite (or has non-positive determinant).") from exc


def solve(A, B, overwrite_a=True, overwrite_b=True, assume_a="gen", sym_pos=False):
    """scipy.linalg.solve as used on the path: SPD systems go through the HIP Cholesky.  General / symmetric-indefinite systems
    are the q x q (or (q + 1) x (q + 1)) blocks of the mean-space algebra: host LAPACK, like the other q x q steps.  A LARGE
    general system is not something the GP path produces (universal kriging goes through the Schur complement or the contrast
    space, core/kriging.py); user code written against the backend namespace gets the same host LAPACK call, never a vendor GPU
    solver."""
    A, B = asarray(A), asarray(B)
    if assume_a == "pos" or sym_pos:
        return cholesky_factor(A).solve(B)
    return asarray(numpy.linalg.solve(to_np(A), to_np(B)))


def qr(A, mode="reduced"):
    """Householder QR on the library's GEMM / column-dot kernels (gpmp_amd/num/householder.py)."""
    from .householder import qr as _qr

    return _qr(A, mode)


def svd(A, full_matrices=True, hermitian=True):
    """(U, s, Vt) with A = U diag(s) Vt, singular values in decreasing order -- scipy.linalg.svd / torch.linalg.svd of the
    reference backends, for the SQUARE matrices the path hands it (gpmp/core/sample_paths.py:54-58: a covariance matrix that is
    only positive semi-definite).  One-sided Jacobi on the library's own kernel (gpmp_jacobi_sweep: rows of G = A rotated until
    mutually orthogonal, the rotations accumulated in W = U^T).  ``hermitian`` is ignored, as in the reference backends: symmetry
    and definiteness are detected, never assumed.  Non-square input does not occur on the path: host LAPACK."""
    lib = _lib.load()
    A = asarray(A)
    if A.dim() != 2 or A.shape[0] != A.shape[1]:
        Uh, sh, Vh = numpy.linalg.svd(to_np(A), full_matrices=full_matrices)
        return asarray(Uh), asarray(sh), asarray(Vh)
    n = A.shape[0]
    if n == 0:
        return A.clone(), zeros((0,)), A.clone()
    G = as_matrix(A, copy=True)
    W = alloc_matrix(n, n)
    W.zero_()
    W.diagonal().fill_(1.0)
    fro = float(torch.sqrt(torch.sum(G[:, :n] * G[:, :n])))
    tiny = n * eps * fro                 # rows below this are rounding noise of a rank-deficient matrix: they would never settle
    tol = 8.0 * eps * math.sqrt(n)       # a length-n dot product is exact to about sqrt(n) eps relative
    conv = torch.zeros(1, dtype=torch.float64, device=_dev())
    for _ in range(40):
        _lib.check(lib.gpmp_jacobi_sweep(_ptr(G), _ld(G), _ptr(W), _ld(W), n, tiny, _ptr(conv), _stream()), "gpmp_jacobi_sweep")
        if float(conv.item()) <= tol:
            break
    s = torch.sqrt(torch.sum(G * G, dim=1))
    order = torch.argsort(s, descending=True)
    s, G, W = s[order], G[order], W[order]
    U = W.T.contiguous()
    ok = s > builtins.max(tiny, n * eps * float(s[0]))
    # ``hermitian`` is accepted and NOT trusted: the reference backends ignore it and compute a general SVD
    # (gpmp/num/torch_backend.py:833-834).  The shortcut v_i = u_i is taken only where it is verified: A symmetric to rounding
    # and every kept singular pair with u_i . v_i > 0 (a positive semi-definite matrix -- the covariance matrices of
    # gpmp/core/sample_paths.py:54-58); there the rows of W (a product of plane rotations) are orthonormal to rounding, whereas
    # g_i / s_i carries the relative noise eps s_0 / s_i of a small singular value.  Everything else gets Vt = G / s row by row --
    # exact by construction (G = diag(s) Vt), also for a symmetric INDEFINITE matrix with a +-lambda pair, whose singular
    # subspace is two-dimensional (v_i != +-u_i in general).
    amax = float(A.abs().max())
    symmetric = bool(float((A - A.T).abs().max()) <= 8.0 * eps * amax)
    dots = torch.sum(G * W, dim=1)
    if symmetric and bool(torch.all(torch.where(ok, dots > 0.5 * s, torch.ones_like(ok)))):
        Vt = W.clone()
    else:
        Vt = torch.where(ok.reshape(-1, 1), G / torch.where(ok, s, torch.ones_like(s)).reshape(-1, 1), W)
        nbad = int((~ok).sum())
        if nbad and not symmetric:
            # numerically zero singular values of a NON-symmetric matrix: the matching rows of W are left null vectors; the
            # right ones are the orthogonal complement of the kept rows of Vt (Householder reflectors, never on the GP path)
            r = n - nbad
            if r == 0:
                Vt = torch.eye(n, dtype=torch.float64, device=_dev())
            else:
                from .householder import HouseholderQR

                h = HouseholderQR(Vt[:r].T.contiguous())
                Vt = torch.cat((Vt[:r], h.columns(r, n).T), dim=0)
    s = torch.where(ok, s, torch.zeros_like(s))
    return U, s, Vt


def matmul(A, B, ta=False, tb=False):
    """Dense product op(A) op(B) on the library's fp64 MFMA GEMM (2-D x 2-D; 1-D operands become columns / rows).
    ``ta`` / ``tb``: use the stored matrix transposed (no copy) -- the P^T U, U^T D products of the mean-space algebra."""
    lib = _lib.load()
    A, B = asarray(A), asarray(B)
    va, vb = A.dim() == 1, B.dim() == 1
    Am = as_matrix((A.reshape(-1, 1) if ta else A.reshape(1, -1)) if va else A)
    Bm = as_matrix((B.reshape(1, -1) if tb else B.reshape(-1, 1)) if vb else B)
    M, K = (Am.shape[1], Am.shape[0]) if ta else Am.shape
    K2, N = (Bm.shape[1], Bm.shape[0]) if tb else Bm.shape
    if K != K2:
        raise ValueError("matmul: inner dimensions differ")
    C = alloc_matrix(M, N)
    if M == 0 or N == 0:
        return C
    if K == 0:
        return C.zero_()
    _lib.check(lib.gpmp_dgemm(1 if ta else 0, 1 if tb else 0, M, N, K, 1.0, _ptr(Am), _ld(Am), _ptr(Bm), _ld(Bm), 0.0, _ptr(C),
                              _ld(C), 0, _stream()), "gpmp_dgemm")
    if va and vb:
        return C.reshape(())
    if va or vb:
        return C.reshape(-1)
    return C


def coldots(V, Y=None):
    """rows k < r: sum_i V[i,:] * Y[i,k]; last row: sum_i V[i,:]^2  (the einsum("i..., i...") reductions)."""
    lib = _lib.load()
    V = as_matrix(V)
    n, m = V.shape
    r = 0 if Y is None else Y.shape[1]
    Ym = None if Y is None else as_matrix(Y)
    out = torch.empty((r + 1, builtins.max(m, 1)), dtype=torch.float64, device=_dev())
    ws = torch.empty(builtins.max(int(lib.gpmp_coldots_ws_rows(n)) * m, 1), dtype=torch.float64, device=_dev())
    _lib.check(
        lib.gpmp_coldots(_ptr(V), n, m, _ld(V), _ptr(Ym), r, 1 if Ym is None else _ld(Ym), _ptr(out), out.stride(0),
                         _ptr(ws), _stream()),
        "gpmp_coldots",
    )
    return out[:, :m]


def coldots_pair(A, B):
    """out[j] = sum_i A[i,j] B[i,j] -- einsum("i..., i...") of two matrices (kriging.py:194 with the weights kept)."""
    lib = _lib.load()
    A, B = as_matrix(A), as_matrix(B)
    n, m = A.shape
    out = torch.empty((builtins.max(m, 1),), dtype=torch.float64, device=_dev())
    ws = torch.empty(builtins.max(int(lib.gpmp_coldots_ws_rows(n)) * m, 1), dtype=torch.float64, device=_dev())
    _lib.check(lib.gpmp_coldots_pair(_ptr(A), _ld(A), _ptr(B), _ld(B), n, m, _ptr(out), _ptr(ws), _stream()), "gpmp_coldots_pair")
    return out[:m]


def einsum(spec, a, b):
    """The two contractions the core uses: "i..., i..." (column dots) and "...i, i..." (mat-vec)."""
    a, b = asarray(a), asarray(b)
    s = spec.replace(" ", "")
    if s == "i...,i...":
        if a.dim() == 1 and b.dim() == 1:
            return torch.dot(a, b)
        if a.dim() == 2 and b.dim() == 1:
            return coldots(a, b.reshape(-1, 1))[0]
        if a.dim() == 1 and b.dim() == 2:
            return coldots(b, a.reshape(-1, 1))[0]
        if a.shape != b.shape:
            raise ValueError("einsum('i..., i...'): shape mismatch")
        if a.dim() == 2:
            return coldots_pair(a, b)
        return torch.sum(a * b, dim=0)
    if s == "...i,i...":
        return matmul(a, b)
    return torch.einsum(spec, a, b)


# RNG (gpmp/num/numpy_backend.py:473-495)
_gen = None


def set_seed(seed: int):
    global _gen
    _gen = torch.Generator(device=_dev())
    _gen.manual_seed(seed)


def _generator():
    if _gen is None:
        set_seed(_config.seed)
    return _gen


def rand(*shape):
    return torch.rand(shape, dtype=torch.float64, device=_dev(), generator=_generator())


def randn(*shape):
    return torch.randn(shape, dtype=torch.float64, device=_dev(), generator=_generator())


from .criterion import (BatchDifferentiableSelectionCriterion, DifferentiableSelectionCriterion,  # noqa: E402,F401
                        SecondOrderDifferentiableFunction)
from .extras import *  # noqa: E402,F401,F403  (the rest of the backend contract: thin wrappers, see extras.py)
from .extras import multivariate_normal, normal, scipy_mvnormal  # noqa: E402,F401
