"""Householder QR of a tall n x q matrix on the library's own kernels -- what stands behind ``gnp.qr``
(gpmp/num/numpy_backend.py: ``qr`` = SciPy / LAPACK ``geqrf`` + ``orgqr``), the contrast matrices of gpmp/core/linalg.py:49-110
and the contrast-space predictor of gpmp/core/kriging.py:202-257.

P = Q [R; 0] with Q = H_0 ... H_{q-1}, H_k = I - tau_k v_k v_k^T (LAPACK's sign convention: R_kk = -sign(x_k) |x|, so Q and R
agree with ``geqrf`` up to rounding).  Q (n x n) is only formed when a caller asks for it: its action on an n x m matrix is q
rank-one updates on the library GEMM, and Q^T K Q is q symmetric rank-two updates."""
import numpy
import torch


def _gnp():
    from .. import num

    return num


class HouseholderQR:
    def __init__(self, P, check_rank=True):
        """``check_rank``: the kriging callers (contrast matrices, contrast-space predictor) need a full-rank mean design and get
        a LinAlgError otherwise; the generic ``gnp.qr`` proceeds like LAPACK's geqrf -- a column that is already zero below the
        diagonal gets tau = 0 (H_k = I) and a zero R_kk."""
        gnp = _gnp()
        A = gnp.asarray(P).clone()
        if A.dim() != 2:
            raise ValueError("expected a 2-D array")
        n, q = A.shape
        if q >= n:
            raise numpy.linalg.LinAlgError("mean design has at least as many columns as observations")
        self.n, self.q, self.v, self.tau = n, q, [], []
        scale = [float(torch.sqrt(torch.sum(A[:, k] * A[:, k]))) for k in range(q)]
        for k in range(q):
            x = A[k:, k]
            nx = float(torch.sqrt(torch.sum(x * x)))
            if not nx > n * gnp.eps * scale[k]:
                if check_rank:
                    raise numpy.linalg.LinAlgError("singular mean design: P is rank deficient")
                if nx == 0.0:                                  # geqrf: nothing to annihilate, H_k = I
                    v = gnp.zeros((n,))
                    v[k] = 1.0
                    self.v.append(v)
                    self.tau.append(0.0)
                    continue
            alpha = -nx if float(x[0]) >= 0.0 else nx
            v = gnp.zeros((n,))
            v[k:] = x
            v[k] -= alpha
            tau = 2.0 / float(torch.sum(v * v))
            A[:, k:] -= tau * v.reshape(-1, 1) * torch.sum(v.reshape(-1, 1) * A[:, k:], dim=0).reshape(1, -1)   # O(n q)
            self.v.append(v)
            self.tau.append(tau)
        self.R = torch.triu(A[:q, :q])

    def _rank_update(self, B, cols_a, cols_b):
        """B -= [a_1 .. a_r] [b_1 .. b_r]^T on the library GEMM (B n x m in place; a_j: n, b_j: m)."""
        gnp = _gnp()
        lib = gnp._lib.load()
        A_ = gnp.as_matrix(torch.stack(cols_a, dim=1), copy=True)
        B_ = gnp.as_matrix(torch.stack(cols_b, dim=1), copy=True)
        gnp._lib.check(lib.gpmp_dgemm(0, 1, B.shape[0], B.shape[1], len(cols_a), -1.0, gnp._ptr(A_), gnp._ld(A_), gnp._ptr(B_),
                                      gnp._ld(B_), 1.0, gnp._ptr(B), gnp._ld(B), 0, gnp._stream()), "gpmp_dgemm")

    def apply(self, B, transpose):
        """B <- Q^T B (transpose) or Q B, in place; B is an n x m matrix."""
        gnp = _gnp()
        order = range(self.q) if transpose else range(self.q - 1, -1, -1)
        for k in order:
            c = gnp.coldots(B, self.v[k].reshape(-1, 1))[0]          # v^T B
            self._rank_update(B, [self.tau[k] * self.v[k]], [c])
        return B

    def congruence(self, K):
        """K <- Q^T K Q in place (K symmetric, full storage)."""
        gnp = _gnp()
        for k in range(self.q):
            v, tau = self.v[k], self.tau[k]
            w = gnp.matmul(K, v)
            s = float(torch.sum(v * w))
            u = tau * w - (0.5 * tau * tau * s) * v
            self._rank_update(K, [v, u], [u, v])
        return K

    def columns(self, j0, j1):
        """Q[:, j0:j1] as an n x (j1 - j0) matrix: Q applied to those columns of the identity."""
        gnp = _gnp()
        E = gnp.alloc_matrix(self.n, j1 - j0)
        E.zero_()
        if j1 > j0:
            E.diagonal(offset=-j0).fill_(1.0)                        # E[j0 + c, c] = 1
        return self.apply(E, transpose=False)


def qr(A, mode="reduced"):
    """(Q, R) of a tall matrix: mode "reduced" -> n x q, q x q; "complete" -> n x n, n x q; "r" -> R only."""
    gnp = _gnp()
    A = gnp.asarray(A)
    if A.dim() != 2:
        raise ValueError("qr: expected a 2-D array")
    n, q = A.shape
    if q >= n or q == 0:
        # square / wide / empty inputs do not occur on the GP path (a mean design has q < n columns): host LAPACK
        Qh, Rh = numpy.linalg.qr(gnp.to_np(A), mode="complete" if mode == "complete" else "reduced")
        return gnp.asarray(Rh) if mode == "r" else (gnp.asarray(Qh), gnp.asarray(Rh))
    h = HouseholderQR(A, check_rank=False)
    if mode == "r":
        return h.R
    if mode == "complete":
        R = gnp.zeros((n, q))
        R[:q] = h.R
        return h.columns(0, n), R
    if mode != "reduced":
        raise ValueError("qr: mode must be 'reduced', 'complete' or 'r'")
    return h.columns(0, q), h.R
