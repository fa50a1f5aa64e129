"""Local arithmetic of the block-cyclic layer on one rank: every method is a thin call into the C ABI (libgpmp_hip.so) on the
rank's device tensors -- the Gram block of the local point subsets, the diagonal factor, panel solves, the trailing update, the
fused ``gpmp_dist_*`` steps.  ``BlockCyclicCholesky`` (cholesky.py) holds the schedule and the communication; tests swap this class
for a torch stand-in to run the same schedule without a GPU (tests/test_dist_cpu.py)."""
from __future__ import annotations

import torch


class HipLocalOps:
    """Local arithmetic through libgpmp_hip.so (the product path)."""

    name = "hip"

    def __init__(self):
        from .. import _lib
        from .. import num as gnp

        self.lib, self._lib, self.gnp = _lib.load(), _lib, gnp
        self.device = gnp._dev()

    def empty(self, rows, cols):
        return self.gnp.alloc_matrix(rows, cols)

    def gram_block(self, cov, x_rows, x_cols, covparam):
        """K(x_rows, x_cols) without the diagonal term (cross-covariance path of the kernel)."""
        return self.gnp.as_matrix(self.gnp.asarray(cov(x_rows, x_cols, covparam)))

    def potrf(self, A):
        """In-place lower Cholesky of the square view A -> (dinv, info tensor)."""
        g = self.gnp
        n = A.shape[0]
        dinv = torch.empty(max(int(self.lib.gpmp_dinv_elems(n)), 1), dtype=torch.float64, device=self.device)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._lib.check(self.lib.gpmp_potrf_lower_async(g._ptr(A), n, g._ld(A), g._ptr(dinv), g._ptr(info), g._stream()),
                        "gpmp_potrf_lower_async")
        return dinv, info

    def diag_inverses(self, L):
        g = self.gnp
        n = L.shape[0]
        dinv = torch.empty(max(int(self.lib.gpmp_dinv_elems(n)), 1), dtype=torch.float64, device=self.device)
        self._lib.check(self.lib.gpmp_trtri_diag_blocks(g._ptr(L), n, g._ld(L), g._ptr(dinv), g._stream()), "gpmp_trtri_diag_blocks")
        return dinv

    def trsm_right(self, L, dinv, B):
        """B <- B L^-T in place (B: M x k view)."""
        g = self.gnp
        if B.shape[0] == 0:
            return
        self._lib.check(self.lib.gpmp_trsm_right_lower(g._ptr(L), L.shape[0], g._ld(L), g._ptr(dinv), g._ptr(B), B.shape[0],
                                                       g._ld(B), g._stream()), "gpmp_trsm_right_lower")

    def trsm_right_via_inverse(self, L, dinv, B, out):
        """out <- B L^-T as ONE product with T = L^-1 (doubling from the 128-block inverses): out = B T^T, the k loop of each
        tile column stopping at the diagonal.  Under a machine-filling GEMM on another stream every launch of the panel chain
        waits for a workgroup slot, so 7 small launches + 1 large beat the 15 of the substitution."""
        g = self.gnp
        k = L.shape[0]
        T = g.alloc_matrix(k, k)
        self._lib.check(self.lib.gpmp_trtri_lower(g._ptr(L), k, g._ld(L), g._ptr(dinv), g._ptr(T), g._ld(T), g._stream()), "gpmp_trtri_lower")
        self._lib.check(self.lib.gpmp_dgemm(0, 1, B.shape[0], k, k, 1.0, g._ptr(B), g._ld(B), g._ptr(T), g._ld(T), 0.0, g._ptr(out),
                                            g._ld(out), 4, g._stream()), "gpmp_dgemm")

    def gemm_nt_sub(self, C, A, B):
        """C -= A B^T  (C: M x N view, A: M x K, B: N x K)."""
        g = self.gnp
        M, N, K = C.shape[0], C.shape[1], A.shape[1]
        if M == 0 or N == 0:
            return
        self._lib.check(self.lib.gpmp_dgemm(0, 1, M, N, K, -1.0, g._ptr(A), g._ld(A), g._ptr(B), g._ld(B), 1.0, g._ptr(C),
                                            g._ld(C), 0, g._stream()), "gpmp_dgemm")

    def solve_lower_vec(self, L, dinv, v):
        """L^-1 v for a small diagonal block (vector)."""
        g = self.gnp
        x = g.as_matrix(v.reshape(-1, 1), copy=True)
        self._lib.check(self.lib.gpmp_trsm_lower(g._ptr(L), L.shape[0], g._ld(L), g._ptr(dinv), g._ptr(x), 1, g._ld(x), 0, None,
                                                 g._stream()), "gpmp_trsm_lower")
        return x.reshape(-1)

    def matvec(self, A, v):
        """A v through the library GEMM (A: M x K view)."""
        g = self.gnp
        M, K = A.shape
        out = g.alloc_matrix(M, 1)
        if M == 0:
            return out.reshape(-1)
        vm = g.as_matrix(v.reshape(-1, 1), copy=True)
        self._lib.check(self.lib.gpmp_dgemm(0, 0, M, 1, K, 1.0, g._ptr(A), g._ld(A), g._ptr(vm), g._ld(vm), 0.0, g._ptr(out),
                                            g._ld(out), 0, g._stream()), "gpmp_dgemm")
        return out.reshape(-1)

    def sum_log_diag(self, L):
        return float(torch.log(torch.diagonal(L)).sum().item())

    def asarray(self, a):
        return self.gnp.asarray(a)

    def trsm_left(self, L, dinv, B):
        """B <- L^-1 B in place (B: k x m view, L: k x k lower)."""
        g = self.gnp
        if B.shape[1] == 0:
            return
        self._lib.check(self.lib.gpmp_trsm_lower(g._ptr(L), L.shape[0], g._ld(L), g._ptr(dinv), g._ptr(B), B.shape[1], g._ld(B), 0, None,
                                                 g._stream()), "gpmp_trsm_lower")

    def trsm_left_t(self, L, dinv, B):
        """B <- L^-T B in place (B: k x m view, L: k x k lower)."""
        g = self.gnp
        if B.shape[1] == 0:
            return
        self._lib.check(self.lib.gpmp_trsm_lower(g._ptr(L), L.shape[0], g._ld(L), g._ptr(dinv), g._ptr(B), B.shape[1], g._ld(B), 1, None,
                                                 g._stream()), "gpmp_trsm_lower")

    def gemm_nn_sub(self, C, A, B):
        """C -= A B  (C: M x N view, A: M x K, B: K x N)."""
        g = self.gnp
        M, N, K = C.shape[0], C.shape[1], A.shape[1]
        if M == 0 or N == 0:
            return
        self._lib.check(self.lib.gpmp_dgemm(0, 0, M, N, K, -1.0, g._ptr(A), g._ld(A), g._ptr(B), g._ld(B), 1.0, g._ptr(C), g._ld(C), 0,
                                            g._stream()), "gpmp_dgemm")

    def coldots(self, V, w):
        """(V^T w, column sums of squares of V) for the local rows: two (m,) device vectors."""
        out = self.gnp.coldots(V, w.reshape(-1, 1))
        return out[0], out[1]

    def coldots_many(self, V, W):
        """(W^T V as an (r, m) array, column sums of squares of V) in one pass over V (W: rows x r, r <= 72)."""
        out = self.gnp.coldots(V, self.gnp.as_matrix(W))
        return out[:-1], out[-1]

    def matmul(self, A, B):
        """A B through the library GEMM (A: M x K view, B: K x r)."""
        g = self.gnp
        M, K = A.shape
        out = g.alloc_matrix(M, B.shape[1])
        if M == 0:
            return out
        Bm = g.as_matrix(B, copy=True)
        self._lib.check(self.lib.gpmp_dgemm(0, 0, M, B.shape[1], K, 1.0, g._ptr(A), g._ld(A), g._ptr(Bm), g._ld(Bm), 0.0, g._ptr(out),
                                            g._ld(out), 0, g._stream()), "gpmp_dgemm")
        return out

    def pairwise_variance(self, cov, xt, covparam):
        return self.gnp.asarray(cov(xt, None, covparam, pairwise=True)).reshape(-1)

    def gemm_tn(self, A, B):
        """A^T B through the library GEMM (A: K x M, B: K x N views) -> M x N."""
        g = self.gnp
        K, M = A.shape
        N = B.shape[1]
        out = g.alloc_matrix(M, N, zero=(K == 0))
        if K == 0 or M == 0 or N == 0:
            return out
        self._lib.check(self.lib.gpmp_dgemm(1, 0, M, N, K, 1.0, g._ptr(A), g._ld(A), g._ptr(B), g._ld(B), 0.0, g._ptr(out), g._ld(out), 0,
                                            g._stream()), "gpmp_dgemm")
        return out

    def gemm_tn_into(self, A, B, C):
        """C <- A^T B through the library GEMM, C a (strided) view of the right shape: no temporary, no copy."""
        g = self.gnp
        K, M = A.shape
        N = B.shape[1]
        if M == 0 or N == 0:
            return
        if K == 0:
            C.zero_()
            return
        self._lib.check(self.lib.gpmp_dgemm(1, 0, M, N, K, 1.0, g._ptr(A), g._ld(A), g._ptr(B), g._ld(B), 0.0, g._ptr(C), g._ld(C), 0,
                                            g._stream()), "gpmp_dgemm")

    def gemm_tn_acc(self, A, B, C):
        """C += A^T B through the library GEMM (C a strided view of the right shape)."""
        g = self.gnp
        K, M = A.shape
        N = B.shape[1]
        if M == 0 or N == 0 or K == 0:
            return
        self._lib.check(self.lib.gpmp_dgemm(1, 0, M, N, K, 1.0, g._ptr(A), g._ld(A), g._ptr(B), g._ld(B), 1.0, g._ptr(C), g._ld(C), 0,
                                            g._stream()), "gpmp_dgemm")

    def grad_trace_cross(self, M, xr, xc, p, covparam, noise, F, G):
        """[sum M sigma^2 Kc, sum M dK/dlog(1/rho_j) ...] over the rectangular block M (rows: points xr, columns: points xc),
        M <- M - F G^T in registers: gpmp_matern_grad_trace_cross.  Returns a (1 + d,) device vector."""
        g = self.gnp
        n, m = M.shape
        d = xr.shape[1]
        out = torch.zeros(1 + d, dtype=torch.float64, device=self.device)
        if n == 0 or m == 0:
            return out
        r = 0 if F is None else F.shape[1]
        Fm = Gm = None
        if r:
            # (same leading dimension for both: the kernel takes one ldf)
            Fm, Gm = g.alloc_matrix(n, r), g.alloc_matrix(m, r)
            Fm.copy_(g.asarray(F))
            Gm.copy_(g.asarray(G))
        xr, xc = g.asarray(xr).contiguous(), g.asarray(xc).contiguous()
        ws = torch.empty(int(self.lib.gpmp_grad_ws_elems(n, d)), dtype=torch.float64, device=self.device)
        self._lib.check(self.lib.gpmp_matern_grad_trace_cross(g._ptr(M), g._ld(M), g._ptr(xr), n, g._ptr(xc), m, d, int(p),
                                                              self._lib.host_vec(covparam), 1 if noise else 0, g._ptr(Fm), g._ptr(Gm), r,
                                                              g._ld(Fm) if r else 1, g._ptr(out), g._ptr(ws), g._stream()),
                        "gpmp_matern_grad_trace_cross")
        return out

    # ---- one block-column step through gpmp_dist_* (include/gpmp_hip.h): what a C++ / RCCL host calls between its
    # collectives (examples/dist_potrf_rccl.cpp).  ``lay`` = (n, nb, Pr, Pc, r, c).  The schedule uses these when the
    # local-ops object has them; the generic tensor-level code they replace stays for the CPU stand-in of the tests.
    step_abi = True

    def diag_factor_msg(self, D, msg):
        g = self.gnp
        self._lib.check(self.lib.gpmp_dist_diag_factor(g._ptr(D), D.shape[0], g._ld(D), g._ptr(msg), g._stream()), "gpmp_dist_diag_factor")

    def panel_solve_msg(self, Lkk, P, panel):
        """panel <- P L_kk^-T (and P in place); ``Lkk`` is the view at the start of the diagonal-block message"""
        g = self.gnp
        bk = Lkk.shape[0]
        ws = torch.empty(int(self.lib.gpmp_dist_panel_ws_elems(bk)), dtype=torch.float64, device=self.device) if bk % 128 == 0 else None
        self._lib.check(self.lib.gpmp_dist_panel_solve(g._ptr(Lkk), bk, g._ptr(P), P.shape[0], g._ld(P), g._ptr(panel), g._ld(panel),
                                                       g._ptr(ws), g._stream()), "gpmp_dist_panel_solve")

    def exchange_rows(self, lay, rp, k):
        n, nb, pr, pc, r, c = lay
        return int(self.lib.gpmp_dist_exchange_rows(n, nb, pr, pc, rp, c, k))

    def exchange_pack(self, panel, piece, lay, k, bk):
        g = self.gnp
        n, nb, pr, pc, r, c = lay
        self._lib.check(self.lib.gpmp_dist_exchange_pack(g._ptr(panel), g._ld(panel), g._ptr(piece), g._ld(piece), n, nb, pr, pc, r, c, k, bk,
                                                         g._stream()), "gpmp_dist_exchange_pack")

    def exchange_unpack(self, piece, colop, lay, rp, k, bk):
        g = self.gnp
        n, nb, pr, pc, r, c = lay
        self._lib.check(self.lib.gpmp_dist_exchange_unpack(g._ptr(piece), g._ld(piece), g._ptr(colop), g._ld(colop), n, nb, pr, pc, rp, c, k,
                                                           bk, g._stream()), "gpmp_dist_exchange_unpack")

    def inverse_gram(self, T, T2, M, lay, c2, lower_only):
        """M <- T^T T2 for the column sets (c, c2) of the block-cyclic inverse factor, every block with its exact contraction range,
        ONE launch (gpmp_dist_inverse_gram); lower_only: the blocks J <= I only"""
        g = self.gnp
        n, nb, pr, pc, r, c = lay
        self._lib.check(self.lib.gpmp_dist_inverse_gram(g._ptr(T), g._ld(T), g._ptr(T2), g._ld(T2), g._ptr(M), g._ld(M), n, nb, pr, pc, r, c,
                                                        int(c2), 1 if lower_only else 0, g._stream()), "gpmp_dist_inverse_gram")

    def trailing_update(self, A, lay, k, panel, colop, jlo, jhi, rows_after):
        g = self.gnp
        n, nb, pr, pc, r, c = lay
        self._lib.check(self.lib.gpmp_dist_trailing_update(g._ptr(A), g._ld(A), n, nb, pr, pc, r, c, k, g._ptr(panel), g._ld(panel),
                                                           g._ptr(colop), g._ld(colop), jlo, jhi, -1 if rows_after is None else rows_after,
                                                           g._stream()), "gpmp_dist_trailing_update")
