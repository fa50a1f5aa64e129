"""CPU stand-in for gpmp_amd.dist's LocalOps (torch-CPU / LAPACK arithmetic) -- TEST INFRASTRUCTURE.

Lets the block-cyclic schedule (ownership maps, panel broadcasts, staircase updates) run under gloo
on a machine without GPUs.  The product path uses gpmp_amd.dist.HipLocalOps."""
import numpy as np
import torch


class CpuLocalOps:
    name = "cpu-test"

    def empty(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float64)

    def asarray(self, a):
        return torch.as_tensor(np.asarray(a, dtype=np.float64)) if not isinstance(a, torch.Tensor) else a.to(torch.float64)

    def gram_block(self, cov, x_rows, x_cols, covparam):
        return torch.as_tensor(np.ascontiguousarray(cov(x_rows.numpy(), x_cols.numpy(), covparam)))

    def potrf(self, A):
        info = torch.zeros(1, dtype=torch.int32)
        L, inf = torch.linalg.cholesky_ex(torch.tril(A) + torch.tril(A, -1).T)
        info[0] = int(inf)
        A.copy_(torch.tril(L))
        n = A.shape[0]
        return torch.zeros(((n + 127) // 128) * 128 * 128, dtype=torch.float64), info

    def trsm_right(self, L, dinv, B):
        if B.shape[0]:
            B.copy_(torch.linalg.solve_triangular(torch.tril(L), B.T.contiguous(), upper=False).T)

    def gemm_nt_sub(self, C, A, B):
        if C.numel():
            C.sub_(A @ B.T)

    def solve_lower_vec(self, L, dinv, v):
        return torch.linalg.solve_triangular(torch.tril(L), v.reshape(-1, 1), upper=False).reshape(-1)

    def matvec(self, A, v):
        return A @ v

    def sum_log_diag(self, L):
        return float(torch.log(torch.diagonal(L)).sum())

    def trsm_left(self, L, dinv, B):
        if B.numel():
            B.copy_(torch.linalg.solve_triangular(torch.tril(L), B, upper=False))

    def trsm_left_t(self, L, dinv, B):
        if B.numel():
            B.copy_(torch.linalg.solve_triangular(torch.tril(L).T, B, upper=True))

    def gemm_nn_sub(self, C, A, B):
        if C.numel():
            C.sub_(A @ B)

    def coldots(self, V, w):
        return V.T @ w, (V * V).sum(dim=0)

    def coldots_many(self, V, W):
        return W.T @ V, (V * V).sum(dim=0)

    def matmul(self, A, B):
        return A @ B

    def pairwise_variance(self, cov, xt, covparam):
        return torch.as_tensor(np.ascontiguousarray(cov(xt.numpy(), None, covparam, True)))

    def gemm_tn(self, A, B):
        return A.T @ B

    def gemm_tn_into(self, A, B, C):
        C.copy_(A.T @ B)

    def gemm_tn_acc(self, A, B, C):
        if C.numel():
            C.add_(A.T @ B)

    def grad_trace_cross(self, M, xr, xc, p, covparam, noise, F, G):
        """NumPy restatement of gpmp_matern_grad_trace_cross (oracle formulas: oracle/gp_oracle.py covariance_gradient_traces)"""
        import math

        from scipy.spatial.distance import cdist

        from oracle import gp_oracle as orc

        Mn = M.numpy().copy()
        if F is not None:
            Mn -= np.asarray(F) @ np.asarray(G).T
        th = np.asarray(covparam, dtype=np.float64)
        off = 2 if noise else 1
        inv = np.exp(th[off:])
        xs, ys = np.asarray(xr) * inv, np.asarray(xc) * inv
        H = cdist(xs, ys)
        s2 = math.exp(th[0])
        out = np.zeros(1 + xs.shape[1])
        out[0] = s2 * np.sum(Mn * orc.maternp_kernel(p, H))
        R = s2 * orc.maternp_dkernel_over_h(p, H) * Mn
        for j in range(xs.shape[1]):
            out[1 + j] = np.sum(R * (xs[:, j][:, None] - ys[:, j][None, :]) ** 2)
        return torch.as_tensor(out)

