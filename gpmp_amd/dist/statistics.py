"""What the GP layer asks of the block-cyclic factor: the kriging predictor (zero mean / universal kriging, weights on request),
NLL, REML, leave-one-out and the ML / REML value + analytic gradient (gpmp/core/kriging.py:35-67, 105-200;
gpmp/core/likelihood.py:18-129; gpmp/core/loo.py).  Mixed into ``BlockCyclicCholesky`` (cholesky.py)."""
from __future__ import annotations

import bisect
import math

import numpy as np
import torch
import torch.distributed as dist


class DistributedStatistics:
    def predict_zero_mean(self, cov, x, z, xt, covparam):
        """Zero-mean kriging from the distributed factor: see ``predict`` (no mean design)."""
        return self.predict(cov, x, z, xt, covparam)

    def predict(self, cov, x, z, xt, covparam, P=None, Pt=None, return_lambdas=False):
        """Posterior mean and variance at xt from the distributed factor of K(x, x).  Zero mean (P None:
        gpmp/core/kriging.py:35-67,170-199) restated as ONE solve, V = L^-1 K(x, xt), mean = V^T (L^-1 z),
        var = k(xt, xt) - colsumsq(V); with a linear predictor (universal kriging, kriging.py:70-116; P: n x q mean design at
        the observations, Pt: m x q at the prediction points) the Schur-complement form of gpmp_amd/core/kriging.py:
            R = Wp^T V - Pt^T,  mu = S^-1 R,  mean = V^T wz - mu^T (Wp^T wz),  var = k_tt - colsumsq(V) + sum(mu * R),
        [wz, Wp] = L^-1 [z, P], S = Wp^T Wp -- the (n + q) x (n + q) system of the reference is never formed.
        The prediction points are split over the process COLUMNS, the rows of V over the process ROWS; the local block
        K(x[rows owned], xt[column shard]) is one Gram call, and the only reductions are (2 + q) x m_c all-reduced inside each
        process column.  Returns (mean, variance, (j0, j1)): this process column's shard of the results (identical on the
        ranks of one process column), as NumPy arrays.  ``return_lambdas=True`` appends the kriging weights of model.py:305-306 as
        this RANK's block (local rows ``global_row_index()`` x prediction points j0:j1, device tensor):
        lambda = L^-T (V - Wp mu) -- the second solve of cholesky_solve, ``solve_upper_many``; V is consumed."""
        from .predict import shard_bounds

        if self.info:
            raise np.linalg.LinAlgError("the distributed factorisation failed (not positive definite): no prediction")
        g, ops = self.grid, self.ops
        x, xt = ops.asarray(x), ops.asarray(xt)
        z = np.asarray(z, dtype=np.float64).reshape(-1)
        q = 0 if P is None else np.asarray(P).reshape(self.n, -1).shape[1]
        j0, j1 = shard_bounds(xt.shape[0], g.pc, g.c)
        xtc = xt[j0:j1].contiguous()
        ri = self.global_row_index()
        xr = x[torch.as_tensor(ri, device=x.device)] if len(ri) else x[:0]
        if len(ri) and j1 > j0:
            Kit = ops.gram_block(cov, xr.contiguous(), xtc, covparam)
        else:
            Kit = ops.empty(len(ri), j1 - j0)
        V = self.solve_lower_many(Kit)
        Y = z.reshape(-1, 1) if q == 0 else np.hstack((z.reshape(-1, 1), np.asarray(P, dtype=np.float64).reshape(self.n, q)))
        W = self.solve_lower_few(Y)                                      # replicated L^-1 [z, P]
        dev = "cuda" if self.backend == "nccl" else "cpu"
        part = torch.zeros((2 + q, j1 - j0), dtype=torch.float64, device=dev)
        if len(ri) and j1 > j0:
            dots, ssq = ops.coldots_many(V, ops.asarray(W[torch.as_tensor(ri, device=W.device)]))
            part[: 1 + q], part[1 + q] = dots.to(dev), ssq.to(dev)
        if g.pr > 1:
            self._step_label = "predict_reduce"
            self._all_reduce(part, dist.ReduceOp.SUM, g.col_group, "mean_var")
        prior = ops.pairwise_variance(cov, xtc, covparam).to(dev) if j1 > j0 else part[1 + q]
        part = part.cpu().numpy()
        mean, reduction = part[0].copy(), part[1 + q].copy()
        if q:
            Wh = W.cpu().numpy()
            wz, Wp = Wh[:, 0], Wh[:, 1:]
            S = Wp.T @ Wp
            R = part[1: 1 + q] - np.asarray(Pt, dtype=np.float64).reshape(xt.shape[0], q)[j0:j1].T     # S mu
            mu = np.linalg.solve(0.5 * (S + S.T), R)
            mean = mean - (Wp.T @ wz) @ mu
            reduction = reduction - np.sum(mu * R, axis=0)
        if return_lambdas:
            if q and len(ri) and j1 > j0:
                # V <- V - Wp mu (rows owned x my points): a rank-q product on the library GEMM
                ops.gemm_nn_sub(V, ops.asarray(W[torch.as_tensor(ri, device=W.device)][:, 1:].contiguous()), ops.asarray(torch.as_tensor(mu)))
            lam = self.solve_upper_many(V)
            return mean, prior.cpu().numpy() - reduction, (j0, j1), lam
        return mean, prior.cpu().numpy() - reduction, (j0, j1)

    def negative_log_likelihood(self, z) -> float:
        """1/2 (n ln 2pi + ln|K| + z^T K^-1 z) -- gpmp/core/likelihood.py:18-52 on the distributed factor."""
        if self.info:
            return math.inf
        w = self.solve_lower_vector(z)
        return 0.5 * (self.n * math.log(2.0 * math.pi) + self.logdet() + float((w * w).sum().item()))

    def inverse_factor_local(self):
        """T = L^-1 in the FACTOR'S OWN 2-D block-cyclic layout (local rows x local columns, global indices
        ``global_row_index()`` / ``global_col_index()``): the many-right-hand-side solve on the identity, with the triangular
        structure exploited -- block row k of T is non-zero in the block columns J <= k only, a PREFIX of the local columns
        (``active_cols``) -- so the work is n^3 / 3 and every process column carries the same share of it (contiguous column
        shards would leave the last process column almost idle and the first with all of the work)."""
        ri, ci = self.global_row_index(), self.global_col_index()
        eye = self.ops.empty(len(ri), len(ci))
        eye.zero_()
        if len(ri) and len(ci):
            pos = {int(gc): lj for lj, gc in enumerate(ci)}
            hit = [(li, pos[int(gr)]) for li, gr in enumerate(ri) if int(gr) in pos]
            if hit:
                hr, hc = zip(*hit)
                eye[torch.as_tensor(hr, device=eye.device), torch.as_tensor(hc, device=eye.device)] = 1.0
        return self.solve_lower_many(eye, active_cols=lambda k: self.coff[self._first_col_after(k)])

    def negative_log_restricted_likelihood(self, z, P) -> float:
        """REML criterion (gpmp/core/likelihood.py:92-129) on the distributed factor, with the exact restatement the
        single-GPU path uses: for W an orthonormal basis of Null(P^T),
            ln|W^T K W| = ln|K| + ln|P^T K^-1 P| - ln|P^T P|,   (W^T z)^T (W^T K W)^-1 (W^T z) = z^T K^-1 z - b^T S^-1 b,
        S = P^T K^-1 P = Wp^T Wp, b = Wp^T wz with [wz, Wp] = L^-1 [z, P]: ONE forward solve with 1 + q replicated columns,
        then q x q algebra on the host.  P: the n x q mean design (replicated)."""
        if self.info:
            return math.inf
        z = np.asarray(z, dtype=np.float64).reshape(-1)
        P = np.asarray(P, dtype=np.float64).reshape(self.n, -1)
        q = P.shape[1]
        W = self.solve_lower_few(np.hstack((z.reshape(-1, 1), P))).cpu().numpy()
        wz, Wp = W[:, 0], W[:, 1:]
        S = Wp.T @ Wp
        b = Wp.T @ wz
        try:
            cS = np.linalg.cholesky(S)
            cP = np.linalg.cholesky(P.T @ P)
        except np.linalg.LinAlgError:
            return math.inf                       # rank-deficient mean design: the reference's safe_inf() convention
        y = np.linalg.solve(cS, b)
        logdet = self.logdet() + 2.0 * np.sum(np.log(np.diag(cS))) - 2.0 * np.sum(np.log(np.diag(cP)))
        return 0.5 * ((self.n - q) * math.log(2.0 * math.pi) + logdet + float(wz @ wz - y @ y))

    def _kinv_rows(self, T, Y, what):
        """From the local part of T = L^-1 (block-cyclic columns): X = K^-1 Y for the rows of this process column's column
        set (X = T^T (L^-1 Y): one pass over T + one all-reduce inside the process column) and diag(K^-1) there (column sums
        of squares of T).  Y: n x r replicated.  Returns (X (m_c x r), diag (m_c,)) as NumPy arrays, identical on the ranks of
        a process column."""
        g, ops = self.grid, self.ops
        ri, ci = self.global_row_index(), self.global_col_index()
        r = Y.shape[1]
        W = self.solve_lower_few(Y)                                      # replicated L^-1 Y
        dev = "cuda" if self.backend == "nccl" else "cpu"
        part = torch.zeros((r + 1, len(ci)), dtype=torch.float64, device=dev)
        if len(ri) and len(ci):
            dots, ssq = ops.coldots_many(T, ops.asarray(W[torch.as_tensor(ri, device=W.device)]))   # (r, m_c), (m_c,)
            part[:r], part[r] = dots.to(dev), ssq.to(dev)
        if g.pr > 1:
            self._step_label = what
            self._all_reduce(part, dist.ReduceOp.SUM, g.col_group, what)
        part = part.cpu().numpy()
        return part[:r].T.copy(), part[r].copy()

    def loo(self, z, P=None):
        """Leave-one-out predictions by virtual cross-validation (gpmp/core/loo.py:65-83 zero mean; :103-130 with a linear
        predictor, in the form Qinv = K^-1 - U S^-1 U^T, U = K^-1 P of gpmp_amd/core/loo.py) on the distributed factor.
        T = L^-1 in the factor's block-cyclic layout (inverse_factor_local: n^3 / 3 flops, balanced); diag(K^-1) = column
        sums of squares of T and K^-1 [z, P] = T^T (L^-1 [z, P]) are ONE pass over the local part of T + one all-reduce
        inside the process column.  Returns (zloo, sigma2loo, eloo, idx): the leave-one-out results of the observations
        ``idx`` (global indices: the block columns this process column owns; identical on its ranks), NumPy arrays."""
        if self.info:
            raise np.linalg.LinAlgError("the distributed factorisation failed (not positive definite): no leave-one-out")
        g = self.grid
        z = np.asarray(z, dtype=np.float64).reshape(-1)
        Y = z.reshape(-1, 1) if P is None else np.hstack((z.reshape(-1, 1), np.asarray(P, dtype=np.float64).reshape(self.n, -1)))
        idx = self.global_col_index()
        T = self.inverse_factor_local()
        X, dK = self._kinv_rows(T, Y, "loo")                             # K^-1 [z, P] (rows idx), diag(K^-1)
        del T
        zs = z[idx]
        if P is None:
            eloo = X[:, 0] / dK
            return zs - eloo, 1.0 / dK, eloo, idx
        # S = P^T K^-1 P and z^T K^-1 P need every column set: one small all-reduce over the process ROW (each set once)
        Pn = np.asarray(P, dtype=np.float64).reshape(self.n, -1)
        U = X[:, 1:]
        G = torch.as_tensor(np.vstack((zs @ U, Pn[idx].T @ U)))
        G = G.to("cuda") if self.backend == "nccl" else G
        if g.pc > 1:
            self._all_reduce(G, dist.ReduceOp.SUM, g.row_group, "loo_meanspace")
        G = G.cpu().numpy()
        S = 0.5 * (G[1:] + G[1:].T)
        US = U @ np.linalg.inv(S)
        Qz = X[:, 0] - US @ G[0]
        Qd = dK - np.sum(US * U, axis=1)
        eloo = Qz / Qd
        return zs - eloo, 1.0 / Qd, eloo, idx

    # ---- analytic gradient of the ML / REML criteria on the distributed factor
    def value_and_grad(self, x, z, covparam, p, noise=False, P=None):
        """Value and gradient of the zero-mean NLL (P None; gpmp/core/likelihood.py:18-52) or of the REML criterion with mean
        design P (likelihood.py:92-129) with respect to the covariance parameters, from the block-cyclic factor of
        K(x, x; covparam) -- the criterion a parameter fit at n beyond one GPU's HBM evaluates
        (gpmp/kernel/parameter_selection.py:35-124; the reference has no analytic form: finite differences / autograd).
            g_j = 1/2 sum_ik (Qinv_ik - beta_i beta_k) dK_ik/dtheta_j,  Qinv = K^-1 - U S^-1 U^T, U = K^-1 P, beta = Qinv z
        as gpmp_amd/core/gradients.py, with K^-1 = T^T T never assembled in one place: T = L^-1 is the many-right-hand-side
        solve on the identity (rows over the process rows, columns over the process columns); process column c forms the blocks
        (column set c, column set c') of T^T T from its own rows -- T travels around the process row (a ring of Pc / 2 shifts,
        blocks c != c' count twice) -- and traces each block against the matching block of dK in one fused pass
        (gpmp_matern_grad_trace_cross: dK is recomputed on the fly, the low-rank part is subtracted in registers).  Partial
        sums over the process rows need no matrix reduction: the trace is linear, so ONE all-reduce of 1 + d doubles ends it.
        Flops: n^3 / 3 for T (triangular structure exploited, balanced over the grid) + n^3 / 3 for the blocks (round 4: every
        unordered pair of block columns once, contraction from the later of the two on -- the blocked lauum's count)."""
        g, ops = self.grid, self.ops
        if self.info:
            return math.inf, np.zeros(len(covparam))
        x = np.asarray(x, dtype=np.float64)
        z = np.asarray(z, dtype=np.float64).reshape(-1)
        n, d = x.shape
        q = 0 if P is None else np.asarray(P).reshape(n, -1).shape[1]
        Pn = None if P is None else np.asarray(P, dtype=np.float64).reshape(n, q)
        off = 2 if noise else 1
        th = np.asarray(covparam, dtype=np.float64)
        sigma2 = math.exp(th[0])
        dev = "cuda" if self.backend == "nccl" else "cpu"
        # ---- T = L^-1 in the factor's block-cyclic layout, X = K^-1 [z, P] = T^T (L^-1 [z, P]) for my column set
        ci = self.global_col_index()
        T = self.inverse_factor_local()
        Y = z.reshape(-1, 1) if Pn is None else np.hstack((z.reshape(-1, 1), Pn))
        r1 = Y.shape[1]
        X, dK = self._kinv_rows(T, Y, "grad_reduce")
        # every rank needs K^-1 [z, P] for ALL rows (the low-rank factors of the other column sets): a small all-gather
        # along the process row, done as an all-reduce of a zero-padded n x (1 + q) array; tr(K^-1) rides along
        Xfull = torch.zeros((n + 1, r1), dtype=torch.float64, device=dev)
        if len(ci):
            Xfull[torch.as_tensor(ci, device=dev)] = torch.as_tensor(X, device=dev)
            Xfull[n, 0] = float(dK.sum())
        if g.pc > 1:
            self._all_reduce(Xfull, dist.ReduceOp.SUM, g.row_group, "kinv_zp")
        trKinv = Xfull[n, :1].clone()
        Xfull = Xfull[:n]
        Xh = Xfull.cpu().numpy()
        alpha = Xh[:, 0]
        logdet = self.logdet()
        if q == 0:
            value = 0.5 * (n * math.log(2.0 * math.pi) + logdet + float(z @ alpha))
            Fh = Gh = alpha.reshape(-1, 1)
        else:
            U = Xh[:, 1:]
            S = Pn.T @ U
            S = 0.5 * (S + S.T)
            b = Pn.T @ alpha
            try:
                cS, cP = np.linalg.cholesky(S), np.linalg.cholesky(Pn.T @ Pn)
            except np.linalg.LinAlgError:
                return math.inf, np.zeros(len(th))
            Sinv = np.linalg.inv(S)
            US = U @ Sinv
            beta = alpha - US @ b
            value = 0.5 * ((n - q) * math.log(2.0 * math.pi) + logdet + 2.0 * np.sum(np.log(np.diag(cS))) - 2.0 * np.sum(np.log(np.diag(cP)))
                           + float(z @ alpha - b @ (Sinv @ b)))
            Fh, Gh = np.hstack((US, beta.reshape(-1, 1))), np.hstack((U, beta.reshape(-1, 1)))
        # ---- K^-1 = T^T T against dK, block pair by block pair, every unordered pair of block columns {I, J} ONCE with its exact
        # contraction range -- the distributed form of the blocked lauum (gpmp_lauum_lower; numpy_backend.py:458-463 forms the
        # inverse): block (I, J) = sum over the block rows k >= max(I, J) of T[k, I]^T T[k, J], n^3 / 3 flops in all (round 3 formed
        # whole (column set, column set) blocks with the contraction cut on one side only: 0.75 n^3).  The rows k are split over the
        # process rows and the trace is linear, so the partial products are traced where they are (no matrix reduction); the
        # column sets meet around the process row (ring of Pc / 2 shifts).  Per shift, with I in my column set c, J in set c2:
        #   row strip of I:     M[I, J <= I] = T[ro(I):, I]^T T2[ro(I):, J <= I]     (the J <= I are a PREFIX of T2's local columns)
        #   column strip of J:  M[I < J, J]  = T[ro(J):, I < J]^T T2[ro(J):, J]      (the I < J are a prefix of T's local columns)
        # sft = 0 (c2 = c): row strips only (the lower block triangle; diagonal blocks count once, the others twice);
        # 0 < sft < Pc / 2: both kinds = the whole (c, c2) block, twice (its mirror (c2, c) is never formed);
        # sft = Pc / 2 (Pc even): row strips only, twice -- the partner rank's row strips are the mirror of my column strips.
        xs_c = x[ci]
        tot = torch.zeros(1 + d, dtype=torch.float64, device=dev)
        half = g.pc // 2
        nrows_loc = self.roff[-1]
        my_blocks = self.col_blocks

        def row_start(I):                         # first local row of a block row >= I
            return self.roff[bisect.bisect_left(self.row_blocks, I)]

        for sft in range(half + 1):
            c2 = (g.c + sft) % g.pc
            T2 = T if sft == 0 else ops.asarray(self._ring_shift(T, sft))
            strips_only = sft == 0 or (g.pc % 2 == 0 and sft == half)
            blocks2 = g.local_col_blocks(self.nblocks, c2)
            off2 = self._offsets(blocks2)
            ci2 = np.concatenate([np.arange(J * self.nb, J * self.nb + self.bs(J)) for J in blocks2]) if blocks2 else np.zeros(0, dtype=np.int64)
            if not (len(ci) and len(ci2) and len(self.row_blocks)):
                continue
            lowG_all = Gh[ci2] if g.r == 0 else None          # the low-rank part enters exactly once per block: on process row 0
            # round 5: with the C ABI's local half (nb = 1024) ALL blocks of this shift are ONE launch -- a staircase tile set (the
            # blocks J <= I) with the contraction start of every block row / block column in the kernel's k loop -- instead of one
            # product per block column, most of them too small to fill the machine
            fused = self._abi and hasattr(ops, "inverse_gram") and self.nb == 1024
            Mfused = None
            if fused:
                Mfused = ops.empty(len(ci), len(ci2))
                ops.inverse_gram(T, T2, Mfused, self._lay, c2, strips_only)
            if strips_only:
                for li, I in enumerate(my_blocks):
                    oI, wI = self.coff[li], self.bs(I)
                    pref = off2[bisect.bisect_right(blocks2, I)]                  # local columns of the J <= I in set c2
                    ro = row_start(I)
                    if pref == 0:
                        continue
                    if fused:
                        strip = Mfused[oI:oI + wI, :pref]
                    else:
                        strip = ops.empty(wI, pref)
                        if ro < nrows_loc:
                            ops.gemm_tn_into(T[ro:, oI:oI + wI], T2[ro:, :pref], strip)
                        else:
                            strip.zero_()                                        # no local row below: only the low-rank part is left
                    lowF = Fh[ci[oI:oI + wI]] if g.r == 0 else None
                    xr = xs_c[oI:oI + wI]
                    if sft == 0:
                        # the diagonal block (I, I) is the last wI columns of the strip: once; everything left of it: twice
                        if pref > wI:
                            tot += 2.0 * ops.grad_trace_cross(strip[:, :pref - wI], xr, x[ci2[:pref - wI]], p, th, noise, lowF,
                                                              None if lowG_all is None else lowG_all[:pref - wI]).to(dev)
                        tot += ops.grad_trace_cross(strip[:, pref - wI:], xr, x[ci2[pref - wI:pref]], p, th, noise, lowF,
                                                    None if lowG_all is None else lowG_all[pref - wI:pref]).to(dev)
                    else:
                        tot += 2.0 * ops.grad_trace_cross(strip, xr, x[ci2[:pref]], p, th, noise, lowF,
                                                          None if lowG_all is None else lowG_all[:pref]).to(dev)
                    del strip
            elif fused:
                lowF = Fh[ci] if g.r == 0 else None
                tot += 2.0 * ops.grad_trace_cross(Mfused, xs_c, x[ci2], p, th, noise, lowF, lowG_all).to(dev)
            else:
                Mblk = ops.empty(len(ci), len(ci2))
                for li, I in enumerate(my_blocks):                               # row strips: J <= I
                    oI, wI = self.coff[li], self.bs(I)
                    pref = off2[bisect.bisect_right(blocks2, I)]
                    ro = row_start(I)
                    if pref == 0:
                        continue
                    if ro < nrows_loc:
                        ops.gemm_tn_into(T[ro:, oI:oI + wI], T2[ro:, :pref], Mblk[oI:oI + wI, :pref])
                    else:
                        Mblk[oI:oI + wI, :pref].zero_()
                for lj, J in enumerate(blocks2):                                 # column strips: I < J
                    o2, w2 = off2[lj], self.bs(J)
                    pref = self.coff[bisect.bisect_left(my_blocks, J)]
                    ro = row_start(J)
                    if pref == 0:
                        continue
                    if ro < nrows_loc:
                        ops.gemm_tn_into(T[ro:, :pref], T2[ro:, o2:o2 + w2], Mblk[:pref, o2:o2 + w2])
                    else:
                        Mblk[:pref, o2:o2 + w2].zero_()
                lowF = Fh[ci] if g.r == 0 else None
                tot += 2.0 * ops.grad_trace_cross(Mblk, xs_c, x[ci2], p, th, noise, lowF, lowG_all).to(dev)
                del Mblk
        self._step_label = "grad_total"
        self._all_reduce(tot, dist.ReduceOp.SUM, g.world_group, "grad_traces")
        tot = tot.cpu().numpy()
        trM = float(trKinv.item()) - float(np.sum(Fh * Gh))
        grad = np.zeros(len(th))
        grad[0] = tot[0] + (0.0 if noise else 10.0 * sigma2 * float(np.finfo(np.float64).eps) * trM)
        if noise:
            grad[1] = math.exp(th[1]) * trM
        grad[off:] = tot[1:]
        return value, 0.5 * grad
