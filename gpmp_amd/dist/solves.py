"""Triangular solves on the block-cyclic factor: L^-1 for a replicated vector or a few columns (one sweep over the block columns,
reduce + broadcast per block), and the many-right-hand-side forward / backward solves of the predictor as three-stream schedules
(prefetch of the next panel's broadcast, the chain on the side stream, the bulk product on the caller's stream).  Mixed into
``BlockCyclicCholesky`` (cholesky.py), which owns the layout, the factor and the communication helpers these methods call."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from .streams import _Streams


class DistributedSolves:
    def solve_lower_vector(self, z):
        """w = L^-1 z for a replicated vector z (n,): see solve_lower_few."""
        return self.solve_lower_few(np.asarray(z, dtype=np.float64).reshape(-1, 1))[:, 0]

    def solve_lower_few(self, Z):
        """W = L^-1 Z for a REPLICATED n x r matrix with a few columns ([z, P] of REML / LOO: r = 1 + q): block forward
        substitution with one broadcast and one all-reduce of the update per block column.  Returns the replicated W
        (device tensor with RCCL, CPU tensor with gloo)."""
        g, ops, nb = self.grid, self.ops, self.nb
        dev = "cuda" if self.backend == "nccl" else "cpu"
        w = torch.as_tensor(np.asarray(Z, dtype=np.float64)).to(dev).clone()
        if w.dim() != 2 or w.shape[0] != self.n:
            raise ValueError("expected an n x r matrix")
        r = w.shape[1]
        gri = torch.as_tensor(self.global_row_index(), dtype=torch.int64, device=dev)    # global row of every local row
        for k in range(self.nblocks):
            rd, cd = g.owner_row(k), g.owner_col(k)
            bk = self.bs(k)
            k0 = k * nb
            wk = w[k0:k0 + bk].clone()
            if g.r == rd and g.c == cd:
                Lkk, dinv = self.diag_cache[k]
                if r == 1:
                    wk = ops.solve_lower_vec(Lkk, dinv, ops.asarray(wk[:, 0])).to(dev).reshape(-1, 1)
                else:
                    blk = ops.empty(bk, r)
                    blk.copy_(ops.asarray(wk))
                    ops.trsm_left(Lkk, dinv, blk)
                    wk = blk.to(dev)
            wk = wk.contiguous()
            self._step_label = f"vec{k}"
            self._world_bcast(wk, g.rank_of(rd, cd))
            w[k0:k0 + bk] = wk
            rest = self.n - (k0 + bk)
            if rest <= 0:
                continue
            delta = torch.zeros((rest, r), dtype=torch.float64, device=dev)
            if g.c == cd:
                i0 = self._first_row_after(k)
                if i0 < len(self.row_blocks):
                    lj = k // g.pc
                    P = self.A[self.roff[i0]:, self.coff[lj]:self.coff[lj + 1]]
                    if r == 1:
                        upd = ops.matvec(P, ops.asarray(wk[:, 0])).to(dev).reshape(-1, 1)
                    else:
                        upd = ops.matmul(P, ops.asarray(wk)).to(dev)
                    delta[gri[self.roff[i0]:] - (k0 + bk)] = upd
            self._all_reduce(delta, dist.ReduceOp.SUM, g.world_group, "vec_update")
            w[k0 + bk:] -= delta
        return w

    # ---- many right-hand sides on the distributed factor (prediction beyond one GPU's HBM)
    def solve_lower_many(self, Bloc: torch.Tensor, overlap: Optional[bool] = None, profile: Optional[bool] = None, active_cols=None) -> torch.Tensor:
        """V = L^-1 B in place for a right-hand side laid out like the factor's rows: ``Bloc`` holds the block rows this
        process row owns (self.local_rows() x m_c) of THIS process column's shard of the columns of B (the columns of B are
        split over the process columns, gpmp_amd.dist.shard_bounds(m, Pc, c)).  Per block column k:
          P(k)  prefetch: L_kk (+ its diagonal-block inverses) travels along process row k mod Pr, then the panel
                L_{I>k, k} along every process row                                     [row communicators; factor data only]
          C(k)  chain:    on process row k mod Pr: B_k -= L_{k,k-1} X_{k-1} (the ONE block row the next solve needs),
                X_k = L_kk^-1 B_k; X_k travels down every process column                 [column communicators]
          U(k)  update:   B_I -= L_Ik X_k for the block rows I > k+1 this rank owns       [local GEMM, n^2 m flops in total]
        Schedule (the factorisation's look-ahead pattern on the same three streams): while U(k) runs on the caller's
        stream, P(k+1) runs on the prefetch stream and C(k+1) on the side stream -- the broadcasts of step k+1 and the small
        products on the critical path are hidden behind the bulk GEMM of step k; two buffers per message kind.  Every
        communicator is used from ONE stream (row communicators: prefetch stream, column communicators: side stream) and in
        the same order on all of its members.  ``overlap=False`` (GPMP_DIST_SOLVE_OVERLAP=0) issues the same operations in
        the same order on the caller's stream alone.
        ``active_cols``: optional k -> number of LEADING local columns of ``Bloc`` that can be non-zero in block row k of the
        solution (a function of k and of the process column only).  For a right-hand side with that structure -- the identity
        in the factor's own block-cyclic column layout, whose solution L^-1 is lower triangular -- C(k) and U(k) then touch
        only those columns: n^3 / 3 flops instead of n^3, evenly spread over the process columns."""
        g, ops, nb = self.grid, self.ops, self.nb
        if overlap is None:
            overlap = os.environ.get("GPMP_DIST_SOLVE_OVERLAP", "1") != "0"
        if profile is not None:
            self.profile = profile
        mloc = Bloc.shape[1]
        nblk = self.nblocks
        row_members = [g.rank_of(g.r, cc) for cc in range(g.pc)]
        col_members = [g.rank_of(rr, g.c) for rr in range(g.pr)]
        nbk = self.bs(0)
        ld0 = (nbk + 15) // 16 * 16
        nd0 = ((nbk + 127) // 128) * 128 * 128
        Lbuf = [self._flat(nbk * ld0 + nd0) for _ in range(2)]
        Pbuf = [ops.empty(self.local_rows(), nb) for _ in range(2)]
        Xbuf = [ops.empty(nb, mloc) for _ in range(2)]
        self._marks = []
        st = self._st = _Streams(getattr(ops, "device", None) if overlap else None, 0, getattr(ops, "lib", None))
        pre_ctx, side_ctx = st.diag_ctx, st.side_ctx          # the "diagonal" stream of the factorisation carries the prefetch here
        start = st.record(False)
        l_ready, p_ready, x_ready, u_done = {}, {}, {}, {}

        def views(k):
            bk = self.bs(k)
            ldk = (bk + 15) // 16 * 16
            ndinv = ((bk + 127) // 128) * 128 * 128
            buf = Lbuf[k % 2][: bk * ldk + ndinv]
            return bk, buf, buf[: bk * ldk].view(bk, ldk)[:, :bk], buf[bk * ldk:]

        def prefetch(k):
            rd, cd = g.owner_row(k), g.owner_col(k)
            bk, buf, Lkk, dinv = views(k)
            self._step_label = f"solve_pre{k}"
            with pre_ctx():
                st.wait_diag(start)
                st.wait_diag(x_ready.get(k - 1))     # C(k-1) has read L buffer (k-2) and panel k-2
                st.wait_diag(u_done.get(k - 2))      # U(k-2) has read panel k-2
                with self._phase("solve_prefetch"):
                    if g.r == rd:
                        if g.c == cd:
                            L0, d0 = self.diag_cache[k]
                            Lkk.copy_(L0)
                            dinv.copy_(d0[: dinv.numel()])
                        if g.pc > 1:
                            self._bcast(buf, g.rank_of(rd, cd), g.row_group, row_members)
                    l_ready[k] = st.record_diag()
                    i0 = self._first_row_after(k)
                    Mr = self.roff[-1] - self.roff[i0]
                    if Mr > 0:
                        panel = Pbuf[k % 2][:Mr, :bk]
                        if g.c == cd:
                            lj = k // g.pc
                            panel.copy_(self.A[self.roff[i0]:, self.coff[lj]:self.coff[lj + 1]])
                        if g.pc > 1:
                            self._bcast(panel, g.rank_of(g.r, cd), g.row_group, row_members)
                    p_ready[k] = st.record_diag()

        def ncols(k):
            return mloc if active_cols is None else max(0, min(mloc, int(active_cols(k))))

        def chain(k):
            rd = g.owner_row(k)
            bk, buf, Lkk, dinv = views(k)
            na = ncols(k)
            xk = Xbuf[k % 2][:bk, :na]
            self._step_label = f"solve_chain{k}"
            with side_ctx():
                st.wait(True, start)
                st.wait(True, l_ready.get(k))
                st.wait(True, p_ready.get(k - 1))
                st.wait(True, u_done.get(k - 2))     # block row k has received the updates 0 ... k-2, X buffer (k-2) is free
                with self._phase("solve_chain"):
                    if g.r == rd and na:
                        li = k // g.pr
                        Bk = Bloc[self.roff[li]:self.roff[li + 1], :na]
                        npv = ncols(k - 1) if k > 0 else 0
                        if npv:
                            # update k-1 of this ONE block row (the bulk update k-1 skips it): first row of panel k-1
                            bp = self.bs(k - 1)
                            ip = self._first_row_after(k - 1)
                            off = self.roff[li] - self.roff[ip]
                            ops.gemm_nn_sub(Bk[:, :npv], Pbuf[(k - 1) % 2][off: off + bk, :bp], Xbuf[(k - 1) % 2][:bp, :npv])
                        ops.trsm_left(Lkk, dinv, Bk)
                        xk.copy_(Bk)
                    if g.pr > 1 and na:
                        self._bcast(xk, g.rank_of(rd, g.c), g.col_group, col_members)
                x_ready[k] = st.record(True)

        def update(k):
            bk = self.bs(k)
            i0 = self._first_row_after(k)
            i1 = self._first_row_after(k + 1)           # block row k+1 (if owned) is updated by C(k+1)
            self._step_label = f"solve_update{k}"
            st.wait(False, x_ready.get(k))
            st.wait(False, p_ready.get(k))
            na = ncols(k)
            with st.main_ctx(), self._phase("solve_update"):
                if na and self.roff[-1] - self.roff[i1] > 0:
                    off = self.roff[i1] - self.roff[i0]
                    ops.gemm_nn_sub(Bloc[self.roff[i1]:, :na], Pbuf[k % 2][off: self.roff[-1] - self.roff[i0], :bk], Xbuf[k % 2][:bk, :na])
            u_done[k] = st.record(False)

        prefetch(0)
        chain(0)
        for k in range(nblk):
            if k + 1 < nblk:
                prefetch(k + 1)
                chain(k + 1)
            update(k)
            for evs in (l_ready, p_ready, x_ready, u_done):      # keep three steps of events
                evs.pop(k - 3, None)
        st.wait(False, x_ready.get(nblk - 1))
        st.wait(False, p_ready.get(nblk - 1))
        st.close()
        return Bloc

    def solve_upper_many(self, Bloc: torch.Tensor, overlap: Optional[bool] = None, profile: Optional[bool] = None) -> torch.Tensor:
        """X = L^-T B in place for a right-hand side distributed like ``solve_lower_many``'s (rows block-cyclic over the process
        rows, columns sharded over the process columns): the SECOND solve of ``cholesky_solve`` (gpmp/num/numpy_backend.py:468),
        which the kriging WEIGHTS lambda_t = L^-T (L^-1 Kit) need (gpmp/core/kriging.py:62, model.py:305-306).  Left-looking
        backward substitution over the block columns k = nblk-1 ... 0:
            S_k = sum_{I > k} L_Ik^T X_I,   X_k = L_kk^-T (B_k - S_k)
        every rank multiplies ITS rows of panel k with ITS rows of X (TN products with a long contraction), the partial sums are
        REDUCED inside the process column to process row k mod Pr, and X_k stays where it lives: nothing is broadcast back.  Per
        block column k, split like the forward solve so that the bulk never waits for the step before it:
          P(k)  prefetch: L_kk (+ its diagonal-block inverses) along process row k mod Pr, the panel L_{I>k, k} along every
                process row                                                            [row communicators; factor data only]
          U(k)  bulk:     S_k <- sum over the local block rows I >= k+2 of L_Ik^T X_I  [local TN GEMM, n^2 m flops in total]
                          -- needs X_{k+2}, NOT X_{k+1}
          C(k)  chain:    the process row that owns block row k+1 adds the ONE missing term L_{k+1,k}^T X_{k+1}; reduce inside
                the process column; on process row k mod Pr: X_k = L_kk^-T (B_k - S_k)   [column communicators]
        Schedule (round 5; the forward solve's pattern on the same three streams): U(k-1) runs on the caller's stream and
        P(k-1) on the prefetch stream WHILE C(k) -- small product, reduce, 1024-row triangular solve -- runs on the side stream;
        two buffers per message kind and per partial sum.  Row communicators are used from the prefetch stream only, column
        communicators from the side stream only, in decreasing k on every member.  ``overlap=False`` (GPMP_DIST_SOLVE_OVERLAP=0)
        issues the same operations in the same order on the caller's stream alone."""
        g, ops, nb = self.grid, self.ops, self.nb
        if overlap is None:
            overlap = os.environ.get("GPMP_DIST_SOLVE_OVERLAP", "1") != "0"
        if profile is not None:
            self.profile = profile
        mloc = Bloc.shape[1]
        nblk = self.nblocks
        row_members = [g.rank_of(g.r, cc) for cc in range(g.pc)]
        nbk = self.bs(0)
        ld0 = (nbk + 15) // 16 * 16
        nd0 = ((nbk + 127) // 128) * 128 * 128
        Lbuf = [self._flat(nbk * ld0 + nd0) for _ in range(2)]
        Pbuf = [ops.empty(self.local_rows(), nb) for _ in range(2)]
        Sbuf = [ops.empty(nb, mloc) for _ in range(2)]
        self._marks = []
        st = self._st = _Streams(getattr(ops, "device", None) if overlap else None, 0, getattr(ops, "lib", None))
        pre_ctx, side_ctx = st.diag_ctx, st.side_ctx          # the "diagonal" stream of the factorisation carries the prefetch here
        start = st.record(False)
        l_ready, p_ready, x_done, u_done = {}, {}, {}, {}

        def views(k):
            bk = self.bs(k)
            ldk = (bk + 15) // 16 * 16
            ndinv = ((bk + 127) // 128) * 128 * 128
            buf = Lbuf[k % 2][: bk * ldk + ndinv]
            return bk, buf, buf[: bk * ldk].view(bk, ldk)[:, :bk], buf[bk * ldk:]

        def prefetch(k):
            rd, cd = g.owner_row(k), g.owner_col(k)
            bk, buf, Lkk, dinv = views(k)
            self._step_label = f"bsolve_pre{k}"
            with pre_ctx():
                st.wait_diag(start)
                st.wait_diag(x_done.get(k + 2))      # C(k+2) has used L buffer and panel buffer (k+2) % 2
                st.wait_diag(u_done.get(k + 2))      # U(k+2) has read panel k+2
                with self._phase("bsolve_prefetch"):
                    if g.r == rd:
                        if g.c == cd:
                            L0, d0 = self.diag_cache[k]
                            Lkk.copy_(L0)
                            dinv.copy_(d0[: dinv.numel()])
                        if g.pc > 1:
                            self._bcast(buf, g.rank_of(rd, cd), g.row_group, row_members)
                    l_ready[k] = st.record_diag()
                    i0 = self._first_row_after(k)
                    Mr = self.roff[-1] - self.roff[i0]
                    if Mr > 0:
                        panel = Pbuf[k % 2][:Mr, :bk]
                        if g.c == cd:
                            lj = k // g.pc
                            panel.copy_(self.A[self.roff[i0]:, self.coff[lj]:self.coff[lj + 1]])
                        if g.pc > 1:
                            self._bcast(panel, g.rank_of(g.r, cd), g.row_group, row_members)
                    p_ready[k] = st.record_diag()

        def bulk(k):
            """S_k <- the block rows I >= k+2 of this rank (zero when it has none): everything of S_k that X_{k+1} is not part of"""
            bk = self.bs(k)
            i0 = self._first_row_after(k)
            i2 = self._first_row_after(k + 1)
            self._step_label = f"bsolve_bulk{k}"
            st.wait(False, p_ready.get(k))
            st.wait(False, x_done.get(k + 2))        # X_{k+2} is final; C(k+2) has finished with partial-sum buffer k % 2
            with st.main_ctx(), self._phase("bsolve_bulk"):
                if mloc:
                    S = Sbuf[k % 2][:bk, :]
                    if self.roff[-1] - self.roff[i2] > 0:
                        off = self.roff[i2] - self.roff[i0]
                        ops.gemm_tn_into(Pbuf[k % 2][off: self.roff[-1] - self.roff[i0], :bk], Bloc[self.roff[i2]:, :], S)
                    else:
                        S.zero_()
            u_done[k] = st.record(False)

        def chain(k):
            rd = g.owner_row(k)
            bk, buf, Lkk, dinv = views(k)
            self._step_label = f"bsolve{k}"
            with side_ctx():
                st.wait(True, start)
                st.wait(True, l_ready.get(k))
                st.wait(True, p_ready.get(k))
                st.wait(True, u_done.get(k))
                st.wait(True, x_done.get(k + 1))
                with self._phase("bsolve_chain"):
                    if mloc:
                        S = Sbuf[k % 2][:bk, :]
                        if k + 1 < nblk and g.r == g.owner_row(k + 1):
                            # the one term the bulk product left out: block row k+1, the FIRST rows of this rank's panel k
                            li1 = (k + 1) // g.pr
                            b1 = self.bs(k + 1)
                            ops.gemm_tn_acc(Pbuf[k % 2][:b1, :bk], Bloc[self.roff[li1]:self.roff[li1 + 1], :], S)
                        if g.pr > 1:
                            self._reduce(S, g.rank_of(rd, g.c), g.col_group, "bsolve")
                        if g.r == rd:
                            li = k // g.pr
                            Bk = Bloc[self.roff[li]:self.roff[li + 1], :]
                            Bk.sub_(S)
                            ops.trsm_left_t(Lkk, dinv, Bk)
                x_done[k] = st.record(True)

        prefetch(nblk - 1)
        bulk(nblk - 1)
        for k in range(nblk - 1, -1, -1):
            if k >= 1:
                prefetch(k - 1)
                bulk(k - 1)                           # independent of C(k): runs beside it
            chain(k)
            for evs in (l_ready, p_ready, x_done, u_done):      # keep three steps of events
                evs.pop(k + 3, None)
        st.wait(False, x_done.get(0))
        st.wait(False, p_ready.get(0))
        st.close()
        return Bloc
