"""2-D block-cyclic Cholesky of K + nugget over a Pr x Pc process grid, and the NLL on top of it.

Layout (ScaLAPACK style): global block (I, J) of size nb lives on rank (I mod Pr, J mod Pc) at local
block (I div Pr, J div Pc) of one dense row-major local matrix.  Because the local matrix is
K(x[rows owned], x[cols owned]), it is built by ONE cross-covariance Gram call on two gathered point
subsets -- no communication.

Right-looking schedule per block column k (owner column cd = k mod Pc):
  1. owner (k mod Pr, cd) factors the diagonal block (single-GPU blocked potrf, MFMA)           [local]
  2. L_kk + its diagonal-block inverses -> broadcast down process column cd                      [RCCL]
  3. ranks of column cd:  panel  A_Ik <- A_Ik L_kk^-T  for their block rows I > k               [local]
  4. every process row r: panel piece broadcast along the row from (r, cd)                        [RCCL]
  5. every process column c: the blocks J > k with J mod Pc == c are exchanged inside the column
     (one broadcast per process row) -> the "transposed" operand of the update                    [RCCL]
  6. every rank: A_IJ -= L_Ik L_Jk^T for its blocks I >= J > k: ONE launch over the staircase of local
     blocks (nb = 1024; other block sizes: a staircase of GEMMs over groups of 4 local block rows)  [local]
Collectives are point-to-point-friendly broadcasts of (n - k nb) nb / Pr resp. / Pc doubles; scalars
(log-det, quadratic form) use one all-reduce of a few doubles.
"""
from __future__ import annotations

import bisect
import contextlib
import math
import os
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .grid import ProcessGrid
from .local_ops import HipLocalOps
from .solves import DistributedSolves
from .statistics import DistributedStatistics
from .streams import _comm_tensor, _gloo_cuda_guard, _Streams


class BlockCyclicCholesky(DistributedSolves, DistributedStatistics):
    """K = L L^T with K 2-D block-cyclic over ``grid``; keeps the local factor for NLL evaluations.

    ``transport``: "bcast" (one RCCL broadcast per message) or "p2p" (the root sends to every peer of the group in one
    grouped send/recv batch: xGMI is point-to-point, so the copies leave the root over separate links instead of
    following a ring).  Default from GPMP_DIST_TRANSPORT, else "bcast".
    ``lookahead``: prepare panel k+1 (column update, diagonal factor, panel solve, all broadcasts) on a side stream
    while the bulk of update k runs on the caller's stream.
    ``profile``: bracket the phases of every step with HIP events (no synchronisation); ``phase_times()`` then returns
    the summed milliseconds per phase on this rank -- diag (factor + column broadcast), trsm, row_bcast, col_exchange
    and lookahead_update on the side stream, update on the caller's stream.
    ``reserve_cus``: run the bulk updates on a stream that leaves this many CUs (one per XCD first) to the panel chain
    (default GPMP_DIST_RESERVE_CUS, else 0).
    ``step_abi`` / ``panel_via_inverse``: test hooks -- False runs a step's local arithmetic through the tensor-level code
    instead of the C ABI's ``gpmp_dist_*`` / the panel solves by substitution instead of ONE product with inv(L_kk)."""

    def __init__(self, grid: ProcessGrid, n: int, nb: int = 1024, ops=None, transport: Optional[str] = None,
                 lookahead: bool = True, profile: bool = False, reserve_cus: Optional[int] = None, step_abi: bool = True,
                 panel_via_inverse: bool = True):
        if nb % 128 != 0:
            raise ValueError("block size must be a multiple of 128 (the GEMM tile)")
        self.grid, self.n, self.nb = grid, n, nb
        self.ops = ops if ops is not None else HipLocalOps()
        self.backend = dist.get_backend(grid.world_group)
        self.transport = transport or os.environ.get("GPMP_DIST_TRANSPORT", "bcast")
        if self.transport not in ("bcast", "p2p"):
            raise ValueError("transport must be 'bcast' or 'p2p'")
        self.lookahead = lookahead
        self.panel_via_inverse = bool(panel_via_inverse)     # False: panel solves by substitution (tests compare both)
        self.reserve_cus = int(os.environ.get("GPMP_DIST_RESERVE_CUS", "0")) if reserve_cus is None else int(reserve_cus)
        self.profile = profile       # record per-phase HIP events in factor(); read them with phase_times()
        self._marks = []             # (phase, start event, end event)
        self._st = None
        self.nblocks = (n + nb - 1) // nb
        self.row_blocks = grid.local_row_blocks(self.nblocks)
        self.col_blocks = grid.local_col_blocks(self.nblocks)
        self.roff = self._offsets(self.row_blocks)
        self.coff = self._offsets(self.col_blocks)
        self.A = None            # local matrix (rows owned x cols owned)
        self.info = 0
        self.diag_cache = {}     # k -> (L_kk, dinv) on the ranks of the owning process column
        self.bytes_received = 0
        # issue log (tests, tools/dist_issue_order.py): when a list, every collective this rank enqueues is appended as
        # (communicator tag, operation, root rank, doubles, step label, stream role) in HOST ISSUE ORDER -- the order RCCL
        # sees.  RCCL needs the per-communicator sequences to be identical on all members of a communicator.
        self.oplog = None
        self._step_label = None
        self._lay = (n, nb, grid.pr, grid.pc, grid.r, grid.c)
        # local arithmetic of a step through the C ABI's gpmp_dist_* (HipLocalOps) or through the tensor-level code below
        # (``step_abi=False``: the tensor-level code with the real kernels -- tests compare the two bit for bit)
        self._abi = bool(getattr(self.ops, "step_abi", False)) and nb <= 1024 and bool(step_abi)

    # ---- index helpers
    def bs(self, I: int) -> int:
        return min(self.nb, self.n - I * self.nb)

    def _offsets(self, blocks: List[int]):
        off, acc = [], 0
        for I in blocks:
            off.append(acc)
            acc += self.bs(I)
        off.append(acc)
        return off

    def local_rows(self):
        return self.roff[-1]

    def local_cols(self):
        return self.coff[-1]

    def global_row_index(self):
        return np.concatenate([np.arange(I * self.nb, I * self.nb + self.bs(I)) for I in self.row_blocks]) if self.row_blocks else np.zeros(0, dtype=np.int64)

    def global_col_index(self):
        return np.concatenate([np.arange(J * self.nb, J * self.nb + self.bs(J)) for J in self.col_blocks]) if self.col_blocks else np.zeros(0, dtype=np.int64)

    def _first_row_after(self, k: int) -> int:
        """Local index of the first owned block row I > k."""
        return bisect.bisect_right(self.row_blocks, k)

    def _first_col_after(self, k: int) -> int:
        return bisect.bisect_right(self.col_blocks, k)

    # ---- build
    def build_local_gram(self, cov, x, covparam, diag_add: float):
        """Local part of K(x, x) + diag_add I: one cross-covariance call on the owned row / column points."""
        ops = self.ops
        x = ops.asarray(x)
        ri, ci = self.global_row_index(), self.global_col_index()
        xr = x[torch.as_tensor(ri, device=x.device)] if len(ri) else x[:0]
        xc = x[torch.as_tensor(ci, device=x.device)] if len(ci) else x[:0]
        if len(ri) == 0 or len(ci) == 0:
            self.A = ops.empty(len(ri), len(ci))
            return self.A
        A = ops.gram_block(cov, xr.contiguous(), xc.contiguous(), covparam)
        # nugget / noise on the global diagonal entries this rank owns
        for li, I in enumerate(self.row_blocks):
            if I % self.grid.pc == self.grid.c:
                lj = I // self.grid.pc
                blk = A[self.roff[li]:self.roff[li + 1], self.coff[lj]:self.coff[lj + 1]]
                torch.diagonal(blk).add_(diag_add)
        self.A = A
        return A

    def set_local(self, A_local: torch.Tensor):
        self.A = A_local

    # ---- communication helpers
    def _comm_tag(self, group) -> str:
        g = self.grid
        if group is g.row_group:
            return f"row{g.r}"
        if group is g.col_group:
            return f"col{g.c}"
        if group is g.diag_col_group:
            return f"diag{g.c}"
        return "world"

    def _stream_role(self) -> str:
        st = self._st
        if st is None or not st.on:
            return "host"
        cur = torch.cuda.current_stream()
        return "diag" if cur == st.diag else "side" if cur == st.side else "main" if cur == st.main else "caller"

    def _log(self, group, op: str, root: int, numel: int):
        if self.oplog is not None:
            self.oplog.append((self._comm_tag(group), op, int(root), int(numel), self._step_label, self._stream_role()))

    def _bcast(self, t: torch.Tensor, src_rank: int, group, members: List[int]) -> torch.Tensor:
        """Root ``src_rank`` -> every rank of ``members`` (``t`` allocated with the same shape everywhere).  Enqueued on
        the current stream with RCCL; blocking with gloo.  ONE message = ONE call = (p2p transport) ONE send/recv group: the
        root's group holds exactly the sends of this message to its peers, a peer's group exactly its one receive, so the
        k-th group a peer issues on a communicator always pairs with the k-th group of that communicator's root of the
        moment -- messages of different steps can never be matched with each other."""
        ct = _comm_tensor(t, self.backend)
        me = self.grid.rank
        _gloo_cuda_guard(ct, group)
        if self.transport == "p2p":
            self._log(group, "p2p_bcast", src_rank, ct.numel())
            if me == src_rank:
                reqs = [dist.P2POp(dist.isend, ct, peer, group) for peer in members if peer != src_rank]
            else:
                reqs = [dist.P2POp(dist.irecv, ct, src_rank, group)]
            for w in (dist.batch_isend_irecv(reqs) if reqs else []):
                w.wait()
        else:
            self._log(group, "broadcast", src_rank, ct.numel())
            dist.broadcast(ct, src=src_rank, group=group)
        if me != src_rank:
            self.bytes_received += ct.numel() * 8
        if ct.data_ptr() != t.data_ptr():
            t.copy_(ct)
        return t

    def _ring_shift(self, t: torch.Tensor, shift: int) -> torch.Tensor:
        """Inside the process row: send ``t`` to the rank ``shift`` process columns to the left, receive the tensor of the rank
        ``shift`` columns to the right: its local columns are the block columns THAT process column owns (their count is known
        from the layout, nothing is negotiated)."""
        g = self.grid
        dst, src = g.rank_of(g.r, (g.c - shift) % g.pc), g.rank_of(g.r, (g.c + shift) % g.pc)
        ct = _comm_tensor(t, self.backend)
        _gloo_cuda_guard(ct, g.row_group)
        ncols2 = sum(self.bs(J) for J in g.local_col_blocks(self.nblocks, (g.c + shift) % g.pc))
        buf = torch.empty((ct.shape[0], ncols2), dtype=ct.dtype, device=ct.device)
        self._log(g.row_group, f"ring_shift{shift}", -1, ct.shape[0])     # (rows: the same on every member; the column counts differ by <= 1)
        reqs = [dist.P2POp(dist.isend, ct, dst, g.row_group), dist.P2POp(dist.irecv, buf, src, g.row_group)]
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
        self.bytes_received += buf.numel() * 8
        return buf if buf.device == t.device else buf.to(t.device)

    def _world_bcast(self, t: torch.Tensor, src_rank: int):
        self._log(self.grid.world_group, "broadcast", src_rank, t.numel())
        dist.broadcast(t, src=src_rank, group=self.grid.world_group)
        return t

    def _all_reduce(self, t: torch.Tensor, op, group, what: str):
        self._log(group, f"all_reduce:{what}", -1, t.numel())
        dist.all_reduce(t, op=op, group=group)
        return t

    def _reduce(self, t: torch.Tensor, dst_rank: int, group, what: str):
        """Sum of ``t`` over ``group`` delivered to ``dst_rank`` (the other members' buffers are left unspecified)."""
        ct = _comm_tensor(t, self.backend)
        self._log(group, f"reduce:{what}", dst_rank, ct.numel())
        dist.reduce(ct, dst=dst_rank, op=dist.ReduceOp.SUM, group=group)
        if self.grid.rank == dst_rank and ct.data_ptr() != t.data_ptr():
            t.copy_(ct)
        return t

    # ---- factorisation
    @contextlib.contextmanager
    def _phase(self, name: str):
        """Bracket a phase with timing events on the current stream when profiling."""
        if not (self.profile and self._st is not None and self._st.on):
            yield
            return
        a = self._st.stamp()
        yield
        self._marks.append((name, a, self._st.stamp()))

    def phase_times(self):
        """Summed milliseconds per phase of the last factor() on this rank (synchronises)."""
        if not self._marks:
            return {}
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self._marks:
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        return out

    def _flat(self, size: int) -> torch.Tensor:
        return self.ops.empty(1, size).reshape(-1)

    def _prepare_panel(self, k: int, diag=None):
        """Steps 1-5 of block column k: diagonal factor (unless ``diag`` = (L_kk, dinv) was prepared ahead on the
        diagonal stream), panel solve, row broadcast, column exchange.
        Returns (panel, colop): the rows L_Ik of this process row (I > k) and the rows L_Jk for the owned block columns
        J > k.  Runs on the current stream; no host synchronisation."""
        g, ops, A = self.grid, self.ops, self.A
        rd, cd = g.owner_row(k), g.owner_col(k)
        bk = self.bs(k)
        in_col = g.c == cd
        self._step_label = f"panel{k}"
        i0, j0 = self._first_row_after(k), self._first_col_after(k)
        Mr = self.roff[-1] - self.roff[i0]
        Nc = self.coff[-1] - self.coff[j0]
        col_members = [g.rank_of(rr, g.c) for rr in range(g.pr)]
        row_members = [g.rank_of(g.r, cc) for cc in range(g.pc)]

        # 1-2. diagonal block: factor on the owner; (L_kk | inverses of its 128-blocks | info) travel down the
        #      process column in ONE message
        Lkk = dinv = None
        if in_col and diag is not None:
            Lkk, dinv = diag
        elif in_col:
            with self._phase("diag"):
                Lkk, dinv = self._diagonal_block(k, rd, cd, bk, col_members)

        # 3. panel solve on the owning process column, 4. broadcast along the process row
        self._step_label = f"panel{k}"
        panel = self._panel_buf[k % 2][:Mr, :bk]
        if in_col and Mr > 0:
            lj = k // g.pc
            P = A[self.roff[i0]:, self.coff[lj]:self.coff[lj + 1]]
            with self._phase("trsm"):
                if self._abi and self.panel_via_inverse:
                    ops.panel_solve_msg(Lkk, P, panel)
                elif self.panel_via_inverse and hasattr(ops, "trsm_right_via_inverse") and bk % 128 == 0:
                    ops.trsm_right_via_inverse(Lkk, dinv, P, panel)
                    P.copy_(panel)
                else:
                    ops.trsm_right(Lkk, dinv, P)
                    panel.copy_(P)
        if g.pc > 1 and Mr > 0:
            with self._phase("row_bcast"):
                self._bcast(panel, g.rank_of(g.r, cd), g.row_group, row_members)

        # 5. column operand: blocks J > k with J mod Pc == c, exchanged inside the process column (the holder of
        #    block row J is process row J mod Pr)
        colop = self._colop_buf[k % 2][:Nc, :bk]
        if Nc > 0:
            with self._phase("col_exchange"):
                self._column_exchange(k, panel, colop, i0, j0, bk, col_members)
        return panel, colop

    def _diagonal_ahead(self, k: int, panel, colop):
        """Critical path first: on the ranks of the process column that owns block column k+1, apply update k to the ONE
        diagonal block (k+1, k+1), factor it and send it down the column -- on the diagonal stream, while the side stream
        is still updating the rest of that block column.  Returns (L, dinv) of block k+1, or None on other ranks."""
        g, ops, A = self.grid, self.ops, self.A
        j = k + 1
        rd, cd = g.owner_row(j), g.owner_col(j)
        if g.c != cd:
            return None
        with self._phase("diag"):
            if g.r == rd:
                i0, j0 = self._first_row_after(k), self._first_col_after(k)
                li, lj = j // g.pr, j // g.pc
                D = A[self.roff[li]:self.roff[li + 1], self.coff[lj]:self.coff[lj + 1]]
                Ai = panel[self.roff[li] - self.roff[i0]: self.roff[li + 1] - self.roff[i0]]
                Bj = colop[self.coff[lj] - self.coff[j0]: self.coff[lj + 1] - self.coff[j0]]
                ops.gemm_nt_sub(D, Ai, Bj)
            return self._diagonal_block(j, rd, cd, self.bs(j), [g.rank_of(rr, g.c) for rr in range(g.pr)])

    def _diagonal_block(self, k, rd, cd, bk, col_members):
        g, ops, A = self.grid, self.ops, self.A
        self._step_label = f"diag{k}"
        ldk = (bk + 15) // 16 * 16
        ndinv = ((bk + 127) // 128) * 128 * 128
        dbuf = self._flat(bk * ldk + ndinv + 1)
        Lkk = dbuf[: bk * ldk].view(bk, ldk)[:, :bk]
        dinv = dbuf[bk * ldk: bk * ldk + ndinv]
        inf = dbuf[bk * ldk + ndinv:]
        if g.r == rd:
            li, lj = k // g.pr, k // g.pc
            D = A[self.roff[li]:self.roff[li + 1], self.coff[lj]:self.coff[lj + 1]]
            if self._abi:
                ops.diag_factor_msg(D, dbuf)
            else:
                dv, info = ops.potrf(D)
                dinv.copy_(dv[:ndinv])
                Lkk.copy_(D)
                inf.copy_(info.to(torch.float64))
        if g.pr > 1:
            # its own communicator (grid.diag_col_group): issued from the diagonal stream with look-ahead, from the side
            # stream without -- never interleaved with the column exchange's collectives on g.col_group
            self._bcast(dbuf, g.rank_of(rd, cd), g.diag_col_group, col_members)
        self._info_acc[k: k + 1].copy_(inf)
        self.diag_cache[k] = (Lkk, dinv)
        return Lkk, dinv

    def _exchange_maps(self, device):
        """Static index maps of the column exchange, built once per factorisation: for every process row rp, the local
        row index (in this rank's row space) and the local column-space index of every row of the blocks J with
        J mod Pc == c and J mod Pr == rp, ordered by J -- a step uses the suffix J > k."""
        g, nb = self.grid, self.nb
        maps = []
        for rp in range(g.pr):
            blocks = [J for J in self.col_blocks if J % g.pr == rp]
            src = [np.arange(self.roff[J // g.pr], self.roff[J // g.pr] + self.bs(J)) for J in blocks] if g.r == rp else []
            dst = [np.arange(self.coff[J // g.pc], self.coff[J // g.pc] + self.bs(J)) for J in blocks]
            cat = lambda parts: torch.as_tensor(np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64), dtype=torch.int64, device=device)  # noqa: E731
            maps.append((blocks, cat(src), cat(dst)))
        return maps

    def _column_exchange(self, k, panel, colop, i0, j0, bk, col_members):
        """colop rows of block J (J > k, J mod Pc == c) = panel rows of block J held by process row J mod Pr: one gather on
        the holder, one broadcast inside the process column, one scatter on the receivers per process row."""
        g, nb = self.grid, self.nb
        if self._abi:
            ops = self.ops
            for rp in range(g.pr):
                rows = ops.exchange_rows(self._lay, rp, k)
                if rows <= 0:
                    continue
                if g.pr == 1:          # the holder is this rank and the piece IS the column operand (consecutive blocks)
                    ops.exchange_pack(panel, colop, self._lay, k, bk)
                    continue
                piece = self._piece_buf[:rows, :bk]
                if g.r == rp:
                    ops.exchange_pack(panel, piece, self._lay, k, bk)
                self._bcast(piece, g.rank_of(rp, g.c), g.col_group, col_members)
                ops.exchange_unpack(piece, colop, self._lay, rp, k, bk)
            return
        for rp, (blocks, src_idx, dst_idx) in enumerate(self._xmaps):
            first = bisect.bisect_right(blocks, k)            # blocks[first:] are the J > k
            if first >= len(blocks):
                continue
            off = first * nb                                   # only the globally last block can be short
            dst = dst_idx[off:] - self.coff[j0]
            rows = int(dst.shape[0])
            if g.r == rp:
                src = src_idx[off:] - self.roff[i0]
                if g.pr == 1:
                    colop.index_copy_(0, dst, panel.index_select(0, src))
                    continue
                piece = self._piece_buf[:rows, :bk]
                torch.index_select(panel, 0, src, out=piece)
                colop.index_copy_(0, dst, piece)
            else:
                piece = self._piece_buf[:rows, :bk]
            self._bcast(piece, g.rank_of(rp, g.c), g.col_group, col_members)
            if g.r != rp:
                colop.index_copy_(0, dst, piece)

    def _update(self, k: int, panel, colop, jlo: int, jhi: int, rows_after: Optional[int] = None):
        """6. A_IJ -= L_Ik L_Jk^T for the local blocks I >= J with local column index in [jlo, jhi): a staircase of GEMMs
        over groups of up to 4 local block rows (the few blocks above the diagonal that a group also touches are never
        read afterwards)."""
        if jhi <= jlo:
            return
        ops, A = self.ops, self.A
        if self._abi:
            if A.shape[0] and A.shape[1]:
                ops.trailing_update(A, self._lay, k, panel, colop, jlo, jhi, rows_after)
            return
        i0, j0 = self._first_row_after(k), self._first_col_after(k)
        nrb = len(self.row_blocks)
        G = 4
        first = i0 if rows_after is None else self._first_row_after(rows_after)   # skip block rows <= rows_after
        for lg in range(first, nrb, G):
            le = min(lg + G, nrb)
            I_last = self.row_blocks[le - 1]
            jend = min(bisect.bisect_right(self.col_blocks, I_last), jhi)     # local columns J <= I_last
            if jend <= jlo:
                continue
            C = A[self.roff[lg]:self.roff[le], self.coff[jlo]:self.coff[jend]]
            Ai = panel[self.roff[lg] - self.roff[i0]: self.roff[le] - self.roff[i0]]
            Bj = colop[self.coff[jlo] - self.coff[j0]: self.coff[jend] - self.coff[j0]]
            ops.gemm_nt_sub(C, Ai, Bj)

    def factor(self):
        """See _factor.  While it runs with look-ahead, the panel chain works beside this rank's own bulk updates, so the
        library is told to use its small-footprint kernel for the chain's small products (gpmp_hint_machine_busy)."""
        lib = getattr(self.ops, "lib", None)
        prev = lib.gpmp_hint_machine_busy(1) if (lib is not None and self.lookahead) else None
        try:
            return self._factor()
        finally:
            if prev is not None:
                lib.gpmp_hint_machine_busy(prev)

    def _factor(self):
        """Right-looking factorisation with a one-step look-ahead.  While the caller's stream applies update k to the
        block columns > k+1, the side stream applies it to block column k+1, factors that column's diagonal block, solves
        its panel and runs every broadcast of step k+1.  Host code only enqueues; the one synchronisation is the
        final read of the ``info`` words."""
        g, ops, nb = self.grid, self.ops, self.nb
        nblk, ncb = self.nblocks, len(self.col_blocks)
        self._marks = []
        self._panel_buf = [ops.empty(self.local_rows(), nb) for _ in range(2)]
        self._colop_buf = [ops.empty(self.local_cols(), nb) for _ in range(2)]
        self._piece_buf = ops.empty(self.local_cols(), nb) if g.pr > 1 else None
        self._info_acc = self._flat(nblk)
        self._info_acc.zero_()
        self._xmaps = self._exchange_maps(self._info_acc.device)
        self.diag_cache = {}
        # (created after the buffers above were queued on the caller's stream: a masked update stream starts behind them)
        st = self._st = _Streams(getattr(ops, "device", None), self.reserve_cus if self.lookahead else 0, getattr(ops, "lib", None))
        ev_main_prev = None
        start = st.record(False)
        d0 = None
        if self.lookahead and g.c == g.owner_col(0):
            # the first diagonal block too goes out from the diagonal stream: grid.diag_col_group is then used from that
            # stream only, g.col_group / g.row_group from the side stream only
            with st.diag_ctx():
                st.wait_diag(start)
                with self._phase("diag"):
                    d0 = self._diagonal_block(0, g.owner_row(0), g.owner_col(0), self.bs(0), [g.rank_of(rr, g.c) for rr in range(g.pr)])
                ev_d0 = st.record_diag()
        with st.side_ctx():
            st.wait(True, start)
            if d0 is not None:
                st.wait(True, ev_d0)
            bufs = self._prepare_panel(0, diag=d0)
            ev_side = st.record(True)
        for k in range(nblk):
            panel, colop = bufs
            jrest = self._first_col_after(k)
            if k + 1 < nblk:
                if self.lookahead:
                    jnext = jrest
                    if g.c == g.owner_col(k + 1):
                        jnext = (k + 1) // g.pc          # local index of block column k + 1
                        jrest = jnext + 1
                    with st.diag_ctx():                   # critical path first: the next diagonal block
                        st.wait_diag(ev_main_prev)
                        st.wait_diag(ev_side)
                        dnext = self._diagonal_ahead(k, panel, colop)
                        ev_diag = st.record_diag()
                    with st.side_ctx():
                        st.wait(True, ev_main_prev)       # update k-1 has finished with column k+1 and with the buffers
                        with self._phase("lookahead_update"):
                            # (the diagonal block (k+1, k+1) has been updated on the diagonal stream)
                            self._update(k, panel, colop, jnext, jrest, rows_after=k + 1 if dnext is not None and g.r == g.owner_row(k + 1) else None)
                        st.wait(True, ev_diag)
                        nbufs = self._prepare_panel(k + 1, diag=dnext)
                        ev_side_next = st.record(True)
                    st.wait(False, ev_side)
                    with st.main_ctx(), self._phase("update"):
                        self._update(k, panel, colop, jrest, ncb)
                    ev_main_prev = st.record(False)
                else:
                    st.wait(False, ev_side)
                    with st.main_ctx(), self._phase("update"):
                        self._update(k, panel, colop, jrest, ncb)
                    ev_main_prev = st.record(False)
                    with st.side_ctx():
                        st.wait(True, ev_main_prev)
                        nbufs = self._prepare_panel(k + 1)
                        ev_side_next = st.record(True)
                bufs, ev_side = nbufs, ev_side_next
        st.wait(False, ev_side)
        st.close()
        self._panel_buf = self._colop_buf = self._piece_buf = None
        # agree on info: the first failing block column wins
        mine = math.inf
        for k, v in enumerate(self._info_acc.cpu().tolist()):
            if v != 0:
                mine = k * nb + int(v)
                break
        it = torch.tensor([mine], dtype=torch.float64)
        it = it.to("cuda") if self.backend == "nccl" else it
        self._step_label = "info"
        self._all_reduce(it, dist.ReduceOp.MIN, g.world_group, "info")
        self.info = 0 if math.isinf(float(it.item())) else int(it.item())
        return self.info

    # ---- scalars
    def logdet(self) -> float:
        """2 sum_i log L_ii, all-reduced."""
        g, ops = self.grid, self.ops
        s = 0.0
        for k in range(self.nblocks):
            if g.r == g.owner_row(k) and g.c == g.owner_col(k):
                s += ops.sum_log_diag(self.diag_cache[k][0])
        t = torch.tensor([2.0 * s], dtype=torch.float64)
        t = t.to("cuda") if self.backend == "nccl" else t
        self._step_label = "logdet"
        self._all_reduce(t, dist.ReduceOp.SUM, g.world_group, "logdet")
        return float(t.item())
