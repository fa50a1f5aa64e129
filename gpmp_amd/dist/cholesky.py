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


# ------------------------------------------------------------------------------------------------
# local compute back-ends
# ------------------------------------------------------------------------------------------------
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


# ------------------------------------------------------------------------------------------------
def _comm_tensor(t: torch.Tensor, backend: str) -> torch.Tensor:
    """Contiguous tensor on the device the communication backend wants."""
    if backend == "nccl":
        return t.contiguous()
    return t.detach().to("cpu").contiguous()


def _gloo_cuda_guard(t: torch.Tensor, group) -> None:
    """gloo moves CUDA tensors, but its point-to-point send does not wait for the kernels that produce the tensor
    (tools/gloo_cuda_p2p_probe.py: the receiver gets stale data); RCCL's operations are stream-ordered.  The device-resident
    communication path is only ever combined with gloo by tests/test_dist_gpu.py's probe of that path on a shared GPU: there,
    the producing stream is drained first.  No effect under RCCL or with CPU tensors."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream(t.device).synchronize()


class _Streams:
    """The stream of the bulk trailing updates ("main") and one high-priority side stream (the panel chain and every
    collective).  With ``reserve_cus`` > 0 the bulk updates run on a CU-masked stream of their own, fenced against the
    caller's stream at both ends, so that the small kernels of the panel chain always find a free CU.  With CPU local
    ops (tests) everything degenerates to program order."""

    def __init__(self, device, reserve_cus: int = 0, lib=None):
        self.on = device is not None and torch.device(device).type == "cuda"
        self.caller = None
        self._masked = None
        if self.on:
            self.caller = self.main = torch.cuda.current_stream(device)
            self.side = torch.cuda.Stream(device=device, priority=-1)
            self.diag = torch.cuda.Stream(device=device, priority=-1)
            if reserve_cus > 0 and lib is not None:
                import ctypes

                h = ctypes.c_void_p()
                rc = lib.gpmp_stream_create_reserving_cus(int(reserve_cus), ctypes.byref(h))
                if rc != 0:
                    raise RuntimeError(f"gpmp_stream_create_reserving_cus failed ({rc})")
                self._masked, self._lib = h, lib
                self.main = torch.cuda.ExternalStream(h.value, device=device)
                self.main.wait_stream(self.caller)

    def main_ctx(self):
        return torch.cuda.stream(self.main) if self.on else contextlib.nullcontext()

    def close(self):
        """Join the masked stream into the caller's stream and release it; hand back what the library keeps for the two
        side streams (they are created per factorisation: the flag block of the one-launch solve must not pile up)."""
        lib = self._lib if self._masked is not None else None
        if self.on:
            try:
                from .. import _lib as _l

                lib = _l.load()
                for s in (self.side, self.diag):
                    s.synchronize()
                    lib.gpmp_stream_release(s.cuda_stream)
            except ImportError:
                pass
        if self._masked is not None:
            self.caller.wait_stream(self.main)
            self.main.synchronize()          # the stream object goes away: nothing of ours may still be queued on it
            self._lib.gpmp_stream_destroy(self._masked)
            self._masked = None
            self.main = self.caller

    def diag_ctx(self):
        return torch.cuda.stream(self.diag) if self.on else contextlib.nullcontext()

    def wait_diag(self, ev):
        if self.on and ev is not None:
            self.diag.wait_event(ev)

    def record_diag(self):
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record(self.diag)
        return ev

    def side_ctx(self):
        return torch.cuda.stream(self.side) if self.on else contextlib.nullcontext()

    def record(self, side: bool):
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record(self.side if side else self.main)
        return ev

    def wait(self, side: bool, ev):
        if self.on and ev is not None:
            (self.side if side else self.main).wait_event(ev)

    def stamp(self):
        """Timing event on the CURRENT stream (None off-GPU)."""
        if not self.on:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev


class BlockCyclicCholesky:
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

