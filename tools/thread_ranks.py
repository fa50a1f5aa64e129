#!/usr/bin/env python3
"""Ranks as THREADS of one process -- test infrastructure for the distributed layer on a one-GPU box (the pool's process guard
allows six processes on the card; the 2 x 4 grid of BASELINE config 5 has eight ranks).

Every thread-rank has its own ``HipLocalOps``, streams, buffers and ``ProcessGrid`` (built on the in-process "threaded" process
group of ``torch.testing``: its groups serve as communicator identities).  Two transports, both subclasses of
``gpmp_amd.dist.BlockCyclicCholesky`` that replace its five communication helpers and nothing else:

``HostCopyCholesky``
    the host-staged branch of the product code (what runs over gloo): messages go device -> host, the in-process group copies them
    between the threads, host -> device.  The group has no reduce and no point-to-point operation: a reduce is an all-reduce whose
    result the destination keeps, the gradient's ring shift a round of row broadcasts.

``StreamOrderedCholesky``
    the DEVICE-RESIDENT branch (what runs under RCCL: ``backend == "nccl"``) with an in-process fabric that has RCCL's stream
    semantics: a collective is enqueued on each member's CURRENT stream and ordered by HIP events only -- receivers' streams wait
    for the sender's "ready" event and copy device to device, the sender's stream waits for the receivers' "done" events before it
    goes on; the host threads only meet to hand each other tensors and events (they never wait for GPU work).  gloo's host
    synchronisations hide a missing stream dependency of the schedule; this fabric does not: with a dependency missing, a buffer is
    read or overwritten early and the values come out wrong.
"""
import threading
import traceback

import torch
import torch.distributed as dist


class Fabric:
    """Meeting point of the thread-ranks: the members of a communicator meet once per phase of a collective, in issue order (every
    member issues the same sequence on a communicator: tests/test_dist_cpu.py, the issue-order test)."""

    def __init__(self, world, timeout=600.0):
        self.world, self.timeout = world, timeout
        self.cv = threading.Condition()
        self.meet, self.left, self.seq = {}, {}, {}
        self.failed = None

    def next_seq(self, rank, tag):
        k = (rank, tag)
        self.seq[k] = self.seq.get(k, 0) + 1
        return self.seq[k]

    def exchange(self, key, rank, nmem, payload):
        """deposit ``payload`` under ``key``, wait until all ``nmem`` members have, return {rank: payload}"""
        with self.cv:
            self.meet.setdefault(key, {})[rank] = payload
            self.cv.notify_all()
            ok = self.cv.wait_for(lambda: self.failed is not None or len(self.meet[key]) == nmem, self.timeout)
            if self.failed is not None:
                raise RuntimeError("another thread-rank failed: " + self.failed)
            if not ok:
                self.failed = f"collective {key} timed out on rank {rank} ({len(self.meet[key])} of {nmem} members arrived)"
                self.cv.notify_all()
                raise RuntimeError(self.failed)
            out = dict(self.meet[key])
            n_left = self.left.get(key, 0) + 1
            if n_left == nmem:
                del self.meet[key]
                self.left.pop(key, None)
            else:
                self.left[key] = n_left
            return out

    def fail(self, why):
        with self.cv:
            if self.failed is None:
                self.failed = why
            self.cv.notify_all()

    def barrier(self, rank):
        self.exchange(("barrier", self.next_seq(rank, "barrier")), rank, self.world, None)

    def allgather(self, rank, obj):
        got = self.exchange(("gather", self.next_seq(rank, "gather")), rank, self.world, obj)
        return [got[r] for r in range(self.world)]


class _Stream:
    """the calling thread's current stream on the tensor's device; without a GPU (CPU stand-in of the tests) program order"""

    def __init__(self, t):
        self.s = torch.cuda.current_stream(t.device) if t.is_cuda else None

    def mark(self):
        if self.s is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.s)
        return ev

    def wait_event(self, ev):
        if self.s is not None and ev is not None:
            self.s.wait_event(ev)


def _members_of(ch, group):
    g = ch.grid
    if group is g.row_group:
        return [g.rank_of(g.r, cc) for cc in range(g.pc)]
    if group is g.col_group or group is g.diag_col_group:
        return [g.rank_of(rr, g.c) for rr in range(g.pr)]
    return list(range(g.world))


def make_classes(fabric):
    from gpmp_amd.dist import BlockCyclicCholesky

    class HostCopyCholesky(BlockCyclicCholesky):
        def _reduce(self, t, dst_rank, group, what):
            self._log(group, f"reduce:{what}", dst_rank, t.numel())
            ct = t.detach().to("cpu").contiguous()
            dist.all_reduce(ct, op=dist.ReduceOp.SUM, group=group)
            if self.grid.rank == dst_rank:
                t.copy_(ct)
            return t

        def _ring_shift(self, t, shift):
            g = self.grid
            src_c = (g.c + shift) % g.pc
            self._log(g.row_group, f"ring_shift{shift}", -1, t.shape[0])
            mine = t.detach().to("cpu").contiguous()
            keep = None
            for cc in range(g.pc):
                w = sum(self.bs(J) for J in g.local_col_blocks(self.nblocks, cc))
                buf = mine if cc == g.c else torch.empty((t.shape[0], w), dtype=t.dtype)
                dist.broadcast(buf, src=g.rank_of(g.r, cc), group=g.row_group)
                if cc == src_c:
                    keep = buf
                    self.bytes_received += buf.numel() * 8
            return keep.to(t.device)

    class StreamOrderedCholesky(BlockCyclicCholesky):
        """device-resident messages, ordered by events on the members' current streams (module docstring)"""

        # adversarial latency (tests): every incoming message is held back on the receiving stream by a pseudo-random number of
        # GPU clock cycles up to this bound before it is copied -- a consumer that does not wait for the message's event reads stale data
        max_delay_cycles = 0

        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self.backend = "nccl"            # the branch of the product code that keeps everything on the device
            self._lcg = 12345 + 7919 * self.grid.rank

        def _hold_back(self, t):
            if self.max_delay_cycles > 0 and t.is_cuda:
                self._lcg = (1103515245 * self._lcg + 12345) % (1 << 31)
                torch.cuda._sleep(int(self._lcg % self.max_delay_cycles))

        # -- one collective = two meetings: (a) tensors + "ready" events, (b) "done" events
        def _meet(self, group, payload_fn):
            me = self.grid.rank
            tag = self._comm_tag(group)
            members = _members_of(self, group)
            seq = fabric.next_seq(me, tag)
            cur = _Stream(payload_fn())
            got = fabric.exchange((tag, seq, "a"), me, len(members), (payload_fn(), cur.mark()))
            return me, tag, members, seq, cur, got

        def _finish(self, tag, seq, me, members, cur, wait_for):
            got = fabric.exchange((tag, seq, "b"), me, len(members), cur.mark())
            for r in wait_for:
                if r != me:
                    cur.wait_event(got[r])

        def _bcast(self, t, src_rank, group, members_arg):
            self._log(group, "broadcast", src_rank, t.numel())
            me, tag, members, seq, cur, got = self._meet(group, lambda: t)
            if me != src_rank:
                cur.wait_event(got[src_rank][1])
                self._hold_back(t)
                t.copy_(got[src_rank][0])
                self.bytes_received += t.numel() * 8
            # the root goes on (and may overwrite its buffer) only when every receiver has copied
            self._finish(tag, seq, me, members, cur, members if me == src_rank else [])
            return t

        def _world_bcast(self, t, src_rank):
            return self._bcast(t, src_rank, self.grid.world_group, None)

        def _combine(self, t, op, group, what, dst):
            me, tag, members, seq, cur, got = self._meet(group, lambda: t)
            tmp = None
            if dst is None or me == dst:
                for r in members:
                    cur.wait_event(got[r][1])
                self._hold_back(t)
                parts = torch.stack([got[r][0] for r in members])
                tmp = parts.sum(dim=0) if op == dist.ReduceOp.SUM else (parts.amin(dim=0) if op == dist.ReduceOp.MIN else parts.amax(dim=0))
            # every member's input has been read by the time the writers' "done" events have fired
            writers = members if dst is None else [dst]
            self._finish(tag, seq, me, members, cur, writers)
            if tmp is not None:
                t.copy_(tmp)
            return t

        def _all_reduce(self, t, op, group, what):
            self._log(group, f"all_reduce:{what}", -1, t.numel())
            return self._combine(t, op, group, what, None)

        def _reduce(self, t, dst_rank, group, what):
            self._log(group, f"reduce:{what}", dst_rank, t.numel())
            return self._combine(t, dist.ReduceOp.SUM, group, what, dst_rank)

        def _ring_shift(self, t, shift):
            g = self.grid
            dst, src = g.rank_of(g.r, (g.c - shift) % g.pc), g.rank_of(g.r, (g.c + shift) % g.pc)
            self._log(g.row_group, f"ring_shift{shift}", -1, t.shape[0])
            me, tag, members, seq, cur, got = self._meet(g.row_group, lambda: t)
            cur.wait_event(got[src][1])
            buf = torch.empty_like(got[src][0])
            self._hold_back(t)
            buf.copy_(got[src][0])
            self.bytes_received += buf.numel() * 8
            self._finish(tag, seq, me, members, cur, [dst])        # my tensor has been read by the rank I send to
            return buf

    return HostCopyCholesky, StreamOrderedCholesky


def run(world, body, limit_s=900.0):
    """Start ``world`` thread-ranks on the in-process "threaded" process group and run ``body(rank, world, fabric, classes)`` in each;
    -> list of error texts (empty: every rank returned).  A watchdog ends the PROCESS if ranks are still running after ``limit_s``."""
    import os

    from torch.testing._internal.distributed import multi_threaded_pg as mtpg

    mtpg._install_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(True)      # (group registry per thread, as torch's MultiThreadedTestCase does)
    store = dist.HashStore()
    fabric = Fabric(world, timeout=limit_s)
    classes = make_classes(fabric)
    errors = []

    def rank_main(rank):
        try:
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            dist.init_process_group(backend="threaded", rank=rank, world_size=world, store=store)
            body(rank, world, fabric, classes)
            fabric.barrier(rank)
            dist.destroy_process_group()
        except BaseException:  # noqa: BLE001 -- a failing rank must release the others from their meetings
            errors.append(f"rank {rank}:\n{traceback.format_exc()}")
            fabric.fail(f"rank {rank} raised")
            mtpg.ProcessLocalGroup.exception_handle(None)

    def watchdog():
        print(f"[thread_ranks] still running after {limit_s:.0f} s: ending the process", flush=True)
        os._exit(3)

    timer = threading.Timer(limit_s, watchdog)
    timer.daemon = True
    timer.start()
    threads = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(world)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        timer.cancel()
        torch._C._distributed_c10d._set_thread_isolation_mode(False)
        mtpg._uninstall_threaded_pg()
    return errors
