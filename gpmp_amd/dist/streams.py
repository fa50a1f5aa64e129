"""Streams and communication-tensor helpers of the block-cyclic layer: the bulk / panel-chain / diagonal streams of a schedule and
the placement of a message for the communication backend (device-resident under RCCL, host-staged under gloo)."""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist


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
