#!/usr/bin/env python3
"""One rank's share of the block-cyclic Cholesky of BASELINE config 5 (n = 131072 on a 2 x 4 grid) on ONE GPU.

The collectives are replaced by no-ops, so a rank outside the owning process column works on whatever its receive
buffers hold: the numbers are meaningless, but every local kernel runs with exactly the shapes, leading dimensions and
offsets it has in the 8-GPU run (local matrix 65536 x 32768 = 17 GB).  Purpose: (1) a fault check of the local kernels
at shapes the single-GPU path never sees (> 4 GB operands, 64 K-row panels), (2) the compute-only time of a rank, i.e.
the lower bound that communication is overlapped against.

    python tools/dist_rank_emulation.py --size-n 131072 --grid 2x4 --coords 0,0
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size-n", dest="n", type=int, default=131072)
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--grid", default="2x4")
    ap.add_argument("--coords", default="0,0")
    ap.add_argument("--no-lookahead", action="store_true")
    ap.add_argument("--reserve-cus", type=int, default=None)
    ap.add_argument("--solve-m", type=int, default=0, help="also time the many-right-hand-side solve for this many prediction points "
                    "(split over the process columns), with and without the prefetch / chain / update overlap")
    ap.add_argument("--grad", action="store_true", help="also time one ML value + gradient on the factor (T = L^-1 in the block-cyclic "
                    "layout, the blocks of T^T T around the process row, the fused cross traces), collectives stubbed")
    ap.add_argument("--step-m", type=int, default=0, help="also time one emulated HEADLINE STEP on the block-cyclic factor: (Gram + factor + "
                    "zero-mean prediction at this many points) + (Gram + factor + NLL), what extra.strong_scaling_block_cyclic of bench.py "
                    "runs on the real grid; compute only (collectives stubbed)")
    ap.add_argument("--link-gbps", type=float, default=0.0, help="model the TIME of every message: the stream that issues a broadcast / "
                    "exchange / ring shift is held for bytes / (this many GB/s) + --link-latency-us (a spin kernel; both for the root and "
                    "for a receiver: a broadcast occupies both ends).  0 = messages take no time (compute-only share)")
    ap.add_argument("--link-latency-us", type=float, default=10.0)
    ap.add_argument("--lambdas", action="store_true", help="with --solve-m: also time the BACKWARD many-right-hand-side solve (the kriging "
                    "weights of return_lambdas=True), with and without the prefetch / bulk / chain overlap")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("gloo", rank=0, world_size=1)
    import gpmp_amd.num as gnp
    from gpmp_amd.dist import BlockCyclicCholesky, HipLocalOps, ProcessGrid
    from gpmp_amd.kernel import MaternCovariance

    pr, pc = (int(v) for v in a.grid.split("x"))
    r, c = (int(v) for v in a.coords.split(","))
    grid = ProcessGrid.__new__(ProcessGrid)
    grid.world, grid.rank, grid.pr, grid.pc, grid.r, grid.c = pr * pc, r * pc + c, pr, pc, r, c
    grid.row_groups, grid.col_groups, grid.diag_col_groups, grid.world_group = [None] * pr, [None] * pc, [None] * pc, None

    # spin-kernel calibration for the link model: cycles of torch.cuda._sleep per microsecond
    cyc_per_us = 0.0
    if a.link_gbps > 0:
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
        cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)
    held = {"us": 0.0, "messages": 0}

    def hold(nbytes):
        """the issuing stream is busy with this message for latency + bytes / bandwidth"""
        if a.link_gbps > 0:
            us = a.link_latency_us + nbytes / (a.link_gbps * 1e3)
            held["us"] += us
            held["messages"] += 1
            torch.cuda._sleep(int(us * cyc_per_us))

    class Emulated(BlockCyclicCholesky):
        def _bcast(self, t, src_rank, group, members):      # noqa: D401 -- no communication: shapes (and, with --link-gbps, time) only
            if self.grid.rank != src_rank:
                self.bytes_received += t.numel() * 8
            hold(t.numel() * 8)
            return t

        def _ring_shift(self, t, shift):                     # the neighbour's part of T has this rank's shape here
            self.bytes_received += t.numel() * 8
            hold(t.numel() * 8)
            return t

        def _all_reduce(self, t, op, group, what):          # (sub-communicators do not exist in the emulation)
            return t

        def _world_bcast(self, t, src_rank):
            return t

        def _reduce(self, t, dst_rank, group, what):          # (the backward solve's partial sums stay where they are)
            return t

    n, d = a.n, 8
    rng = np.random.default_rng(1234)
    x = gnp.asarray(rng.random((n, d)))
    theta = np.concatenate(([0.0], -np.log(0.5 * (1.0 + np.arange(d) / d))))
    ch = Emulated(grid, n, nb=a.block, ops=HipLocalOps(), lookahead=not a.no_lookahead, profile=True, reserve_cus=a.reserve_cus)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ch.build_local_gram(MaternCovariance(2), x, theta, 1e-4)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    # factor() ends with a world all-reduce of info: the 1-rank gloo group serves it
    ch.factor()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    factor_phases = {k: round(v, 2) for k, v in ch.phase_times().items()}
    local_shape, reserve_cus, bytes_received = [ch.local_rows(), ch.local_cols()], ch.reserve_cus, ch.bytes_received
    share = (n ** 3 / 3.0) / (pr * pc)
    solve = None
    if a.solve_m > 0:
        from gpmp_amd.dist import shard_bounds

        j0, j1 = shard_bounds(a.solve_m, pc, c)
        mc = j1 - j0
        solve = {"m": a.solve_m, "m_local": mc, "rank_share_flops": float(n) * n * a.solve_m / (pr * pc)}
        for overlap in (True, False, True):
            B = gnp.alloc_matrix(ch.local_rows(), mc)
            B.copy_(torch.randn(ch.local_rows(), mc, dtype=torch.float64, device=B.device) * 1e-3)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            ch.solve_lower_many(B, overlap=overlap, profile=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t3
            solve["overlap" if overlap else "in_order"] = {"s": dt, "rank_share_tflops": solve["rank_share_flops"] / dt / 1e12,
                                                           "phases_ms": {k: round(v, 2) for k, v in ch.phase_times().items()}}
            del B
        if a.lambdas:
            solve["backward"] = {}
            for overlap in (True, False, True):
                B = gnp.alloc_matrix(ch.local_rows(), mc)
                B.copy_(torch.randn(ch.local_rows(), mc, dtype=torch.float64, device=B.device) * 1e-3)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                ch.solve_upper_many(B, overlap=overlap, profile=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t3
                solve["backward"]["overlap" if overlap else "in_order"] = {"s": dt, "rank_share_tflops": solve["rank_share_flops"] / dt / 1e12,
                                                                           "phases_ms": {k: round(v, 2) for k, v in ch.phase_times().items()}}
                del B
    grad = None
    if a.grad:
        xh = np.asarray(rng.random((n, d)))
        zh = np.sin(2 * np.pi * xh[:, 0]) + xh[:, 1:].sum(axis=1)
        grad = {}
        ch.info = 0          # (the stubbed factorisation "fails": the values are meaningless, the kernels and their shapes are not)
        ch.backend = "nccl"  # few-column solves and scalars stay on the device, as under RCCL (nothing is sent: stubs); with "gloo"
                             # every block of the few-column solve crosses PCIe twice, which an RCCL run never does
        for rep in ("warm", "timed"):
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            T = ch.inverse_factor_local()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            del T
            ch.value_and_grad(xh, zh, theta, 2)
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            grad[rep] = {"inverse_factor_s": t4 - t3, "inverse_factor_rank_share_tflops": (n ** 3 / 3.0) / (pr * pc) / (t4 - t3) / 1e12,
                         "value_and_grad_s": t5 - t4, "note": "value_and_grad includes its own inverse factor; ring shifts stubbed"}
    step = None
    if a.step_m > 0:
        from gpmp_amd.dist import shard_bounds

        xh = rng.random((n, d))
        zh = np.sin(2 * np.pi * xh[:, 0]) + xh[:, 1:].sum(axis=1)
        xt = np.random.default_rng(4321).random((a.step_m, d))
        del ch
        torch.cuda.empty_cache()
        step = {"m_total": a.step_m, "m_local": shard_bounds(a.step_m, pc, c)[1] - shard_bounds(a.step_m, pc, c)[0]}
        for rep in ("warm", "timed"):
            parts = {}
            torch.cuda.synchronize()
            t_begin = time.perf_counter()
            for what in ("predict", "nll"):
                t3 = time.perf_counter()
                c2 = Emulated(grid, n, nb=a.block, ops=HipLocalOps(), lookahead=not a.no_lookahead)
                c2.backend = "nccl"          # scalars and the few-column solve stay on the device, as under RCCL (nothing is sent: stubs)
                c2.build_local_gram(MaternCovariance(2), gnp.asarray(xh), theta, 1e-4)
                c2.factor()
                c2.info = 0
                torch.cuda.synchronize()
                t4 = time.perf_counter()
                if what == "predict":
                    c2.predict_zero_mean(MaternCovariance(2), xh, zh, xt, theta)
                else:
                    c2.negative_log_likelihood(zh)
                torch.cuda.synchronize()
                t5 = time.perf_counter()
                parts[what] = {"gram_and_factor_s": t4 - t3, "rest_s": t5 - t4}
                del c2
            step[rep] = {"step_s": time.perf_counter() - t_begin, "parts": parts}
        ch = None
    print(json.dumps({"tool": "dist_rank_emulation", "n": n, "grid": a.grid, "coords": a.coords, "block": a.block,
                      "local_shape": local_shape, "lookahead": not a.no_lookahead, "reserve_cus": reserve_cus,
                      "gram_s": t1 - t0, "factor_s": t2 - t1, "rank_share_tflops": share / (t2 - t1) / 1e12,
                      "frac_of_fp64_mfma_peak": share / (t2 - t1) / 1e12 / 78.6,
                      "bytes_received_GB": bytes_received / 1e9,
                      "link_model": None if a.link_gbps <= 0 else {"GBps_per_message": a.link_gbps, "latency_us": a.link_latency_us,
                                                                   "messages": held["messages"], "stream_seconds_held": held["us"] / 1e6,
                                                                   "note": "every broadcast / exchange / ring shift holds its issuing stream for latency + bytes / bandwidth (all phases of this run together)"},
                      "phases_ms": factor_phases, "solve": solve, "grad": grad, "headline_step": step,
                      "note": "collectives stubbed: timing and fault check only, values are not a factorisation"}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
