"""Stand-in workload for bench.py's process handling -- TEST INFRASTRUCTURE (selected with GPMP_BENCH_STUB_MODULE).

bench.py's launcher, rank bookkeeping, barrier / max-over-ranks timing, the isolated process group of the distributed
extra and its kill-on-timeout path need no GPU to be tested: this module supplies a workload whose "step" sleeps a few
milliseconds and a distributed extra that can succeed, fail on one rank or hang inside a collective
(GPMP_STUB_DIST = ok | error | hang; GPMP_STUB_HEADLINE_FAIL_RANK = r makes rank r of the headline exit with code 7).
Nothing here computes anything; its line says "data": "stub"."""
import os
import sys
import time


class Workload:
    data = "stub"

    def __init__(self, args, rank, world):
        self.args, self.rank, self.world, self.steps_done = args, rank, world, 0

    def device(self):
        return "cpu"

    def sync(self):
        pass

    def step(self, m=None):
        if os.environ.get("GPMP_STUB_HEADLINE_FAIL_RANK") == str(self.rank):
            sys.stderr.write(f"[stub] rank {self.rank} fails on purpose\n")
            os._exit(7)
        time.sleep(0.005 * (1 + self.rank))        # ranks differ: the line must carry the MAX over ranks
        self.steps_done += 1

    def timed_begin(self):
        self.t_begin = self.steps_done

    def timed_end(self):
        self.timed_steps = self.steps_done - self.t_begin

    def check(self):
        assert self.timed_steps == self.args.steps

    def report(self, line):
        line["roofline"] = None
        line["extra"]["stub_timed_steps"] = self.timed_steps

    def release(self):
        pass


class _StubBlockCyclicRunner:
    """what bench.time_block_cyclic_strong_scaling drives: step(shared_factor), sync(), check(), .m"""

    m = 50000

    def __init__(self, rank):
        self.rank, self.calls = rank, []

    def sync(self):
        pass

    def step(self, shared_factor):
        self.calls.append(bool(shared_factor))
        time.sleep(0.004 if shared_factor else 0.008)

    def check(self):
        return {"ok": True, "calls": len(self.calls)} if self.rank == 0 else None


def dist_extra(world, rank, res):
    import torch
    import torch.distributed as dist

    import bench

    mode = os.environ.get("GPMP_STUB_DIST", "ok")
    res["phase"] = "stub: all-reduce"
    t = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(t)
    assert int(t.item()) == world
    res["pids_sum_check"] = int(t.item())
    def tmax(v):
        t_ = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    res["phase"] = "stub: strong_scaling_block_cyclic"
    res["strong_scaling_block_cyclic"] = bench.time_block_cyclic_strong_scaling(_StubBlockCyclicRunner(rank), dist, world, rank, 2, tmax)
    if mode == "error" and rank == world - 1:
        raise RuntimeError("stub failure on the last rank")
    if mode == "hang":
        res["phase"] = "stub: hanging in a barrier"
        if rank == world - 1:
            time.sleep(3600)          # never reaches the barrier: every other rank waits inside the collective
    dist.barrier()
    res["phase"] = "done"
