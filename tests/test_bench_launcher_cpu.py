"""bench.py's process handling at N > 1, without a GPU (stub workload over gloo, tests/bench_stub.py):

* `python bench.py --gpus 2` starts two worker processes itself (GPU-free launcher) and relays rank 0's line: n_gpus == 2,
  two distinct worker PIDs, neither of them the launcher's;
* the distributed extra runs by default in its OWN two fresh processes, and a failure or a hang there is confined to it:
  the headline value is still printed, `extra.dist_potrf.status` / `phase` say what happened, the exit code is 4 / 3 and
  no process of the killed group survives;
* a failing headline worker's exit code propagates; a --gpus / WORLD_SIZE mismatch is an error, not a one-rank run;
* the same under `python -m torch.distributed.run` (the driver's multi-GPU command): rank 0 coordinates the extra."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GPMP_BENCH_STUB_MODULE="tests.bench_stub", GPMP_BENCH_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    env.update(kw)
    return env


def _run(args, timeout=180, **envkw):
    p = subprocess.run([sys.executable, BENCH] + args, env=_env(**envkw), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    last = json.loads(lines[-1]) if lines and lines[-1].lstrip().startswith("{") else None
    return p, last


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    # a zombie still answers kill(pid, 0): look at its state
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def test_launcher_starts_n_ranks_and_relays_rank0_line():
    p, line = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["data"] == "stub" and line["steps"] == 3
    pids = line["extra"]["worker_pids"]
    assert len(pids) == 2 and len(set(pids)) == 2
    la = line["extra"]["launcher"]
    assert sorted(la["worker_pids"]) == sorted(pids) and la["pid"] not in pids
    # value = N * m / (max over ranks of the step time): rank 1 sleeps 10 ms per step, rank 0 5 ms
    assert line["ms_per_step"] >= 10.0
    assert abs(line["value"] - 2 * 50000 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    ss = line["extra"]["strong_scaling"]
    assert ss["m_total"] == 50000 and ss["m_per_gpu"] == [25000, 25000] and ss["points_per_s"] > 0
    assert "EVERY rank" in line["config"]["parallelism"]
    # the distributed extra ran by default, in two OTHER processes
    dp = line["extra"]["dist_potrf"]
    assert dp["status"] == "ok" and dp["phase"] == "done" and dp["worker_rc"] == [0, 0]
    assert len(set(dp["worker_pids"]) | set(pids)) == 4
    # round 4: the record proves its own membership -- one gathered (rank, pid, host, device) entry per rank, in the headline
    # line and in the extra's own process group; and the block-cyclic strong-scaling entry of the headline workload
    for seen, who in ((line["ranks_seen"], pids), (dp["ranks_seen"], dp["worker_pids"])):
        assert seen["world"] == 2 and seen["backend"] == "gloo" and seen["distinct_pids"] == 2
        assert [r["rank"] for r in seen["ranks"]] == [0, 1] and sorted(r["pid"] for r in seen["ranks"]) == sorted(who)
        assert all(set(r) >= {"host", "local_rank", "device_index", "uuid", "pci_bus_id"} for r in seen["ranks"])
    sb = dp["strong_scaling_block_cyclic"]
    for key, floor_ms in (("two_factorisations", 8.0), ("shared_factor", 4.0)):
        assert sb[key]["steps"] == 2 and sb[key]["ms_per_step"] >= floor_ms
        assert abs(sb[key]["points_per_s"] - 50000 / (sb[key]["ms_per_step"] * 1e-3)) < 1e-6 * sb[key]["points_per_s"]
    assert sb["values_check"] == {"ok": True, "calls": 6}          # (1 warm-up + 2 timed) x 2 variants on rank 0


def test_hung_collective_in_the_extra_is_killed_and_costs_only_the_extra():
    t0 = time.monotonic()
    p, line = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"], GPMP_STUB_DIST="hang", GPMP_BENCH_DIST_TIMEOUT="6")
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert time.monotonic() - t0 < 120
    assert line["n_gpus"] == 2 and line["value"] > 0               # the headline line survived
    dp = line["extra"]["dist_potrf"]
    assert dp["status"] == "timeout" and dp["phase"] == "stub: hanging in a barrier"
    time.sleep(0.5)
    assert not any(_alive(pid) for pid in dp["worker_pids"])       # the whole group is gone, the sleeper included


def test_error_in_the_extra_sets_exit_code_4_and_keeps_the_headline():
    p, line = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"], GPMP_STUB_DIST="error", GPMP_BENCH_DIST_TIMEOUT="60")
    assert p.returncode == 4, (p.returncode, p.stderr[-2000:])
    assert line["value"] > 0
    dp = line["extra"]["dist_potrf"]
    assert dp["status"] == "error" and 4 in dp["worker_rc"]
    assert not any(_alive(pid) for pid in dp["worker_pids"])


def test_extra_can_be_switched_off():
    p, line = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], GPMP_BENCH_DIST="0")
    assert p.returncode == 0 and "dist_potrf" not in line["extra"]


def test_failing_headline_worker_propagates_its_exit_code():
    p, line = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"], GPMP_STUB_HEADLINE_FAIL_RANK="1")
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert line["value"] is None and "error" in line


def test_world_size_mismatch_is_an_error_not_a_one_rank_run():
    # a worker that finds WORLD_SIZE != --gpus (round 2: it ran as one rank and printed "n_gpus": 1)
    p, line = _run(["--gpus", "2", "--role", "headline"])
    assert p.returncode == 2 and line is None and "WORLD_SIZE=1" in p.stderr
    p, line = _run(["--gpus", "2"], RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    assert p.returncode == 2 and line is None


@pytest.mark.timeout(300)
def test_under_torch_distributed_run_rank0_coordinates_the_extra():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1                                          # ONE line, rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and len(set(line["extra"]["worker_pids"])) == 2 and "launcher" not in line["extra"]
    dp = line["extra"]["dist_potrf"]
    assert dp["status"] == "ok" and not (set(dp["worker_pids"]) & set(line["extra"]["worker_pids"]))
