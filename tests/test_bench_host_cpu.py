"""Host-side pieces of bench.py that need no GPU: deterministic synthetic inputs and the lookup of the latest
committed PMC pass (profile versions compare numerically: v10 after v9)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_synthetic_inputs_are_deterministic_and_shaped():
    import bench

    xi, zi, xt, theta = bench.synth(64, 10, 8, 0)
    xi2, zi2, xt2, theta2 = bench.synth(64, 10, 8, 0)
    assert np.array_equal(xi, xi2) and np.array_equal(zi, zi2) and np.array_equal(xt, xt2) and np.array_equal(theta, theta2)
    assert xi.shape == (64, 8) and xt.shape == (10, 8) and zi.shape == (64,) and theta.shape == (9,)
    assert np.allclose(zi, np.sin(2 * np.pi * xi[:, 0]) + xi[:, 1:].sum(axis=1))
    assert np.allclose(np.exp(-theta[1:]), 0.5 * (1.0 + np.arange(8) / 8))          # rho_j = 0.5 (1 + j / d)
    assert not np.array_equal(bench.synth(64, 10, 8, 1)[2], xt)                        # another rank, another shard of targets


def test_latest_pmc_pass_is_chosen_numerically(tmp_path, monkeypatch):
    import bench

    prof = tmp_path / "profiles" / "r1"
    prof.mkdir(parents=True)
    for ver, kb in ((9, 100.0), (10, 7.0), (6, 55.0)):
        for name in ("fetch", "write"):
            (prof / f"bench_v{ver}_pmc_{name}_size_by_kernel.csv").write_text(
                "kernel,dispatches,total_KB_raw,per_dispatch_KB_raw\n" f'"void k<true, false, true>(P)",3,{3 * kb},{kb}\n')
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.pmc_traffic_per_launch("k<true, false, true>")
    assert got == (2.0 * 7.0 + 7.0) * 1024.0       # v10: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, in bytes


def test_cpu_baseline_is_a_same_run_measurement_with_a_bounded_child(monkeypatch):
    """(round 5) `cpu_baseline.value` = ONE full-size oracle step measured in a child process of this run (cores stated, not
    assembled); a child that exceeds its limit costs only itself: the assembled model takes over and says so."""
    import bench

    monkeypatch.setenv("GPMP_BENCH_CPU_FULL", "1")
    out = bench.cpu_baseline_for_line(1536, 700, 8, bench._host_threads(), 256)
    assert out["assembled"] is False and out["kind"] == "port" and out["cores"] == bench._host_threads()
    assert out["full_size_run"]["note"] == "measured in THIS run"
    assert abs(out["value"] - 700 / out["full_size_run"]["s_per_step"]) < 1e-9 * out["value"]
    assert out["host_potrf"]["tflops"] > 0 and np.isfinite(out["full_size_run"]["nll"])
    monkeypatch.setenv("GPMP_BENCH_CPU_FULL_TIMEOUT", "0.05")
    out = bench.cpu_baseline_for_line(1536, 700, 8, bench._host_threads(), 256)
    assert out["assembled"] is True and out["sample"].startswith("FALL-BACK") and "exceeded its limit" in out["measured_step_failed"]


def test_latest_committed_cpu_log_is_chosen_by_round_number(tmp_path, monkeypatch):
    import json

    import bench

    for rnd, s in ((4, 175.0), (10, 150.0), (9, 160.0)):
        d = tmp_path / "profiles" / f"r{rnd}"
        d.mkdir(parents=True)
        (d / "cpu_fullsize_step.log").write_text(json.dumps({"tool": "cpu_fullsize_step", "n": 8, "m": 4, "d": 2, "full_step_s": s,
                                                             "points_per_s": 4 / s, "threads": 16}) + "\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench._latest_cpu_fullsize_log(8, 4, 2)["s_per_step"] == 150.0        # r10 is later than r9 and r4
