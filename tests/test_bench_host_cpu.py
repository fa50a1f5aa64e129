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
