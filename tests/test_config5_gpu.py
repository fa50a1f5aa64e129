"""BASELINE config 5 (n = 131072, 2-D block-cyclic Cholesky) with VALUES on the one GPU of this pool: tools/config5_full.py runs the
single-GPU product path and the real ``BlockCyclicCholesky`` (ranks sharing the GPU over gloo: the schedule, streams and kernels of
the 8-GPU run) on the same inputs and compares log-determinant, NLL, sampled entries of the factor, posterior mean / variance,
kriging weights and the ML value + gradient.  The pool's process guard allows SIX processes on the card and the test runner is
one of them (it has used the GPU in earlier tests), so inside the suite the process form runs on the 2 x 2 grid (n = 16384) and
the 2 x 4 grid of config 5 itself as eight THREAD-ranks (n = 32768: tools/thread_ranks.py, device-resident stream-ordered messages);
``GPMP_TEST_CONFIG5_FULL=1`` for n = 131072 on 2 x 2 processes (K = 137 GB on the single-GPU side, 4 x 34 GB on the distributed side).
Run on its own (``python tools/config5_full.py all``) the tool takes the 2 x 3 grid: profiles/r5/config5_full_n131072_grid2x3_shared_gpu.log.
What is compared is what gpmp/num/numpy_backend.py:465-469 and gpmp/core/likelihood.py:18-52 compute."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, limit):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "config5_full.py"), "all"] + args, env=env, capture_output=True,
                       text=True, timeout=limit)
    assert r.returncode == 0 and "CONFIG5 FULL OK" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    return r.stdout


def test_config5_schedule_with_values_at_n32768_on_the_2x4_grid_of_thread_ranks(tmp_path):
    """the grid of BASELINE config 5 itself: eight ranks as THREADS of one process (tools/thread_ranks.py; every rank its own
    HipLocalOps, streams and buffers on the shared GPU) -- the process guard denies eight processes -- on the DEVICE-RESIDENT branch
    of the product code (what runs under RCCL), with messages ordered by stream events only, as RCCL orders them: no host
    synchronisation hides a missing dependency between the schedule's three streams.  Pc = 4 also takes the gradient's ring through
    all three of its branches (same column set, full pair, half-way pair)."""
    _run(["--threads", "--device-comm", "--grid", "2x4", "--size-n", "32768", "--grad-n", "16384", "--m", "2048", "--limit", "300", "--out", str(tmp_path / "s.npz"),
          "--dist-out", str(tmp_path / "d.npz")], 700)


def test_config5_schedule_with_values_at_n16384_on_the_2x2_grid_of_processes(tmp_path):
    """the host-staged branch: one PROCESS per rank over gloo (the test runner is the fifth GPU process)"""
    _run(["--grid", "2x2", "--size-n", "16384", "--grad-n", "8192", "--m", "2048", "--limit", "300", "--out", str(tmp_path / "s.npz"),
          "--dist-out", str(tmp_path / "d.npz")], 700)


@pytest.mark.skipif(os.environ.get("GPMP_TEST_CONFIG5_FULL") != "1", reason="opt-in (GPMP_TEST_CONFIG5_FULL=1): n = 131072, ~10 minutes, 137 GB")
def test_config5_at_its_own_size_n131072(tmp_path):
    import torch

    if torch.cuda.is_available() and torch.cuda.get_device_properties(0).total_memory < 250e9:
        pytest.skip("needs 250 GB of HBM")
    _run(["--grid", "2x2", "--limit", "1500", "--out", str(tmp_path / "s.npz"), "--dist-out", str(tmp_path / "d.npz")], 3300)
