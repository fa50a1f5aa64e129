"""The reference's own test for this path is "the examples run" (tests/test_examples.py: each example's main()).
Same here for the example flows carried on the HIP path (examples/*.py)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = ["example02_1d_interpolation", "example03_06_remap_2d_and_side_information", "example07_nd_noisy_regression",
            "example10_sample_paths",
            "example11_22_noisy_paths_and_ml"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", EXAMPLES)
def test_example_runs(name, capsys):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    out = capsys.readouterr().out
    assert len(out.strip().splitlines()) >= 2 and "nan" not in out.lower()


@pytest.mark.gpu
def test_distributed_reml_fit_example_runs_on_one_gpu_over_rccl():
    """examples/dist_reml_fit.py as its own process on a 1 x 1 grid over RCCL (the multi-GPU launch is the same script under
    torch.distributed.run): the REML fit moves the parameters to a model that predicts the noise-free function well"""
    import re
    import subprocess
    import sys

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29641")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "dist_reml_fit.py"), "--size-n", "3000", "--size-m", "500", "--block", "512",
                        "--maxiter", "12"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    m = re.search(r"reml=([-+0-9.eE]+) .* noise_sd=([0-9.]+) .* rmse=([0-9.]+)", p.stdout)
    assert m, p.stdout
    assert 0.07 < float(m.group(2)) < 0.14 and float(m.group(3)) < 0.08        # noise sd 0.1 recovered; error well below the noise


@pytest.mark.gpu
def test_bench_under_the_drivers_multi_gpu_launch_line_over_rccl_at_world_1():
    """The driver measures N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` over RCCL.  A
    one-GPU box allows N = 1 of it: with GPMP_BENCH_DIST=force that is still the whole N > 1 code path of bench.py -- the process group
    on the device, barriers, the MAX all-reduce of the timing, `ranks_seen` gathered through the group, rank 0 leaving the group and
    running the distributed extra in a fresh process under RCCL (both transports, the block-cyclic headline step with its values
    check, one prediction, one value + gradient) -- at reduced sizes here; profiles/r5/bench_torchrun_nccl_world1_dist_extra_forced.log
    is the same command at the bench's own sizes."""
    import json
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPMP_BENCH_DIST="force", GPMP_BENCH_DIST_N="6144", GPMP_BENCH_STRONG_NM="4096,3000",
               GPMP_BENCH_DIST_TIMEOUT="300")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GPMP_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29643",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size-n", "4096", "--size-m", "3000",
           "--no-cpu-baseline", "--no-extras", "--no-live-pmc"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["extra"]["backend"] == "nccl"
    assert j["ranks_seen"]["backend"] == "nccl" and j["ranks_seen"]["distinct_devices"] == 1 and j["ranks_seen"]["rccl_version"]
    dp = j["extra"]["dist_potrf"]
    assert dp["status"] == "ok" and dp["backend"] == "nccl" and dp["grid"] == "1x1" and dp["n"] == 6144
    assert dp["bcast_timed"]["info"] == 0 and dp["p2p_timed"]["info"] == 0
    assert dp["strong_scaling_block_cyclic"]["values_check"]["ok"] and dp["predict"]["finite"]
