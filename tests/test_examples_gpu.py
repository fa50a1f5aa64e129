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
