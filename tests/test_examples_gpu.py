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
