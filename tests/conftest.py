import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class _G:
        def __init__(self):
            self._cache = {}

        def __call__(self, name):
            if name not in self._cache:
                self._cache[name] = np.load(os.path.join(GOLDEN, f"ref_{name}.npz"), allow_pickle=False)
            return self._cache[name]

    return _G()
