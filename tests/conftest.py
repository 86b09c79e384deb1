import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class _G:
        def __init__(self):
            self._c = {}

        def __call__(self, group):
            if group not in self._c:
                self._c[group] = np.load(os.path.join(ROOT, "tests", "golden", f"{group}.npz"))
            return self._c[group]
    return _G()
