import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy access to tests/golden/<name>.npz (outputs of the reference)."""

    def __init__(self):
        self._cache = {}

    def __call__(self, name):
        if name not in self._cache:
            self._cache[name] = np.load(os.path.join(HERE, "golden", name + ".npz"))
        return self._cache[name]


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co
    co.build()
    return co
