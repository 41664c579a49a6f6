import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A gpu-marked test selected on a machine without a device (a plain `pytest tests/` run) is skipped, not failed: the product has
    no CPU fallback and would raise GenposeHipError / 'No HIP GPUs are available' from its first call."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items:
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False)")
    for it in gpu_items:
        it.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden
