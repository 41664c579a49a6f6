import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def host_threads():
    """Intra-op threads for the CPU side of the tests (the oracle).  torch's default is one thread per host core - 256 on the GPU box -
    and on the oracle's small convolutions / 3 200 - 12 800-row GEMMs that is 20 - 90x SLOWER than 16 (measured there, round 5:
    32-cloud encoder slice 25 s vs 1.4 s; one 12 800-row score evaluation 4.8 s vs 55 ms; profiles/r5_oracle_host_timing.txt)."""
    return max(1, min(int(os.environ.get("GP_TEST_THREADS", "16")), os.cpu_count() or 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.environ.setdefault("OMP_NUM_THREADS", str(host_threads()))  # inherited by every process the tests spawn
    import torch
    torch.set_num_threads(host_threads())


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
