import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


# The CPU oracle is a chain of small torch ops: on a GPU box's 128 hardware threads torch's intra-op pool makes them 10x SLOWER than on 8
# (bench.py's cpu_baseline probe: 6 forwards/s at 128 threads, 68-79 at 8-16), and four pods share those cores.
torch.set_num_threads(min(8, os.cpu_count() or 8))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().flatten()
    b = torch.as_tensor(b).detach().double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "reference_outputs.npz"))


@pytest.fixture(scope="session")
def golden_tables():
    return np.load(os.path.join(GOLDEN, "tables.npz"))
