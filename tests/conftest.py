import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _same_random_inputs_every_run(request):
    """Tests that draw inputs without seeding (a handful) get the same draws in every run: a test either passes always or never."""
    import zlib

    import numpy as np
    import torch
    seed = zlib.crc32(request.node.nodeid.encode()) % (2 ** 31)
    torch.manual_seed(seed)
    np.random.seed(seed % (2 ** 32))
    yield
