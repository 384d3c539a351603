import importlib
import os
import sys

import pytest

try:  # a process that uses both torch's GPU side and libmp2gpu must load torch (and with it torch's own HIP runtime) FIRST, as bench.py
    import torch  # noqa: F401  # does: loaded after libmp2gpu's libamdhip64, torch.cuda finds no device ("No HIP GPUs are available")
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mp2():
    """The product package (directory name has a hyphen, so import it by path)."""
    return importlib.import_module("mapreduce-plonky2_amd")


@pytest.fixture(scope="session")
def ctx(mp2):
    c = mp2.Context(0)
    yield c
    c.close()
