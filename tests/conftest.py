import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """The HIP runtime of THIS process comes up before any test starts a child process that uses the GPU (tests/test_gpu_cli.py
    runs the command-line program): a parent that initialises HIP just as such a child is being torn down has been seen to find
    no device (hipGetDeviceCount -> 0), depending on which tests were selected.  No GPU: nothing happens."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    """HIP context of the product library; fails loudly (no fallback) when the device or the library is missing."""
    # torch brings its own copy of the HIP runtime: where a test also hands torch device buffers to the library (as
    # bench.py and ShardedTree do), torch initialises first, like there -- a second runtime that comes up after the
    # library's has been seen to find no device on some boxes
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from linearsfm_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
