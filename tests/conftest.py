import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _release_gpu_contexts():
    """Detector contexts hold gigabytes of HBM (frame rings, pooled Hough buffers) and pinned host memory; a test that lets one go
    out of scope without close() keeps it until the garbage collector gets to it.  Collect after every test, so that a long
    session's footprint is one test's, not the sum of all of them."""
    yield
    import gc
    gc.collect()
