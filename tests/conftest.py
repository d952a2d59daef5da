import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are never silently skipped on a GPU box; on a CPU-only box they are deselected by -m "not gpu"
    # and, if selected anyway, fail loudly inside the test through internvideo_amd.lib.require_gpu().
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
