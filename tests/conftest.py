import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout; registered here too so that the marker is known without the plugin)")


@pytest.fixture(scope="session")
def engine():
    """The initialised HIP library; GPU tests fail (not skip) when it is missing."""
    from rag_system_amd import _lib
    return _lib.engine()
