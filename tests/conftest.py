import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle.oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference (oracle/_ref/libxsmm_ref.so); tests that need it skip when it was not built."""
    from oracle import pyoracle
    pyoracle.build()
    if not pyoracle.have_reference():
        pytest.skip("oracle/_ref/libxsmm_ref.so not built (no /root/reference here)")
    return pyoracle.reference()


@pytest.fixture(scope="session")
def api():
    from libxsmm_amd import capi
    return capi.load()
