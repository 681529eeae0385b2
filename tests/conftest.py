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


@pytest.fixture(autouse=True)
def _default_launch_mode(request):
    """GPU tests share one process and the library keeps its launch mode per THREAD: a test that switches to stream-ordered launches (libxsmm_hip_set_stream /
    _set_async, a pipeline section) must not leak that into the next one, whose host-memory operands are only staged by SYNCHRONOUS calls.  After every GPU
    test: close an open section, drain, clear the sticky error, back to synchronous calls on the null stream."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    try:
        from libxsmm_amd import capi
        api = capi.load()
        if api.hip_available() == 1:
            api.hip_sync()
            api.hip_set_stream(None)
            api.hip_set_async(0)
            api.hip_set_streaming_hint(0)
    except Exception:
        pass
