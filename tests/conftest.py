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


def pytest_collection_finish(session):
    """LIBXSMM_TEST_GUARD=end|front (set by tests/test_oob_guard_gpu.py for its pytest SUBPROCESSES): every device operand the parity tests upload through
    tests/helpers.py (GemmCase.run_gpu) or tests/test_sparse_gpu.py (_dev) is placed flush against unmapped address space (tests/guard.py), so the SAME parity
    tests also prove that no kernel loads or stores outside its operands -- an access outside page-faults and aborts the subprocess."""
    side = os.environ.get("LIBXSMM_TEST_GUARD", "")
    if side not in ("end", "front"):
        return
    import tempfile
    import guard
    import helpers
    guard.load(guard.build(tempfile.mkdtemp(prefix="guard_")))
    helpers.UPLOAD_HOOK = guard.hook(side == "front")
    for name in ("test_sparse_gpu", "test_gemm_ragged_gpu", "test_gemm_f64_gpu"):
        mod = sys.modules.get(name)
        if mod is not None and hasattr(mod, "_dev"):
            mod._dev = guard.hook(side == "front")
