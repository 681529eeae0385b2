"""Dispatch registry and handle table (one entry per distinct descriptor, thread-local cache in front,
capacity 131072 registered kernels [ref: src/libxsmm_main.h:18-22], hits in tens of nanoseconds).

examples/registry_check.c is a plain C program against include/libxsmm.h.  Without a GPU it runs with
LIBXSMM_HIP_DRYRUN=1 (dispatch works, calling a kernel is an error): the registry, the thunk pool and the per-thread
cache are host code.  The gpu-marked test runs the same binary on the device, where the handles are also callable."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libxsmm_amd", "lib")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("reg") / "registry_check")
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "registry_check.c"), "-L" + LIBDIR, "-lxsmm_amd", "-Wl,-rpath," + LIBDIR, "-lpthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def run(exe, *args, dry=True, **env):
    e = dict(os.environ)
    if dry:
        e["LIBXSMM_HIP_DRYRUN"] = "1"
    e.update({k: str(v) for k, v in env.items()})
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=300, env=e)


def test_registry_holds_131072_kernels_then_refuses(exe):
    r = run(exe, "capacity", 131072 + 5000, 131072)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "capacity=131072 size=131072" in r.stdout


def test_slot_exhaustion_returns_null_and_keeps_earlier_handles(exe):
    r = run(exe, "capacity", 700, 500, LIBXSMM_HIP_MAX_HANDLES=500)
    assert r.returncode == 0, r.stdout + r.stderr


def test_static_trampolines_serve_when_executable_memory_is_unavailable(exe):
    r = run(exe, "capacity", 300, 256, LIBXSMM_HIP_THUNKS=0)
    assert r.returncode == 0, r.stdout + r.stderr


def test_concurrent_dispatch_gives_one_handle_per_descriptor(exe):
    """Eight threads register and hit the same 20 000 descriptors in different orders [ref: tests/threadsafety.c]; the same run under
    ThreadSanitizer is part of tools/sanitize_host.sh."""
    r = run(exe, "threads", 8, 20000)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok threads" in r.stdout


def test_dispatch_hit_is_allocation_free_and_fast(exe):
    r = run(exe, "hit", 2000000)
    assert r.returncode == 0, r.stdout + r.stderr
    ns = float(re.search(r"([\d.]+) ns per dispatch hit", r.stdout).group(1))
    assert ns < 200.0, f"dispatch hit takes {ns} ns"


def test_finalize_invalidates_thread_caches_and_reinitialises(exe):
    r = run(exe, "cycle")
    assert r.returncode == 0, r.stdout + r.stderr


def test_tpp_handles_have_distinct_names_and_release_of_registered_is_a_noop(exe):
    r = run(exe, "info")
    assert r.returncode == 0, r.stdout + r.stderr


def test_without_dryrun_and_without_a_device_dispatch_fails_loudly(exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    r = run(exe, "info", dry=False)
    assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("capacity", 131072 + 100, 131072), ("hit", 2000000), ("cycle",), ("info",)])
def test_registry_on_the_device(exe, args):
    r = run(exe, *args, dry=False)
    assert r.returncode == 0, r.stdout + r.stderr
