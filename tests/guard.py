"""Guarded device buffers for the out-of-bounds tests (tests/test_oob_guard_gpu.py): every operand is placed so that its LAST byte (or, `front`, its FIRST
byte) is the last (first) mapped byte of its own virtual-address reservation -- tests/guard_alloc.c, HIP's virtual-memory API.  A kernel that touches one element
outside an operand page-faults, which aborts the process: the runner below is therefore executed in a subprocess and prints the case it is about to run."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build(outdir):
    so = os.path.join(outdir, "libguard.so")
    cmd = ["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(HERE, "guard_alloc.c"),
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return so


def load(so):
    global _lib
    _lib = C.CDLL(so)
    _lib.guard_alloc.restype, _lib.guard_alloc.argtypes = C.c_void_p, [C.c_size_t, C.c_int]
    _lib.guard_free.restype, _lib.guard_free.argtypes = None, [C.c_void_p]
    _lib.guard_granularity.restype = C.c_size_t
    return _lib


class GuardBuf:
    """Device image of a numpy array whose end (front = False) or start (front = True) touches unmapped address space."""

    def __init__(self, arr, front=False):
        from libxsmm_amd import capi
        self.api = capi.load()
        self.arr = np.ascontiguousarray(arr)
        self.nbytes = max(self.arr.nbytes, 1)
        self.ptr = _lib.guard_alloc(self.nbytes, 1 if front else 0)
        assert self.ptr, "guard_alloc failed (no virtual-memory API on this box?)"
        if self.arr.nbytes:
            assert self.api.hip_memcpy_h2d(self.ptr, self.arr.ctypes.data, self.arr.nbytes) == 0

    def data_ptr(self):
        return self.ptr

    def cpu(self):
        return self

    def numpy(self):
        out = np.empty_like(self.arr)
        if out.nbytes:
            assert self.api.hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes) == 0
        return out

    def __del__(self):
        try:
            _lib.guard_free(self.ptr)
        except Exception:
            pass


def hook(front):
    return lambda x: GuardBuf(x, front)
