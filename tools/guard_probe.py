"""Does memory from HIP's virtual-memory API keep stale contents visible to kernels when it is unmapped, released and re-allocated?  (tests/guard_alloc.c quarantines
freed blocks because of what this probe shows.)  Per round: guard buffer X <- pattern r by libxsmm_hip_memcpy_h2d, Y = copy of X by a TPP kernel, compare Y with the
pattern; free both with reuse on / off."""
import ctypes as C, json, os, sys, tempfile
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import guard
from libxsmm_amd import capi
from libxsmm_amd.capi import DT
api = capi.load()
lib = guard.load(guard.build(tempfile.mkdtemp(prefix="guard_")))
lib.guard_set_reuse.argtypes = [C.c_int]
m, n = 64, 64
copy = api.dispatch_meltw_unary(capi.UNARY.IDENTITY, capi.UnaryShape(m, n, m, m, DT.F32, DT.F32, DT.F32), 0)
out = {"granularity": int(lib.guard_granularity())}
for reuse in (1, 0):
    lib.guard_set_reuse(reuse)
    bad, addrs = 0, set()
    for r in range(20):
        pat = np.full((n, m), float(r + 1), dtype=np.float32)
        X = guard.GuardBuf(pat); Y = guard.GuardBuf(np.zeros_like(pat))
        addrs.add(X.data_ptr())
        p = capi.UnaryParam(); p.in_.primary, p.out.primary = X.data_ptr(), Y.data_ptr()
        capi.Api.call(copy, p); api.hip_sync(); api.check()
        got = Y.numpy()
        bad += int(not np.array_equal(got, pat))
        del X, Y
    out["reuse" if reuse else "quarantine"] = {"rounds": 20, "wrong_rounds": bad, "distinct_addresses_of_X": len(addrs)}
print(json.dumps(out))
