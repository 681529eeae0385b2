"""The run-time code generators without a GPU.  In dry-run mode (LIBXSMM_HIP_DRYRUN=1, no device) the creators get as far as hiprtc: the
generated source is compiled for gfx950 and kept, never loaded or launched.  LIBXSMM_HIP_JIT_DUMP=<dir> writes every generated kernel's
source and code object, whose AMDGPU metadata gives the register / scratch budget (tools/kernel_resources.py).  Checked here: every
generator's output compiles (fixed-pattern sparse kernels of BASELINE configs[2], B-sparse CSR / CSC, packed GEMMs, FsSpMDM, element-wise
and phased equation kernels), no generated kernel needs scratch, and the packed CSR kernels of config #3 keep three (beta = 0) / two (beta = 1) waves per SIMD."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr   # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists("/opt/rocm/lib/libhiprtc.so") and os.path.exists(os.path.join(kr.LLVM, "llvm-readelf"))),
                                reason="needs hiprtc and the ROCm LLVM tools")

CHILD = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG
import sparse_helpers as sh
import test_meqn as tm
api = capi.load()
rng = np.random.default_rng(555)
names = {}

def keep(label, handle):
    names[label] = api.hip_kernel_name(handle, 0).decode() if handle else None

# BASELINE configs[2]: 35 x 35 at 15 %%, packed width 65 536: A-sparse CSR, f32 / f64, beta 0 / 1
M = K = N = 35
rowptr, colidx = sh.random_csr(rng, M, K, 0.15)
vals = rng.standard_normal(len(colidx))
for dt, tag in ((DT.F32, "f32"), (DT.F64, "f64")):
    v = vals.astype(np.float32 if dt == DT.F32 else np.float64)
    for beta0 in (1, 0):
        shape = capi.gemm_shape(M, N, K, 0, N, N, dt, dt, dt, dt)
        keep(f"csr_asparse_{tag}_beta{1 - beta0}", api.create_packed_spgemm_csr(shape, GEMM_FLAG.BETA_0 if beta0 else 0, 0, 65536, rowptr.ctypes.data, colidx.ctypes.data, v.ctypes.data))
# EDGE-style B-sparse: CSR by rows of k, CSC by columns of n (M = 9 elements rows, P = 4096)
rp, ci = sh.random_csr(rng, 20, 24, 0.2)
vb = rng.standard_normal(len(ci))
shape = capi.gemm_shape(9, 24, 20, 20, 0, 24, DT.F32, DT.F32, DT.F32, DT.F32)
keep("csr_bsparse_f32", api.create_packed_spgemm_csr(shape, 0, 0, 4096, rp.ctypes.data, ci.ctypes.data, vb.astype(np.float32).ctypes.data))
cp, ri, vc = sh.csr_to_csc(rp, ci, vb, 20, 24)
keep("csc_bsparse_f32", api.create_packed_spgemm_csc(shape, 0, 0, 4096, cp.ctypes.data, ri.ctypes.data, vc.astype(np.float32).ctypes.data))
# dense packed GEMMs
shape = capi.gemm_shape(9, 9, 9, 9, 9, 9, DT.F32, DT.F32, DT.F32, DT.F32)
keep("packed_gemm", api.create_packed_gemm(shape, GEMM_FLAG.BETA_0, 0, 4096))
shape = capi.gemm_shape(9, 16, 20, 20, 16, 16, DT.F32, DT.F32, DT.F32, DT.F32)
keep("packed_gemm_ac_rm", api.create_packed_gemm_ac_rm(shape, 0, 0, 4096))
# FsSpMDM (row-major, N = 4800 as in the PyFR driver): an opaque handle, its generated kernel shows up in the dump directory
a_dense = np.zeros((M, K)); a_dense[np.repeat(np.arange(M), np.diff(rowptr)), colidx] = vals
alpha, beta = C.c_double(1.0), C.c_double(0.0)
opaque = {"fsspmdm_f64": bool(api.fsspmdm_create(DT.F64, M, 4800, K, K, 4800, 4800, C.addressof(alpha), C.addressof(beta), a_dense.ctypes.data, 0, None))}
# equations: element-wise trees and trees with reductions to one number (one workgroup, phases)
for name in ("simple", "bias_relu_bf16", "tanh_sigmoid_chain", "layernorm_affine", "mixed_precision", "dot_to_scalar", "softmax_fwd", "softmax_bwd", "sum_of_squares"):
    tree, shapes, out = tm.CASES[name]
    idx = tm.build(api, tree, shapes)
    keep("meqn_" + name, api.dispatch_meqn(idx, capi.MeqnArgShape(*out)))
print("OPAQUE " + json.dumps(opaque))
print("NAMES " + json.dumps(names))
"""


@pytest.fixture(scope="module")
def generated(tmp_path_factory):
    dump = str(tmp_path_factory.mktemp("jit"))
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_HIP_JIT="2", LIBXSMM_HIP_JIT_DUMP=dump)
    env.pop("LIBXSMM_VERBOSE", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("NAMES ")][-1]
    names = json.loads(line[6:])
    opaque = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("OPAQUE ")][-1][7:])
    assert all(opaque.values()), opaque
    table = {}
    for f in sorted(os.listdir(dump)):
        if f.endswith(".co"):
            for t in kr.collect(os.path.join(dump, f)):
                table[t["name"]] = t
    return names, table, dump


def test_every_generator_emits_code_that_compiles(generated):
    names, table, dump = generated
    assert all(names.values()), f"creators that returned NULL in dry-run mode: {[k for k, v in names.items() if not v]}"
    jitted = {k: v for k, v in names.items() if "jit" in v}
    # every sparse / packed creator and every equation is served by a generated kernel (LIBXSMM_HIP_JIT=2), and its code object was kept
    assert set(jitted) == set(names), f"not specialised: {sorted(set(names) - set(jitted))}"
    for label, kernel in jitted.items():
        assert kernel in table, (label, kernel, sorted(table))
        assert os.path.getsize(os.path.join(dump, kernel + ".hip")) > 100
    # FsSpMDM's own kernel (an opaque handle) is in the dump directory as well -- under its own name or, since f64 kernels keep one double per lane
    # (round 3), as the very kernel the packed CSR creator of the same pattern generated (cached by source)
    assert len(table) >= len(set(jitted.values()))


def test_generated_kernels_need_no_scratch(generated):
    _, table, _ = generated
    bad = {n: (t["scratch"], t["spills"], t["vgpr"]) for n, t in table.items() if t["scratch"] > 0}
    assert not bad, f"generated kernels with scratch (bytes, spilled registers, registers): {bad}"


def test_config3_kernel_occupancy(generated):
    names, table, _ = generated
    # 35 rows of X in registers at four floats per lane: 148 registers with beta = 0, 170 with the six C rows that beta = 1 requests ahead
    for label, least in (("csr_asparse_f32_beta0", 3), ("csr_asparse_f32_beta1", 2), ("csr_asparse_f64_beta0", 3), ("csr_asparse_f64_beta1", 2)):
        t = table[names[label]]
        assert t["waves"] >= least, (label, t)
    for label in ("meqn_softmax_fwd", "meqn_dot_to_scalar"):
        assert names[label].startswith("meqn_jit_r"), names[label]      # reductions to one number: the phased one-workgroup kernel
    assert names["meqn_simple"].startswith("meqn_jit_e")
