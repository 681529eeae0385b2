"""Which descriptors the dispatcher accepts and which it refuses -- host logic, so it is checked without a GPU: with LIBXSMM_HIP_DRYRUN=1 a
machine without a device dispatches (handles cannot be called).  One row per precision line of the reference's samples/xgemm/gemm_kernel.c
(its table of accepted combinations, :3872-3940) plus the refusals this library documents.  Runs in a child process: the
dry-run switch is read once, at libxsmm_init."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F
api = capi.load()
V, VB, TB, IL, B0, DC = F.VNNI_A, F.VNNI_B, F.TRANS_B, F.INTLV_A_FORMAT, F.BETA_0, F.DECOMPRESS_A_VIA_BITMASK
rows = [
  # name, a, b, c, comp, flags, (m, n, k)
  ("f32", DT.F32, DT.F32, DT.F32, DT.F32, 0, (23, 23, 23)),
  ("f64", DT.F64, DT.F64, DT.F64, DT.F64, 0, (23, 23, 23)),
  ("bf32", DT.BF32, DT.BF32, DT.F32, DT.F32, 0, (32, 32, 32)),
  ("i16", DT.I16, DT.I16, DT.I32, DT.I32, V, (32, 32, 32)),
  ("bf16_vnni", DT.BF16, DT.BF16, DT.BF16, DT.F32, V, (64, 64, 64)),
  ("bf16_f32", DT.BF16, DT.BF16, DT.F32, DT.F32, V, (64, 64, 64)),
  ("f16", DT.F16, DT.F16, DT.F16, DT.F32, V, (32, 32, 32)),
  ("f16_comp_f16", DT.F16, DT.F16, DT.F16, DT.F16, V, (32, 32, 32)),
  ("f16_implicit", DT.F16, DT.F16, DT.F32, DT.IMPLICIT, V, (32, 32, 32)),
  ("u8_i8", DT.U8, DT.I8, DT.I32, DT.I32, V, (32, 32, 64)),
  ("i8_i8_f32", DT.I8, DT.I8, DT.F32, DT.I32, V, (32, 32, 64)),
  ("bf8", DT.BF8, DT.BF8, DT.F32, DT.F32, V, (32, 32, 64)),
  ("bf8_out", DT.BF8, DT.BF8, DT.BF8, DT.F32, V, (32, 32, 64)),
  ("hf8_out", DT.HF8, DT.HF8, DT.HF8, DT.F32, V, (32, 32, 64)),
  ("bf8_x_bf16", DT.BF8, DT.BF16, DT.BF16, DT.F32, V, (32, 32, 64)),
  ("hf8_x_bf16", DT.HF8, DT.BF16, DT.F32, DT.F32, V, (32, 32, 64)),
  ("i8_x_bf16", DT.I8, DT.BF16, DT.BF16, DT.F32, 0, (32, 32, 64)),
  ("mxfp4_x_bf16", DT.MXFP4X2, DT.BF16, DT.BF16, DT.F32, V, (32, 32, 64)),
  ("mxfp4_x_i8", DT.MXFP4X2, DT.I8, DT.F32, DT.I32, V | IL, (32, 32, 64)),
  ("u4_x_u8", DT.U4X2, DT.U8, DT.I32, DT.I32, V | IL, (32, 32, 64)),
  ("i4_x_u8", DT.I4X2, DT.U8, DT.I32, DT.I32, V | IL, (32, 32, 64)),
  ("i2_x_i8", DT.I2X4, DT.I8, DT.I32, DT.I32, V | IL, (32, 32, 64)),
  ("i1_x_u8", DT.I1X8, DT.U8, DT.I32, DT.I32, V, (32, 32, 64)),
  ("mxfp4_mx", DT.MXFP4X2, DT.MXFP4X2, DT.F32, DT.F32, V | VB | TB, (32, 32, 64)),
  ("mxfp4_mx_out", DT.MXFP4X2, DT.MXFP4X2, DT.MXFP4X2, DT.F32, V | VB | TB | B0, (32, 32, 64)),
  ("mxbf8_mx_out", DT.MXBF8, DT.MXBF8, DT.MXBF8, DT.F32, V | VB | TB | B0, (32, 32, 64)),
  ("mxhf8_mx", DT.MXHF8, DT.MXHF8, DT.F32, DT.F32, V | VB | TB, (32, 32, 64)),
  ("mxbf6_mx", DT.MXBF6, DT.MXBF6, DT.F32, DT.F32, V | VB | TB, (32, 32, 64)),
  ("mxhf6_mx", DT.MXHF6, DT.MXHF6, DT.F32, DT.F32, V | VB | TB, (32, 32, 64)),
  ("bitmask_f32", DT.F32, DT.F32, DT.F32, DT.F32, DC, (32, 32, 32)),
  ("bitmask_bf16", DT.BF16, DT.BF16, DT.BF16, DT.F32, DC | V, (32, 32, 32)),
  # refused on purpose
  ("no:mx_out_beta1", DT.MXFP4X2, DT.MXFP4X2, DT.MXFP4X2, DT.F32, V | VB | TB, (32, 32, 64)),
  ("no:mxhf8_out", DT.MXHF8, DT.MXHF8, DT.MXHF8, DT.F32, V | VB | TB | B0, (32, 32, 64)),
  ("no:mx_without_vnni", DT.MXFP4X2, DT.MXFP4X2, DT.F32, DT.F32, 0, (32, 32, 64)),
  ("no:i2_not_interleaved", DT.I2X4, DT.I8, DT.I32, DT.I32, V, (32, 32, 64)),
  ("no:i1_interleaved", DT.I1X8, DT.I8, DT.I32, DT.I32, V | IL, (32, 32, 64)),
  ("no:i8_x_i8_bf16", DT.I8, DT.I8, DT.BF16, DT.I32, V, (32, 32, 64)),
  ("no:u8_x_bf16", DT.U8, DT.BF16, DT.BF16, DT.F32, 0, (32, 32, 64)),
  ("no:i8_x_f16", DT.I8, DT.F16, DT.F16, DT.F32, 0, (32, 32, 64)),
  ("no:f16_to_bf16", DT.F16, DT.F16, DT.BF16, DT.F32, V, (32, 32, 32)),
  ("no:mixed_f64_f32", DT.F64, DT.F32, DT.F32, DT.F32, 0, (32, 32, 32)),
  ("no:bitmask_batch_reduce", DT.F32, DT.F32, DT.F32, DT.F32, DC | F.BATCH_REDUCE_STRIDE, (32, 32, 32)),
  ("no:bf16_odd_k_vnni", DT.BF16, DT.BF16, DT.BF16, DT.F32, V, (32, 32, 31)),
]
out = {}
for name, a, b, c, comp, flags, (m, n, k) in rows:
    ldb = n if (flags & TB) else k
    shape = capi.gemm_shape(m, n, k, m, ldb, m, a, b, c, comp)
    if flags & F.BATCH_REDUCE_STRIDE:
        h = api.dispatch_brgemm(shape, flags & ~F.BATCH_REDUCE_STRIDE, 0, capi.br_config(capi.BR_STRIDE, 4096, 4096, 0))
    else:
        h = api.dispatch_gemm(shape, flags, 0)
    out[name] = bool(h)
print(json.dumps(out))
"""


def test_dispatcher_accepts_and_refuses_what_the_documentation_says():
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_VERBOSE="0")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    wrong = {k: v for k, v in got.items() if v == k.startswith("no:")}
    assert not wrong, f"accepted / refused against the table: {wrong}"
    assert sum(1 for k in got if not k.startswith("no:")) >= 30


TPP_CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, UNARY, UNARY_FLAG as UF, BINARY, BINARY_FLAG as BF, TERNARY, TERNARY_FLAG as TF
api = capi.load()
U = lambda i, o, m=64, n=48, ldi=64, ldo=64: capi.UnaryShape(m, n, ldi, ldo, i, o, DT.F32)
B = lambda i0, i1, o, m=64, n=48: capi.BinaryShape(m, n, 64, 64, 64, i0, i1, o, DT.F32)
out = {}
def u(name, typ, shape, flags=0): out[name] = bool(api.dispatch_meltw_unary(typ, shape, flags))
def b(name, typ, shape, flags=0): out[name] = bool(api.dispatch_meltw_binary(typ, shape, flags))
u("identity_f32_bf16", UNARY.IDENTITY, U(DT.F32, DT.BF16))
u("tanh_bf16", UNARY.TANH, U(DT.BF16, DT.BF16))
u("gelu_inv", UNARY.GELU_INV, U(DT.F32, DT.F32))
u("relu_bitmask", UNARY.RELU, U(DT.F32, DT.F32), UF.BITMASK_2BYTEMULT)
u("exp_hf8", UNARY.EXP, U(DT.HF8, DT.HF8))
u("transpose_f64", UNARY.TRANSFORM_NORM_TO_NORMT, capi.UnaryShape(16, 24, 16, 24, DT.F64, DT.F64, DT.F64))
u("norm_to_vnni4_i8", UNARY.TRANSFORM_NORM_TO_VNNI4, U(DT.I8, DT.I8))
u("reduce_cols_add", UNARY.REDUCE_X_OP_ADD, U(DT.F32, DT.F32), UF.REDUCE_COLS)
u("reduce_x_x2_rows", UNARY.REDUCE_X_X2_OP_ADD, U(DT.BF16, DT.F32), UF.REDUCE_ROWS)
u("reduce_listed_cols_max", UNARY.REDUCE_COLS_IDX_OP_MAX, U(DT.F32, DT.F32), UF.IDX_SIZE_4BYTES)
u("dropout", UNARY.DROPOUT, U(DT.F32, DT.F32), UF.BITMASK_2BYTEMULT)
u("dropout_inv", UNARY.DROPOUT_INV, U(DT.BF16, DT.BF16), UF.BITMASK_2BYTEMULT)
u("stochastic_bf8", UNARY.IDENTITY, U(DT.F32, DT.BF8), UF.STOCHASTIC_ROUND)
u("quant_i8", UNARY.QUANT, U(DT.F32, DT.I8))
u("quant_mxfp4", UNARY.QUANT, U(DT.F32, DT.MXFP4X2))
u("quant_nvfp4", UNARY.QUANT, U(DT.BF16, DT.NVFP4X2))
u("dequant_i16", UNARY.DEQUANT, U(DT.I16, DT.F32))
u("gather_cols", UNARY.GATHER, U(DT.F32, DT.F32), UF.GS_COLS | UF.IDX_SIZE_4BYTES)
u("replicate_col_var_n0", UNARY.REPLICATE_COL_VAR, U(DT.F32, DT.F32, n=0))
b("add_bcast_col", BINARY.ADD, B(DT.BF16, DT.BF16, DT.BF16), BF.BCAST_COL_IN_0)
b("cmp_gt_bitmask", BINARY.CMP_OP_GT, B(DT.F32, DT.F32, DT.F32), BF.BITMASK_2BYTEMULT)
b("zip", BINARY.ZIP, B(DT.U16, DT.U16, DT.F32))
b("dot_to_scalar", BINARY.MUL_AND_REDUCE_TO_SCALAR_OP_ADD, B(DT.BF16, DT.F32, DT.F32))
b("add_stochastic_bf8", BINARY.ADD, B(DT.F32, DT.F32, DT.BF8), BF.STOCHASTIC_ROUND)
out["select"] = bool(api.dispatch_meltw_ternary(TERNARY.SELECT, capi.TernaryShape(64, 48, 64, 64, 64, 64, DT.F32, DT.F32, DT.IMPLICIT, DT.F32, DT.F32), TF.BITMASK_2BYTEMULT))
out["muladd"] = bool(api.dispatch_meltw_ternary(TERNARY.MULADD, capi.TernaryShape(64, 48, 64, 64, 64, 64, DT.BF16, DT.BF16, DT.BF16, DT.BF16, DT.F32), 0))
# refused on purpose
b("no:matmul_as_a_tpp", BINARY.MATMUL, B(DT.F32, DT.F32, DT.F32, m=8, n=8))
u("no:stochastic_to_bf16", UNARY.IDENTITY, U(DT.F32, DT.BF16), UF.STOCHASTIC_ROUND)
u("no:dropout_with_broadcast", UNARY.DROPOUT, U(DT.F32, DT.F32), UF.BITMASK_2BYTEMULT | UF.BCAST_COL)
b("no:dot_to_scalar_f64", BINARY.MUL_AND_REDUCE_TO_SCALAR_OP_ADD, capi.BinaryShape(64, 48, 64, 64, 64, DT.F64, DT.F64, DT.F64, DT.F64))
print(json.dumps(out))
"""


def test_tpp_dispatcher_accepts_and_refuses_what_the_documentation_says():
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_VERBOSE="0")
    r = subprocess.run([sys.executable, "-c", TPP_CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    wrong = {k: v for k, v in got.items() if v == k.startswith("no:")}
    assert not wrong, f"accepted / refused against the table: {wrong}"


CREATOR_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F
api = capi.load()
rowptr = np.array([0, 2, 3, 3, 5], dtype=np.uint32); colidx = np.array([0, 2, 1, 0, 3], dtype=np.uint32)
v32, v64 = np.ones(5, dtype=np.float32), np.ones(5, dtype=np.float64)
P = 64
out = {}

def csr(name, dt_a, dt_b, dt_c, lda, ldb, ldc, flags=0, pw=P, vals=v32, rp=rowptr, ci=colidx, fn=None):
    shape = capi.gemm_shape(4, 4, 4, lda, ldb, ldc, dt_a, dt_b, dt_c, dt_a)
    fn = fn or api.create_packed_spgemm_csr
    out[name] = bool(fn(shape, flags, 0, pw, rp.ctypes.data if rp is not None else None, ci.ctypes.data if ci is not None else None, vals.ctypes.data if vals is not None else None))

# packed CSR / CSC: A-sparse (lda = 0), B-sparse (ldb = 0), C-sparse (ldc = 0) [ref: src/libxsmm_main.c:3553-3640, generator_packed_spgemm.c:27-81]
csr("csr_asparse_f32", DT.F32, DT.F32, DT.F32, 0, 4, 4)
csr("csr_asparse_f64", DT.F64, DT.F64, DT.F64, 0, 4, 4, vals=v64)
csr("csr_bsparse_f32", DT.F32, DT.F32, DT.F32, 4, 0, 4)
csr("no:csr_csparse_f32", DT.F32, DT.F32, DT.F32, 4, 4, 0)                                  # C-sparse exists for CSC only, as in the reference
csr("csc_csparse_f32", DT.F32, DT.F32, DT.F32, 4, 4, 0, fn=api.create_packed_spgemm_csc)
csr("csc_bsparse_f32", DT.F32, DT.F32, DT.F32, 4, 0, 4, fn=api.create_packed_spgemm_csc)
csr("no:csr_bf16", DT.BF16, DT.BF16, DT.BF16, 0, 4, 4)
csr("no:csr_mixed_types", DT.F32, DT.F64, DT.F32, 0, 4, 4)
csr("no:csr_f32_to_f64", DT.F32, DT.F32, DT.F64, 0, 4, 4)
csr("no:csr_width_0", DT.F32, DT.F32, DT.F32, 0, 4, 4, pw=0)
csr("no:csr_trans_a", DT.F32, DT.F32, DT.F32, 0, 4, 4, flags=F.TRANS_A)
csr("no:csr_null_values", DT.F32, DT.F32, DT.F32, 0, 4, 4, vals=None)
csr("no:csr_null_rowptr", DT.F32, DT.F32, DT.F32, 0, 4, 4, rp=None)
csr("no:csr_all_dense", DT.F32, DT.F32, DT.F32, 4, 4, 4)
csr("no:csr_two_sparse", DT.F32, DT.F32, DT.F32, 0, 0, 4)
csr("no:csr_ldb_below_n", DT.F32, DT.F32, DT.F32, 0, 2, 4)
csr("no:csr_half_tilecfg", DT.F32, DT.F32, DT.F32, 0, 4, 4, flags=F.NO_RESET_TILECONFIG)

def bcsc(name, dt_a, dt_b, dt_c, comp, flags, K=256, N=64, bk=32, bn=16, pw=64, ldb=0):
    shape = capi.gemm_shape(8, 0, K, K, ldb, N, dt_a, dt_b, dt_c, comp)
    out[name] = bool(api.create_packed_spgemm_bcsc(shape, flags, 0, capi.SpgemmConfig(pw, bk, bn)))

# BCSC [ref: src/libxsmm_main.c:3640-3700, samples/xgemm_sparse/spmm_kernel.c:423-456]: BASELINE configs[3] in its three precisions
bcsc("bcsc_bf16", DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A)
bcsc("bcsc_bf16_flat", DT.BF16, DT.BF16, DT.BF16, DT.F32, 0)
bcsc("bcsc_bf16_f32out", DT.BF16, DT.BF16, DT.F32, DT.F32, F.VNNI_A)
bcsc("bcsc_f32", DT.F32, DT.F32, DT.F32, DT.F32, 0)
bcsc("bcsc_u8_i8", DT.U8, DT.I8, DT.I32, DT.I32, F.VNNI_A)
bcsc("bcsc_i8_u8", DT.I8, DT.U8, DT.I32, DT.I32, F.VNNI_A)
bcsc("bcsc_bf16_bn32", DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A, bn=32)
bcsc("no:bcsc_f32_vnni", DT.F32, DT.F32, DT.F32, DT.F32, F.VNNI_A)
bcsc("no:bcsc_i8_flat", DT.U8, DT.I8, DT.I32, DT.I32, 0)
bcsc("no:bcsc_f16", DT.F16, DT.F16, DT.F16, DT.F32, F.VNNI_A)
bcsc("no:bcsc_trans_b", DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A | F.TRANS_B)
bcsc("no:bcsc_k_not_blocks", DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A, K=250)
bcsc("no:bcsc_n_not_blocks", DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A, N=72)
bcsc("no:bcsc_width_0", DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A, pw=0)
bcsc("no:bcsc_ldb_set", DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A, ldb=64)
# a handle created without a device cannot compute anything: calling it is a loud error and leaves C alone (no CPU path)
shape = capi.gemm_shape(4, 4, 4, 0, 4, 4, DT.F32, DT.F32, DT.F32, DT.F32)
h = api.create_packed_spgemm_csr(shape, 0, 0, P, rowptr.ctypes.data, colidx.ctypes.data, v32.ctypes.data)
B, Cm = np.ones(4 * 4 * P, dtype=np.float32), np.zeros(4 * 4 * P, dtype=np.float32)
p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = v32.ctypes.data, B.ctypes.data, Cm.ctypes.data
capi.Api.call(h, p)
print("CALL " + json.dumps({"error": int(api.hip_get_last_error()), "c_sum": float(Cm.sum())}))
print("TABLE " + json.dumps(out))
"""


def test_sparse_creators_accept_and_refuse_what_the_documentation_says():
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_HIP_JIT="0")
    env.pop("LIBXSMM_VERBOSE", None)
    r = subprocess.run([sys.executable, "-c", CREATOR_CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    table = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("TABLE ")][-1][6:])
    wrong = {k: v for k, v in table.items() if v == k.startswith("no:")}
    assert not wrong, f"accepted / refused against the table: {wrong}"
    assert len(table) == 32
    call = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("CALL ")][-1][5:])
    assert call["error"] != 0 and call["c_sum"] == 0.0 and "no HIP device: kernel not launched" in r.stderr


MEQN_CHILD = r"""
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from libxsmm_amd import capi
from libxsmm_amd.capi import BINARY, DT, UNARY
import test_meqn as tm
api = capi.load()
out = {}
for name, (tree, shapes, oshape) in sorted(tm.CASES.items()):
    h = api.dispatch_meqn(tm.build(api, tree, shapes), capi.MeqnArgShape(*oshape))
    out[name] = api.hip_kernel_name(h, 0).decode() if h else None
# an operator whose second operand was never pushed [ref: src/libxsmm_matrixeqn.c: the tree must be complete at dispatch]
idx = api.meqn_create()
api.meqn_push_back_binary_op(capi.MeqnMetadata(idx, -1), BINARY.ADD, DT.F32, 0)
api.meqn_push_back_arg(capi.MeqnMetadata(idx, 0), capi.MeqnArgShape(8, 8, 8, DT.F32), tm.SINGULAR)
out["no:incomplete"] = bool(api.dispatch_meqn(idx, capi.MeqnArgShape(8, 8, 8, DT.F32)))
out["no:unknown_equation"] = bool(api.dispatch_meqn(12345, capi.MeqnArgShape(8, 8, 8, DT.F32)))
print("TABLE " + json.dumps(out))
"""


def test_equations_dispatch_and_fuse_as_documented():
    """Every tree of tests/test_meqn.py dispatches without a device; with LIBXSMM_HIP_JIT=2 exactly the documented set becomes ONE generated kernel
    (element-wise: meqn_jit_e..., with reductions to one number: meqn_jit_r...), the rest stays a chain of TPP / GEMM launches."""
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_HIP_JIT="2")
    env.pop("LIBXSMM_VERBOSE", None)
    r = subprocess.run([sys.executable, "-c", MEQN_CHILD % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    table = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("TABLE ")][-1][6:])
    assert table.pop("no:incomplete") is False and table.pop("no:unknown_equation") is False
    assert all(table.values()), {k: v for k, v in table.items() if not v}
    elementwise = {"simple", "bias_relu_bf16", "ternary_muladd", "mixed_precision", "tanh_sigmoid_chain", "layernorm_affine"}
    phased = {"dot_to_scalar", "mul_dot_to_scalar", "softmax_fwd", "softmax_bwd", "sum_of_squares",
              "reduce_bcast"}          # a vector-valued reduction broadcast back: a phase of the one-workgroup kernel up to 2^14 elements (round 3)
    for name, kernel in table.items():
        expected = "meqn_jit_e" if name in elementwise else "meqn_jit_r" if name in phased else None
        if expected:
            assert kernel.startswith(expected), (name, kernel)
        else:
            assert not kernel.startswith("meqn_jit"), (name, kernel)          # MATMUL nodes (and vector-valued reductions above 2^14 elements): a chain of launches


MEQN_HEADS_CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, UNARY, UNARY_FLAG as UF, BINARY, TERNARY, TERNARY_FLAG as TF
api = capi.load()
SINGULAR = capi.MatrixArgAttributes(0, 0, 0, 0)
out = {}
def scatter_head(name, sum_type, out_type):
    idx = api.meqn_create()
    md = lambda pos=-1: capi.MeqnMetadata(idx, pos)
    api.meqn_push_back_unary_op(md(), UNARY.SCATTER, out_type, UF.GS_COLS | UF.IDX_SIZE_4BYTES)
    api.meqn_push_back_binary_op(md(), BINARY.ADD, sum_type, 0)
    api.meqn_push_back_arg(md(0), capi.MeqnArgShape(32, 8, 32, out_type), SINGULAR)
    api.meqn_push_back_arg(md(1), capi.MeqnArgShape(32, 8, 32, out_type), SINGULAR)
    out[name] = bool(api.dispatch_meqn(idx, capi.MeqnArgShape(32, 24, 32, out_type)))
scatter_head("scatter_head_f32", DT.F32, DT.F32)
scatter_head("scatter_head_bf16_sum_bf16", DT.BF16, DT.BF16)
scatter_head("no:scatter_head_bf16_sum_f32", DT.F32, DT.BF16)          # a 4-byte operand scattered into 2-byte columns would be written past them
def scatter_inside():
    idx = api.meqn_create()
    md = lambda pos=-1: capi.MeqnMetadata(idx, pos)
    api.meqn_push_back_unary_op(md(), UNARY.TANH, DT.F32, 0)
    api.meqn_push_back_unary_op(md(), UNARY.SCATTER, DT.F32, UF.GS_COLS | UF.IDX_SIZE_4BYTES)
    api.meqn_push_back_arg(md(0), capi.MeqnArgShape(32, 8, 32, DT.F32), SINGULAR)
    out["no:scatter_below_the_head"] = bool(api.dispatch_meqn(idx, capi.MeqnArgShape(32, 8, 32, DT.F32)))
scatter_inside()
def gemm_head(name, flags):
    idx = api.meqn_create()
    md = lambda pos=-1: capi.MeqnMetadata(idx, pos)
    api.meqn_push_back_ternary_op(md(), TERNARY.MATMUL, DT.F32, flags)
    api.meqn_push_back_arg(md(0), capi.MeqnArgShape(32, 16, 32, DT.F32), SINGULAR)
    api.meqn_push_back_arg(md(1), capi.MeqnArgShape(16, 48, 16, DT.F32), SINGULAR)
    api.meqn_push_back_arg(md(2), capi.MeqnArgShape(32, 48, 32, DT.F32), SINGULAR)
    out[name] = bool(api.dispatch_meqn(idx, capi.MeqnArgShape(32, 48, 40, DT.F32)))
gemm_head("accumulating_matmul_head", TF.REUSE_IN_2_AS_OUT)
gemm_head("no:ternary_matmul_without_reuse", 0)
print(json.dumps(out))
"""


def test_equation_heads_accepted_and_refused():
    """SCATTER exists as the head only and copies elements of its operand's width; a MATMUL that accumulates into its third operand may be the head (round 3)."""
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_VERBOSE="0")
    r = subprocess.run([sys.executable, "-c", MEQN_HEADS_CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    wrong = {k: v for k, v in got.items() if v == k.startswith("no:")}
    assert not wrong, f"accepted / refused against the table: {wrong}"


PLAN_CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F
api = capi.load()
out = {}
def g(name, m, n, k, a, b, c, comp, flags=0, lda=None, ldb=None, ldc=None):
    ta, tb = bool(flags & F.TRANS_A), bool(flags & F.TRANS_B)
    sh = capi.gemm_shape(m, n, k, lda or (k if ta else m), ldb or (n if tb else k), ldc or m, a, b, c, comp)
    h = api.dispatch_gemm(sh, flags | F.BETA_0, 0)
    out[name] = api.hip_kernel_name(h, 0).decode() if h else None
g("f32_32", 32, 32, 32, DT.F32, DT.F32, DT.F32, DT.F32)
g("f32_23", 23, 23, 23, DT.F32, DT.F32, DT.F32, DT.F32)
g("f64_32", 32, 32, 32, DT.F64, DT.F64, DT.F64, DT.F64)
g("f64_23", 23, 23, 23, DT.F64, DT.F64, DT.F64, DT.F64)
g("bf16_64_vnni", 64, 64, 64, DT.BF16, DT.BF16, DT.BF16, DT.F32, F.VNNI_A)
g("i8_64", 64, 64, 64, DT.I8, DT.I8, DT.I32, DT.I32, F.VNNI_A)
g("i8_96x64", 96, 64, 64, DT.U8, DT.I8, DT.I32, DT.I32, F.VNNI_A)
g("i8_40", 40, 40, 40, DT.U8, DT.I8, DT.I32, DT.I32, F.VNNI_A)
g("i8_32x32x48", 32, 32, 48, DT.I8, DT.U8, DT.I32, DT.I32, F.VNNI_A)
g("i8_flat", 12, 10, 7, DT.I8, DT.I8, DT.I32, DT.I32)
g("bf8_64", 64, 64, 64, DT.BF8, DT.BF8, DT.F32, DT.F32, F.VNNI_A)
g("hf8_17x9x12", 17, 9, 12, DT.HF8, DT.HF8, DT.F32, DT.F32, F.VNNI_A, ldc=20)
g("hf8_40_c8", 40, 40, 40, DT.HF8, DT.HF8, DT.HF8, DT.F32, F.VNNI_A)
g("hf8_flat", 12, 10, 7, DT.HF8, DT.HF8, DT.F32, DT.F32)
print("TABLE " + json.dumps(out))
"""


def test_dense_plan_names_without_a_device():
    """Which kernel family the planner names for a descriptor (host logic: LIBXSMM_HIP_DRYRUN=1).  Round 4: nothing with 8-bit operands in VNNI-4 and whole k-quads is
    left on the one-element-per-thread kernel, no f64 descriptor at all; launch-time refinements (streaming / blocked / lean variants) are the GPU tests' matter."""
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_VERBOSE="0")
    r = subprocess.run([sys.executable, "-c", PLAN_CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    t = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("TABLE ")][-1][6:])
    assert all(v is not None for v in t.values()), t
    assert "mfma_f32" in t["f32_32"] and "mfma_f32" in t["f32_23"]
    assert "f64" in t["f64_32"] and "f64" in t["f64_23"] and "generic" not in t["f64_23"]
    assert t["bf16_64_vnni"] == "gemm_mfma_bf16_kernel<2,2>"
    assert t["i8_64"] == "gemm_i8_stream_kernel<2,2>" and t["i8_96x64"] == "gemm_i8_stream_kernel<1,1>"
    assert t["i8_40"] == "gemm_mfma_8bit_kernel<2,2>" and t["i8_32x32x48"] == "gemm_mfma_8bit_kernel<1,1>"
    assert t["bf8_64"] == "gemm_fp8_stream_kernel<2,2>"
    assert t["hf8_17x9x12"] == "gemm_mfma_8bit_kernel<1,1>" and t["hf8_40_c8"] == "gemm_mfma_8bit_kernel<2,2>"
    assert t["i8_flat"] == "gemm_generic_kernel" and t["hf8_flat"] == "gemm_generic_kernel"
