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
