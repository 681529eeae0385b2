"""The planner's decisions as ONE committed table (round-4 review item 9): tests/golden/plan_table.json holds, for a grid of descriptors (types x shapes x flags x
leading dimensions), the kernel family plan_gemm names at dispatch time (host logic only: LIBXSMM_HIP_DRYRUN=1) or null where the dispatcher refuses.  The test
regenerates the table and diffs it -- a planner edit that moves a shape to another family shows up as a reviewable diff of that file
(regenerate: python tests/test_plan_table_cpu.py --write).  Launch-time refinements (streaming / blocked / workgroup forms) are decided per launch from alignment
and batch form and are pinned by the GPU tests' kernel-name assertions."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "plan_table.json")

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F
api = capi.load()
V, VB, TA, TB, IL = F.VNNI_A, F.VNNI_B, F.TRANS_A, F.TRANS_B, F.INTLV_A_FORMAT
TYPES = [("f32", DT.F32, DT.F32, DT.F32, DT.F32, [0, TA, TB]), ("f64", DT.F64, DT.F64, DT.F64, DT.F64, [0, TA]), ("bf16", DT.BF16, DT.BF16, DT.BF16, DT.F32, [V, 0, V | TB | VB]),
         ("bf16_f32", DT.BF16, DT.BF16, DT.F32, DT.F32, [V]), ("f16", DT.F16, DT.F16, DT.F16, DT.F32, [V, 0]), ("u8i8", DT.U8, DT.I8, DT.I32, DT.I32, [V, 0]),
         ("i8i8_f32", DT.I8, DT.I8, DT.F32, DT.I32, [V]), ("bf8", DT.BF8, DT.BF8, DT.F32, DT.F32, [V, 0]), ("hf8_hf8", DT.HF8, DT.HF8, DT.HF8, DT.F32, [V]),
         ("bf8_x_bf16", DT.BF8, DT.BF16, DT.BF16, DT.F32, [V]), ("i8_x_bf16", DT.I8, DT.BF16, DT.BF16, DT.F32, [0]), ("mxfp4_x_bf16", DT.MXFP4X2, DT.BF16, DT.BF16, DT.F32, [V]),
         ("mxfp4_mx", DT.MXFP4X2, DT.MXFP4X2, DT.F32, DT.F32, [V | VB | TB]), ("mxhf6_mx", DT.MXHF6, DT.MXHF6, DT.F32, DT.F32, [V | VB | TB]), ("mxbf6_mx", DT.MXBF6, DT.MXBF6, DT.F32, DT.F32, [V | VB | TB]),
         ("i4_x_u8", DT.I4X2, DT.U8, DT.I32, DT.I32, [V | IL]), ("i2_x_i8", DT.I2X4, DT.I8, DT.I32, DT.I32, [V | IL]), ("i1_x_u8", DT.I1X8, DT.U8, DT.I32, DT.I32, [V]),
         ("bf32", DT.BF32, DT.BF32, DT.F32, DT.F32, [0]), ("i16", DT.I16, DT.I16, DT.I32, DT.I32, [V])]
SHAPES = [(16, 16, 16), (23, 23, 23), (32, 32, 32), (32, 32, 64), (40, 40, 40), (48, 48, 48), (64, 64, 64), (72, 72, 72), (96, 96, 96), (96, 64, 64), (128, 128, 64), (64, 32, 96), (17, 9, 12)]
out = {}
for tn, a, b, c, comp, flagsets in TYPES:
    for fl in flagsets:
        for m, n, k in SHAPES:
            ta, tb = bool(fl & TA), bool(fl & TB)
            sh = capi.gemm_shape(m, n, k, k if ta else m, n if tb else k, m, a, b, c, comp)
            h = api.dispatch_gemm(sh, fl | F.BETA_0, 0)
            out[f"{tn}|flags={int(fl)}|{m}x{n}x{k}"] = api.hip_kernel_name(h, 0).decode() if h else None
print("TABLE " + json.dumps(out, sort_keys=True))
"""


def generate():
    env = dict(os.environ, LIBXSMM_HIP_DRYRUN="1", LIBXSMM_VERBOSE="0")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("TABLE ")][-1][6:])


def test_planner_table_matches_the_committed_one():
    got, want = generate(), json.load(open(GOLD))
    moved = {k: (want.get(k), got.get(k)) for k in sorted(set(got) | set(want)) if got.get(k) != want.get(k)}
    assert not moved, f"planner decisions changed (committed, now) -- regenerate tests/golden/plan_table.json if intended: {dict(list(moved.items())[:12])}"
    assert len(got) > 300 and sum(v is None for v in got.values()) < len(got) // 2


if __name__ == "__main__" and "--write" in sys.argv:
    json.dump(generate(), open(GOLD, "w"), indent=0, sort_keys=True)
    print("wrote", GOLD)
