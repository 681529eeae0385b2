"""CPU-side checks of the C-ABI boundary: the library loads without a GPU, exports every symbol the headers
declare, keeps the reference's calling conventions (shapes by value, NULL on illegal input), and FAILS LOUDLY
rather than computing on the host when no HIP device is present."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG


def test_every_declared_symbol_is_exported(api):
    declared = capi.declared_symbols()
    assert len(declared) > 90
    missing = [s for s in declared if not hasattr(api.lib, s)]
    assert not missing, missing


def test_enum_values_match_reference_header():
    """Spot values from include/libxsmm_typedefs.h of the reference (218-246, 278-405, 468-529)."""
    assert (DT.F64, DT.F32, DT.BF16, DT.I32, DT.U8, DT.IMPLICIT, DT.UNSUPPORTED) == (0, 1, 2, 8, 13, 25, 26)
    assert capi.UNARY.RELU == 5 and capi.UNARY.TRANSFORM_NORM_TO_NORMT == 29 and capi.UNARY.GATHER == 51 and capi.UNARY.TRANSFORM_VNNI8_TO_NORM == 76
    assert capi.BINARY.ADD == 1 and capi.BINARY.ZIP == 26 and capi.BINARY.CMP_OP_NE == 32
    assert capi.TERNARY.SELECT == 3 and capi.TERNARY.NMULADD == 4
    assert GEMM_FLAG.BETA_0 == 4 and GEMM_FLAG.VNNI_A == 256 and GEMM_FLAG.USE_XGEMM_EXT_ABI == 4096
    assert GEMM_FLAG.BATCH_REDUCE_ADDRESS == 8192 and GEMM_FLAG.BATCH_REDUCE_STRIDE == 32768
    assert capi.UNARY_FLAG.BITMASK_2BYTEMULT == 1 and capi.UNARY_FLAG.GS_OFFS == 8192 and capi.BINARY_FLAG.BCAST_COL_IN_0 == 4


def test_struct_sizes_are_the_reference_abi():
    # sizeof() of the reference on LP64 (gemm_param 176, gemm_ext_param 368)
    assert C.sizeof(capi.GemmParam) == 176 and C.sizeof(capi.GemmExtParam) == 368
    assert C.sizeof(capi.UnaryParam) == 128 and C.sizeof(capi.BinaryParam) == 176 and C.sizeof(capi.TernaryParam) == 224
    assert C.sizeof(capi.GemmShape) == 40 and C.sizeof(capi.BrConfig) == 16 and C.sizeof(capi.SpgemmConfig) == 12


def test_shard_range_partitions_exactly(api):
    b, e = C.c_size_t(), C.c_size_t()
    for count, gran, world in [(4096, 1, 8), (1 << 20, 1, 8), (10, 1, 4), (7, 1, 8), (4800, 16, 3), (100, 64, 8), (0, 1, 2)]:
        covered, prev = 0, 0
        for r in range(world):
            api.hip_shard_range(count, gran, world, r, C.byref(b), C.byref(e))
            assert b.value == prev and e.value >= b.value and (b.value % gran == 0 or b.value == count)
            covered += e.value - b.value
            prev = e.value
        assert covered == count and prev == count


def test_no_device_means_null_handles_and_a_loud_message():
    """Run in a child process with the GPU hidden: dispatch must return NULL and say why on stderr."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from libxsmm_amd import capi\n"
        "api = capi.load()\n"
        "s = capi.gemm_shape(32,32,32,32,32,32,1,1,1,1)\n"
        "h = api.dispatch_gemm(s, 0, 0)\n"
        "u = api.dispatch_meltw_unary(1, capi.UnaryShape(8,8,8,8,1,1,1), 0)\n"
        "print('HANDLES', h, u, api.hip_available())\n" % capi.ROOT)
    env = {"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1", "PATH": "/usr/bin:/bin", "LD_LIBRARY_PATH": "/opt/rocm/lib"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "HANDLES None None 0" in r.stdout, r.stdout + r.stderr
    assert "no HIP device" in r.stderr and "no CPU path" in r.stderr


def test_product_package_does_not_touch_the_oracle():
    """libxsmm_amd/ (python and C++) must not import, link or load anything under oracle/."""
    import os
    pkg = os.path.join(capi.ROOT, "libxsmm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and "libxsmm_ref" not in text, f
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_public_header_struct_layouts_match_reference_when_compiled_as_c(reference, tmp_path):
    """sizeof / offsetof of the public structs as a C compiler sees include/libxsmm.h == the reference's own
    (first 18 entries = the ctypes probe list, then matdiff / meqn / info structs and two offsets)."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text(r'''
#include <libxsmm.h>
#include <stddef.h>
#include <stdio.h>
int main(void) {
  const size_t s[] = {
    sizeof(libxsmm_gemm_param), sizeof(libxsmm_gemm_ext_param), sizeof(libxsmm_matrix_arg), sizeof(libxsmm_matrix_op_arg),
    sizeof(libxsmm_meltw_unary_param), sizeof(libxsmm_meltw_binary_param), sizeof(libxsmm_meltw_ternary_param),
    sizeof(libxsmm_gemm_shape), sizeof(libxsmm_gemm_batch_reduce_config), sizeof(libxsmm_gemm_ext_unary_argops),
    sizeof(libxsmm_gemm_ext_binary_postops), sizeof(libxsmm_meltw_unary_shape), sizeof(libxsmm_meltw_binary_shape),
    sizeof(libxsmm_meltw_ternary_shape), sizeof(libxsmm_spgemm_config), sizeof(libxsmm_kernel_info),
    sizeof(libxsmm_mmkernel_info), sizeof(libxsmm_descriptor_blob),
    sizeof(libxsmm_matdiff_info), offsetof(libxsmm_matdiff_info, rsq), offsetof(libxsmm_matdiff_info, v_ref), offsetof(libxsmm_matdiff_info, m),
    sizeof(libxsmm_meqn_param), sizeof(libxsmm_meqn_arg_shape), sizeof(libxsmm_matrix_arg_attributes), sizeof(libxsmm_meqn_op_metadata),
    sizeof(libxsmm_meltwkernel_info), sizeof(libxsmm_registry_info), offsetof(libxsmm_gemm_ext_param, d), offsetof(libxsmm_meqn_param, output) };
  size_t i; for (i = 0; i < sizeof(s) / sizeof(*s); ++i) printf("%zu\n", s[i]);
  return 0;
}
''')
    exe = tmp_path / "sizes"
    r = subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ours = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    assert ours == reference.struct_sizes(len(ours))


@pytest.mark.parametrize("header", ["libxsmm.h", "libxsmm_utils.h", "libxsmm_source.h", "libxsmm_macros.h", "libxsmm_math.h"])
@pytest.mark.parametrize("compiler,std", [("gcc", "-std=c99"), ("g++", "-std=c++11")])
def test_public_headers_are_clean_c99_and_cxx11(header, compiler, std, tmp_path):
    """A drop-in header is included by other people's C and C++ code: no warnings under -Wall -Wextra -pedantic."""
    import shutil
    import subprocess
    if not shutil.which(compiler):
        pytest.skip(f"{compiler} not installed")
    src = tmp_path / ("t.c" if compiler == "gcc" else "t.cpp")
    src.write_text(f"#include <{header}>\nint main(void) {{ return 0; }}\n")
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
    r = subprocess.run([compiler, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


# ---- input preparation: host code, runs without a GPU -------------------------------------------------------------------
def _mtx(tmp_path, rows, cols, entries, name="a.mtx"):
    path = tmp_path / name
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% comment\n")
        f.write(f"{rows} {cols} {len(entries)}\n")
        for r, c, v in entries:
            f.write(f"{r + 1} {c + 1} {v!r}\n")
    return str(path).encode()


@pytest.mark.parametrize("by_column", [0, 1])
@pytest.mark.parametrize("vt", ["F32", "F64"])
def test_mtx_reader_builds_csr_and_csc_from_unsorted_entries(tmp_path, by_column, vt):
    import ctypes as C
    import numpy as np
    from libxsmm_amd import capi
    from libxsmm_amd.capi import DT
    api = capi.load()
    rng = np.random.default_rng(4)
    rows, cols = 35, 20
    pos = rng.choice(rows * cols, size=90, replace=False)
    pos = pos[(pos // cols != 7) & (pos % cols != 3)]                    # an empty row and an empty column
    rng.shuffle(pos)                                                     # entries in no particular order
    ent = [(int(q // cols), int(q % cols), float(np.round(rng.random() - 0.5, 3))) for q in pos]
    path = _mtx(tmp_path, rows, cols, ent)
    ptr, idx, val = C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)(), C.c_void_p()
    r, c, n = C.c_uint(), C.c_uint(), C.c_uint()
    dt = DT.F32 if vt == "F32" else DT.F64
    assert api.hip_mtx_read(path, by_column, dt, C.byref(ptr), C.byref(idx), C.byref(val), C.byref(r), C.byref(c), C.byref(n)) == 0
    assert (r.value, c.value, n.value) == (rows, cols, len(ent))
    outer = cols if by_column else rows
    p = np.ctypeslib.as_array(ptr, (outer + 1,)).copy(); x = np.ctypeslib.as_array(idx, (n.value,)).copy()
    v = np.ctypeslib.as_array(C.cast(val, C.POINTER(C.c_float if vt == "F32" else C.c_double)), (n.value,)).copy()
    dense = np.zeros((rows, cols))
    for a, b, w in ent:
        dense[a, b] = w
    want = dense.T if by_column else dense
    assert p[0] == 0 and p[-1] == len(ent) and np.all(np.diff(p.astype(np.int64)) >= 0)
    got = np.zeros_like(want)
    for o in range(outer):
        seg = x[p[o]:p[o + 1]]
        assert np.all(np.diff(seg.astype(np.int64)) > 0)                # inner indices ascending, no duplicates
        got[o, seg] = v[p[o]:p[o + 1]]
    assert np.allclose(got, want, rtol=1e-6 if vt == "F32" else 0, atol=0)
    for q in (ptr, idx):
        api.free(C.cast(q, C.c_void_p))
    api.free(val)
    # failure modes: missing file, index outside the header's shape
    assert api.hip_mtx_read(b"/nonexistent.mtx", 0, dt, C.byref(ptr), C.byref(idx), C.byref(val), C.byref(r), C.byref(c), C.byref(n)) != 0
    bad = _mtx(tmp_path, 3, 3, [(0, 0, 1.0), (5, 1, 2.0)], "bad.mtx")
    assert api.hip_mtx_read(bad, 0, dt, C.byref(ptr), C.byref(idx), C.byref(val), C.byref(r), C.byref(c), C.byref(n)) != 0
    # a header that claims more entries than the file can hold (round-2 advisor: reserve(n) threw through the C boundary), an over-long line
    huge = tmp_path / "huge.mtx"
    huge.write_text("%%MatrixMarket matrix coordinate real general\n3 3 4000000000\n1 1 1.0\n")
    assert api.hip_mtx_read(str(huge).encode(), 0, dt, C.byref(ptr), C.byref(idx), C.byref(val), C.byref(r), C.byref(c), C.byref(n)) != 0
    longline = tmp_path / "long.mtx"
    longline.write_text("%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 " + "0" * 600 + "1.0\n")
    assert api.hip_mtx_read(str(longline).encode(), 0, dt, C.byref(ptr), C.byref(idx), C.byref(val), C.byref(r), C.byref(c), C.byref(n)) != 0


@pytest.mark.parametrize("npdt,dtname", [("float32", "F32"), ("uint16", "BF16"), ("int8", "I8")])
def test_bcsc_builder_drops_zero_blocks_and_keeps_the_reference_layout(npdt, dtname):
    import ctypes as C
    import numpy as np
    from libxsmm_amd import capi
    from libxsmm_amd.capi import DT
    api = capi.load()
    rng = np.random.default_rng(5)
    K, N, bk, bn = 64, 48, 16, 8
    dense = rng.integers(1, 100, size=(N, K)).astype(npdt)              # B[n*K + k]
    keep = rng.random((N // bn, K // bk)) < 0.4
    keep[2, :] = False                                                   # an empty block column
    for nb in range(N // bn):
        for kb in range(K // bk):
            if not keep[nb, kb]:
                dense[nb * bn:(nb + 1) * bn, kb * bk:(kb + 1) * bk] = 0
    cp, ri, val, nn = C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)(), C.c_void_p(), C.c_uint()
    assert api.hip_bcsc_from_dense(getattr(DT, dtname), dense.ctypes.data, K, N, bk, bn, C.byref(cp), C.byref(ri), C.byref(val), C.byref(nn)) == 0
    assert nn.value == int(keep.sum())
    colptr = np.ctypeslib.as_array(cp, (N // bn + 1,)); rowidx = np.ctypeslib.as_array(ri, (max(1, nn.value),))
    vals = np.frombuffer((C.c_char * (nn.value * bk * bn * dense.itemsize)).from_address(val.value), dtype=npdt).reshape(nn.value, bn, bk)
    b = 0
    for nb in range(N // bn):
        assert colptr[nb] == b
        for kb in range(K // bk):
            if keep[nb, kb]:
                assert rowidx[b] == kb and np.array_equal(vals[b], dense[nb * bn:(nb + 1) * bn, kb * bk:(kb + 1) * bk])
                b += 1
    assert colptr[-1] == b
    assert api.hip_bcsc_from_dense(getattr(DT, dtname), dense.ctypes.data, K, N, 24, bn, C.byref(cp), C.byref(ri), C.byref(val), C.byref(nn)) != 0   # bk does not divide K


def test_launch_modes_round_trip_without_a_device():
    """0 blocking (the reference's semantics), 1 stream-ordered, 2 stream-ordered + coalescing (include/libxsmm_hip.h); anything else non-zero is 1."""
    api = capi.load()
    try:
        for want, got in ((0, 0), (1, 1), (2, 2), (7, 1), (0, 0)):
            api.hip_set_async(want)
            assert api.hip_get_async() == got
    finally:
        api.hip_set_async(0)
