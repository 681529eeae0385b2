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
