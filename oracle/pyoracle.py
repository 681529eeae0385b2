"""ctypes access to the TEST-ONLY checkers: oracle/liboracle.so (the CPU restatement) and, when it
has been built, oracle/_ref/libxsmm_ref.so (the reference itself).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (libxsmm_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from libxsmm_amd import capi  # struct layouts and enums are shared with the product binding  # noqa: E402

ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libxsmm_ref.so")


class GemmDesc(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("k", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("a_type", C.c_int), ("b_type", C.c_int), ("c_type", C.c_int), ("comp_type", C.c_int),
                ("flags", C.c_uint), ("br_stride_a", C.c_longlong), ("br_stride_b", C.c_longlong),
                ("colbias", C.c_int), ("act", C.c_int)]


class MeltwDesc(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("ldi", C.c_int), ("ldo", C.c_int), ("ldi2", C.c_int), ("ldi3", C.c_int),
                ("in0_type", C.c_int), ("in1_type", C.c_int), ("in2_type", C.c_int), ("comp_type", C.c_int), ("out_type", C.c_int),
                ("flags", C.c_uint), ("type", C.c_int), ("operation", C.c_int)]


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(ORACLE_SO)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    ref_src = os.environ.get("LIBXSMM_REFERENCE", "/root/reference")
    if os.path.exists(os.path.join(ref_src, "include", "libxsmm_source.h")):
        if force or not os.path.exists(REF_SO) or os.path.getmtime(os.path.join(_HERE, "ref_shim.c")) > os.path.getmtime(REF_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref", f"REFERENCE={ref_src}"])
        # the reference's sample drivers against this repository's headers + libxsmm_amd.so (tests/test_reference_drivers_gpu.py)
        subprocess.check_call(["make", "-s", "-C", _HERE, "drivers", f"REFERENCE={ref_src}"])


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        L = self.lib
        L.oracle_gemm.argtypes = [vp, C.POINTER(GemmDesc)]; L.oracle_gemm.restype = None
        L.oracle_gemm_f32_fma.argtypes = [vp, C.POINTER(GemmDesc)]; L.oracle_gemm_f32_fma.restype = None
        for n in ("oracle_meltw_unary", "oracle_meltw_binary", "oracle_meltw_ternary"):
            getattr(L, n).argtypes = [vp, C.POINTER(MeltwDesc)]; getattr(L, n).restype = None
        i = C.c_int
        L.oracle_packed_spgemm_csr_asparse.argtypes = [i, i, i, i, i, vp, vp, vp, vp, i, vp, i, i]
        L.oracle_packed_spgemm_csc_bsparse.argtypes = [i, i, i, i, i, vp, vp, vp, vp, i, vp, i, i]
        L.oracle_packed_spgemm_csr_bsparse.argtypes = [i, i, i, i, i, vp, vp, vp, vp, i, vp, i, i]
        L.oracle_packed_spgemm_bcsc.argtypes = [i, i, i, i, i, i, i, i, i, vp, vp, vp, vp, vp, i]
        L.oracle_fsspmdm.argtypes = [i, i, i, i, vp, vp, vp, vp, i, vp, i, i]
        L.oracle_packed_spgemm_csc_csparse.argtypes = [i, i, i, vp, vp, vp, i, vp, i, vp, i]; L.oracle_packed_spgemm_csc_csparse.restype = None
        for n in ("oracle_packed_gemm", "oracle_packed_gemm_ac_rm", "oracle_packed_gemm_bc_rm"):
            getattr(L, n).argtypes = [i, i, i, i, i, vp, i, vp, i, vp, i, i]; getattr(L, n).restype = None
        for n in ("oracle_packed_spgemm_csr_asparse", "oracle_packed_spgemm_csc_bsparse", "oracle_packed_spgemm_csr_bsparse",
                  "oracle_packed_spgemm_bcsc", "oracle_fsspmdm"):
            getattr(L, n).restype = None
        L.oracle_f32_to_bf16_rne.argtypes = [C.c_float]; L.oracle_f32_to_bf16_rne.restype = C.c_ushort
        L.oracle_f32_to_bf16_trunc.argtypes = [C.c_float]; L.oracle_f32_to_bf16_trunc.restype = C.c_ushort
        L.oracle_bf16_to_f32.argtypes = [C.c_ushort]; L.oracle_bf16_to_f32.restype = C.c_float
        L.oracle_normf_rel.argtypes = [i, C.c_longlong, vp, vp]; L.oracle_normf_rel.restype = C.c_double

    def gemm(self, param, desc: GemmDesc, fma: bool = False):
        (self.lib.oracle_gemm_f32_fma if fma else self.lib.oracle_gemm)(C.byref(param), C.byref(desc))

    def meltw(self, param, desc: MeltwDesc):
        fn = {capi.UnaryParam: self.lib.oracle_meltw_unary, capi.BinaryParam: self.lib.oracle_meltw_binary,
              capi.TernaryParam: self.lib.oracle_meltw_ternary}[type(param)]
        fn(C.byref(param), C.byref(desc))


_oracle = None


def oracle() -> Oracle:
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def have_reference() -> bool:
    return os.path.exists(REF_SO)


_ref = None


class Reference(capi.Api):
    """The real reference behind the same binding as the product library (prefix xref_)."""

    def __init__(self):
        super().__init__(REF_SO, "xref_")
        L, vp = self.lib, C.c_void_p
        L.xref_struct_sizes.argtypes = [C.POINTER(C.c_size_t), C.c_int]
        L.xref_reference_gemm.argtypes = [vp, capi.GemmShape, C.c_uint, C.c_uint, capi.BrConfig]; L.xref_reference_gemm.restype = C.c_int
        L.xref_reference_gemm_ext.argtypes = [vp, capi.GemmShape, C.c_uint, C.c_uint, capi.BrConfig, capi.ExtUnaryArgops, capi.ExtBinaryPostops]
        L.xref_reference_gemm_ext.restype = C.c_int
        L.xref_reference_meltw_unary.argtypes = [vp, C.c_int, capi.UnaryShape, C.c_uint]
        L.xref_reference_meltw_binary.argtypes = [vp, C.c_int, capi.BinaryShape, C.c_uint]
        L.xref_reference_meltw_ternary.argtypes = [vp, C.c_int, capi.TernaryShape, C.c_uint]
        L.xref_convert_f32_to_bf16_rne.argtypes = [C.c_float]; L.xref_convert_f32_to_bf16_rne.restype = C.c_ushort
        L.xref_convert_f32_to_bf16_truncate.argtypes = [C.c_float]; L.xref_convert_f32_to_bf16_truncate.restype = C.c_ushort
        for name, arg, res in (("f32_to_f16", C.c_float, C.c_ushort), ("f16_to_f32", C.c_ushort, C.c_float), ("f32_to_bf8_rne", C.c_float, C.c_ubyte),
                               ("f16_to_hf8_rne", C.c_ushort, C.c_ubyte), ("f32_to_hf8_rne", C.c_float, C.c_ubyte),
                               ("bf8_to_f32", C.c_ubyte, C.c_float), ("hf8_to_f32", C.c_ubyte, C.c_float)):
            fn = getattr(L, "xref_convert_" + name); fn.argtypes = [arg]; fn.restype = res
        L.xref_convert_f32_to_bf8_stochastic.argtypes = [C.c_float, C.c_uint]; L.xref_convert_f32_to_bf8_stochastic.restype = C.c_ubyte
        L.xref_matdiff_normf_rel.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]; L.xref_matdiff_normf_rel.restype = C.c_double
        L.xref_time_gemm_batch.argtypes = [vp, C.POINTER(capi.GemmParam), C.c_size_t, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int]
        L.xref_time_gemm_batch.restype = C.c_double
        L.xref_time_gemm_ext_batch.argtypes = [vp, C.POINTER(capi.GemmExtParam), C.c_size_t, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int]
        L.xref_time_gemm_ext_batch.restype = C.c_double
        L.xref_time_fsspmdm.argtypes = [vp, vp, vp, C.c_int]
        L.xref_time_fsspmdm.restype = C.c_double
        L.xref_get_target_arch.restype = C.c_char_p
        self.init()

    def struct_sizes(self, n: int):
        out = (C.c_size_t * n)()
        self.lib.xref_struct_sizes(out, n)
        return list(out)


def reference() -> Reference:
    global _ref
    if _ref is None:
        _ref = Reference()
    return _ref
