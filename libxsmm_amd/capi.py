"""ctypes mirror of include/libxsmm.h -- the host-side binding of the C-ABI library.

This is the Python counterpart of what a C caller of the reference writes: the same
struct layouts [ref: include/libxsmm_typedefs.h:570-773], the same dispatch functions
[ref: include/libxsmm.h:125-229], the same by-value shape/config arguments.  Enumerations
are not retyped here: they are parsed out of the X-tables of include/libxsmm.h so that the
header stays the single source of truth.

The binding is deliberately library-agnostic: `Api(path, prefix)` binds any shared object
that exports the LIBXSMM entry points under `prefix` -- the product library
(libxsmm_amd.so, prefix "libxsmm_"); the test-suite re-uses the class to drive the reference
build through the same structs (a different prefix, from test code only).  There is NO fallback:
if libxsmm_amd.so is missing `load()` raises, it never substitutes a CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "libxsmm.h")
LIB_PATH = os.path.join(_HERE, "lib", "libxsmm_amd.so")


# --------------------------------------------------------------------------------------
# enumerations from the header's X-tables
# --------------------------------------------------------------------------------------
def _parse_tables(path: str) -> Dict[str, Dict[str, int]]:
    text = open(path).read().replace("\\\n", " ")
    tables: Dict[str, Dict[str, int]] = {}
    for m in re.finditer(r"#define\s+(LIBXSMM_\w+_TABLE)\(X\)\s+(.*)", text):
        entries = {}
        for e in re.finditer(r"X\(\s*(\w+)\s*,\s*([^)]+)\)", m.group(2)):
            entries[e.group(1)] = int(eval(e.group(2), {"__builtins__": {}}))  # "32 | 16" style literals
        tables[m.group(1)] = entries
    return tables


_T = _parse_tables(HEADER)


class _Enum:
    def __init__(self, table: Dict[str, int], enumerate_values: bool = False):
        items = {k: i for i, k in enumerate(table)} if enumerate_values else dict(table)
        self._items = items
        for k, v in items.items():
            setattr(self, k, v)

    def name(self, value: int) -> str:
        for k, v in self._items.items():
            if v == value:
                return k
        return str(value)

    def __getitem__(self, key):
        return self._items[key]


DT = _Enum(_T["LIBXSMM_DATATYPE_TABLE"], enumerate_values=True)          # libxsmm_datatype
DT_SIZE = {DT[k]: v for k, v in _T["LIBXSMM_DATATYPE_TABLE"].items()}
GEMM_FLAG = _Enum(_T["LIBXSMM_GEMM_FLAG_TABLE"])
UNARY = _Enum(_T["LIBXSMM_MELTW_UNARY_TABLE"])
UNARY_FLAG = _Enum(_T["LIBXSMM_MELTW_UNARY_FLAG_TABLE"])
BINARY = _Enum(_T["LIBXSMM_MELTW_BINARY_TABLE"])
BINARY_FLAG = _Enum(_T["LIBXSMM_MELTW_BINARY_FLAG_TABLE"])
TERNARY = _Enum(_T["LIBXSMM_MELTW_TERNARY_TABLE"])
TERNARY_FLAG = _Enum(_T["LIBXSMM_MELTW_TERNARY_FLAG_TABLE"])
BR_NONE, BR_ADDRESS, BR_OFFSET, BR_STRIDE = 0, 1, 2, 4                  # libxsmm_gemm_batch_reduce_type


# --------------------------------------------------------------------------------------
# structs
# --------------------------------------------------------------------------------------
class MatrixArg(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("primary", "secondary", "tertiary", "quaternary", "quinary", "senary")]


class MatrixOpArg(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("primary", "secondary", "tertiary", "quaternary")]


class MeqnParam(C.Structure):
    _fields_ = [("ops_args", C.POINTER(MatrixOpArg)), ("inputs", C.POINTER(MatrixArg)), ("output", MatrixArg)]


class GemmParam(C.Structure):
    _fields_ = [("op", MatrixOpArg), ("a", MatrixArg), ("b", MatrixArg), ("c", MatrixArg)]


class GemmExtParam(C.Structure):
    _fields_ = [("op", MatrixOpArg), ("a", MatrixArg), ("b", MatrixArg), ("c", MatrixArg),
                ("d", MatrixArg), ("ap", MatrixArg), ("bp", MatrixArg), ("cp", MatrixArg)]


class UnaryParam(C.Structure):
    _fields_ = [("op", MatrixOpArg), ("in_", MatrixArg), ("out", MatrixArg)]


class BinaryParam(C.Structure):
    _fields_ = [("op", MatrixOpArg), ("in0", MatrixArg), ("in1", MatrixArg), ("out", MatrixArg)]


class TernaryParam(C.Structure):
    _fields_ = [("op", MatrixOpArg), ("in0", MatrixArg), ("in1", MatrixArg), ("in2", MatrixArg), ("out", MatrixArg)]


class GemmShape(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("k", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("a_in_type", C.c_int), ("b_in_type", C.c_int), ("out_type", C.c_int), ("comp_type", C.c_int)]


class BrConfig(C.Structure):
    _fields_ = [("br_type", C.c_int), ("br_stride_a_hint", C.c_int), ("br_stride_b_hint", C.c_int), ("br_unroll_hint", C.c_ubyte)]


class SpgemmConfig(C.Structure):
    _fields_ = [("packed_width", C.c_int), ("bk", C.c_int), ("bn", C.c_int)]


class ExtUnaryArgops(C.Structure):
    _fields_ = [("ldap", C.c_int), ("ap_unary_type", C.c_int), ("ap_unary_flags", C.c_uint), ("store_ap", C.c_int),
                ("ldbp", C.c_int), ("bp_unary_type", C.c_int), ("bp_unary_flags", C.c_uint), ("store_bp", C.c_int),
                ("ldcp", C.c_int), ("cp_unary_type", C.c_int), ("cp_unary_flags", C.c_uint), ("store_cp", C.c_int)]


class ExtBinaryPostops(C.Structure):
    _fields_ = [("ldd", C.c_int), ("d_in_type", C.c_int), ("d_binary_type", C.c_int), ("d_binary_flags", C.c_uint)]


class MeqnArgShape(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("ld", C.c_int), ("type", C.c_int)]


class MatrixArgAttributes(C.Structure):
    _fields_ = [("type", C.c_int), ("set_type", C.c_int), ("set_cardinality_hint", C.c_int), ("set_stride_hint", C.c_int)]


class MeqnMetadata(C.Structure):      # libxsmm_meqn_op_metadata and libxsmm_meqn_arg_metadata share this layout
    _fields_ = [("eqn_idx", C.c_int), ("pos", C.c_int)]


class UnaryShape(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("ldi", C.c_int), ("ldo", C.c_int),
                ("in0_type", C.c_int), ("out_type", C.c_int), ("comp_type", C.c_int)]


class BinaryShape(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("ldi", C.c_int), ("ldi2", C.c_int), ("ldo", C.c_int),
                ("in0_type", C.c_int), ("in1_type", C.c_int), ("out_type", C.c_int), ("comp_type", C.c_int)]


class TernaryShape(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("ldi", C.c_int), ("ldi2", C.c_int), ("ldi3", C.c_int), ("ldo", C.c_int),
                ("in0_type", C.c_int), ("in1_type", C.c_int), ("in2_type", C.c_int), ("out_type", C.c_int), ("comp_type", C.c_int)]


class HipShard(C.Structure):           # libxsmm_hip_shard (include/libxsmm_hip.h): what ONE device does in a multi-device launch
    _fields_ = [("device", C.c_int), ("kernel", C.c_void_p), ("param", C.c_void_p), ("count", C.c_size_t), ("stride", C.c_longlong * 5),
                ("gather_src", C.c_void_p), ("gather_bytes", C.c_size_t), ("gather_dst_offset", C.c_size_t),
                ("gather_rows", C.c_size_t), ("gather_src_pitch", C.c_size_t), ("gather_dst_pitch", C.c_size_t)]


class KernelInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("nflops", C.c_uint), ("code_size", C.c_size_t), ("is_reference_kernel", C.c_uint)]


class MmKernelInfo(C.Structure):
    _fields_ = [("iprecision", C.c_int), ("oprecision", C.c_int), ("prefetch", C.c_int),
                ("lda", C.c_uint), ("ldb", C.c_uint), ("ldc", C.c_uint), ("m", C.c_uint), ("n", C.c_uint), ("k", C.c_uint), ("flags", C.c_int)]


# the order matches xref_struct_sizes() in oracle/ref_shim.c (layout check against the reference)
LAYOUT_PROBE = [GemmParam, GemmExtParam, MatrixArg, MatrixOpArg, UnaryParam, BinaryParam, TernaryParam, GemmShape, BrConfig,
                ExtUnaryArgops, ExtBinaryPostops, UnaryShape, BinaryShape, TernaryShape, SpgemmConfig, KernelInfo, MmKernelInfo]

GEMM_FN = C.CFUNCTYPE(None, C.POINTER(GemmParam))
GEMM_EXT_FN = C.CFUNCTYPE(None, C.POINTER(GemmExtParam))
UNARY_FN = C.CFUNCTYPE(None, C.POINTER(UnaryParam))
BINARY_FN = C.CFUNCTYPE(None, C.POINTER(BinaryParam))
TERNARY_FN = C.CFUNCTYPE(None, C.POINTER(TernaryParam))
MEQN_FN = C.CFUNCTYPE(None, C.POINTER(MeqnParam))

# every symbol include/libxsmm.h and libxsmm_hip.h declare (checked by tests/test_capi_symbols.py)
_DECL_RE = re.compile(r"LIBXSMM_API\s+[^;(]*?\b(libxsmm_\w+)\s*\(")


def declared_symbols():
    names = []
    for h in ("libxsmm.h", "libxsmm_hip.h", "libxsmm_utils.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        names += _DECL_RE.findall(text)
    return sorted(set(names))


# --------------------------------------------------------------------------------------
# binding
# --------------------------------------------------------------------------------------
class Api:
    """Binds the dispatch/param entry points of one shared library."""

    def __init__(self, path: str, prefix: str = "libxsmm_"):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self._bind()

    def _f(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype, fn.argtypes = restype, argtypes
        return fn

    def has(self, name: str) -> bool:
        return hasattr(self.lib, self.prefix + name)

    def _bind(self):
        f = self._f
        vp = C.c_void_p
        self.init = f("init", None, [])
        self.finalize = f("finalize", None, [])
        self.dispatch_gemm = f("dispatch_gemm", vp, [GemmShape, C.c_uint, C.c_uint])
        self.dispatch_brgemm = f("dispatch_brgemm", vp, [GemmShape, C.c_uint, C.c_uint, BrConfig])
        self.dispatch_brgemm_ext = f("dispatch_brgemm_ext", vp, [GemmShape, C.c_uint, C.c_uint, BrConfig, ExtUnaryArgops, ExtBinaryPostops])
        self.dispatch_meltw_unary = f("dispatch_meltw_unary", vp, [C.c_int, UnaryShape, C.c_uint])
        self.dispatch_meltw_binary = f("dispatch_meltw_binary", vp, [C.c_int, BinaryShape, C.c_uint])
        self.dispatch_meltw_ternary = f("dispatch_meltw_ternary", vp, [C.c_int, TernaryShape, C.c_uint])
        ip = C.POINTER(C.c_int)
        self.sgemm = f("sgemm", None, [C.c_char_p, C.c_char_p, ip, ip, ip, vp, vp, ip, vp, ip, vp, vp, ip])
        self.dgemm = f("dgemm", None, [C.c_char_p, C.c_char_p, ip, ip, ip, vp, vp, ip, vp, ip, vp, vp, ip])
        # matrix equations
        self.meqn_create = f("meqn_create", C.c_int, [])
        self.meqn_push_back_arg = f("meqn_push_back_arg", C.c_int, [MeqnMetadata, MeqnArgShape, MatrixArgAttributes])
        self.meqn_push_back_unary_op = f("meqn_push_back_unary_op", C.c_int, [MeqnMetadata, C.c_int, C.c_int, C.c_uint])
        self.meqn_push_back_binary_op = f("meqn_push_back_binary_op", C.c_int, [MeqnMetadata, C.c_int, C.c_int, C.c_uint])
        self.meqn_push_back_ternary_op = f("meqn_push_back_ternary_op", C.c_int, [MeqnMetadata, C.c_int, C.c_int, C.c_uint])
        self.dispatch_meqn = f("dispatch_meqn", vp, [C.c_int, MeqnArgShape])
        self.create_packed_spgemm_csr = f("create_packed_spgemm_csr", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int, vp, vp, vp])
        self.create_packed_spgemm_csc = f("create_packed_spgemm_csc", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int, vp, vp, vp])
        self.create_packed_gemm = f("create_packed_gemm", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int])
        self.create_packed_gemm_ac_rm = f("create_packed_gemm_ac_rm", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int])
        self.create_packed_gemm_bc_rm = f("create_packed_gemm_bc_rm", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int])
        self.create_packed_spgemm_bcsc = f("create_packed_spgemm_bcsc", vp, [GemmShape, C.c_uint, C.c_uint, SpgemmConfig])
        self.release_kernel = f("release_kernel", None, [vp])
        self.get_kernel_info = f("get_kernel_info", C.c_int, [vp, C.POINTER(KernelInfo)])
        self.fsspmdm_create = f("fsspmdm_create", vp, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp])
        self.fsspmdm_execute = f("fsspmdm_execute", None, [vp, vp, vp])
        self.fsspmdm_destroy = f("fsspmdm_destroy", None, [vp])
        if self.prefix == "libxsmm_":
            self.dispatch_tilecfg_gemm = f("dispatch_tilecfg_gemm", vp, [GemmShape, C.c_uint])
            self.create_spgemm_csr_areg = f("create_spgemm_csr_areg", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int, vp, vp, vp])
            self.get_mmkernel_info = f("get_mmkernel_info", C.c_int, [vp, C.POINTER(MmKernelInfo)])
            self.hip_available = f("hip_available", C.c_int, [])
            self.hip_device_count = f("hip_device_count", C.c_int, [])
            self.hip_set_device = f("hip_set_device", C.c_int, [C.c_int])
            self.hip_set_stream = f("hip_set_stream", None, [vp])
            self.hip_set_async = f("hip_set_async", None, [C.c_int])
            self.hip_get_async = f("hip_get_async", C.c_int, [])
            self.hip_set_streaming_hint = f("hip_set_streaming_hint", None, [C.c_int])
            self.hip_get_streaming_hint = f("hip_get_streaming_hint", C.c_int, [])
            self.hip_streaming_window_verdict = f("hip_streaming_window_verdict", C.c_int, [])
            self.hip_sync = f("hip_sync", None, [])
            self.hip_get_last_error = f("hip_get_last_error", C.c_int, [])
            self.hip_get_last_error_string = f("hip_get_last_error_string", C.c_char_p, [])
            self.hip_clear_last_error = f("hip_clear_last_error", None, [])
            self.hip_kernel_name = f("hip_kernel_name", C.c_char_p, [vp, C.c_int])
            self.hip_launch_count = f("hip_launch_count", C.c_ulonglong, [C.c_int])
            self.hip_bcsc_bind_pattern = f("hip_bcsc_bind_pattern", C.c_int, [vp, vp, vp, C.c_ulonglong])
            self.hip_pipeline_begin = f("hip_pipeline_begin", C.c_int, [C.c_int])
            self.hip_pipeline_end = f("hip_pipeline_end", C.c_int, [])
            self.hip_probe_mfma = f("hip_probe_mfma", C.c_int, [C.c_int, vp, C.c_int, C.POINTER(C.c_double)])
            self.hip_set_jit = f("hip_set_jit", None, [C.c_int])
            self.hip_get_jit = f("hip_get_jit", C.c_int, [])
            ll = C.c_longlong
            self.hip_gemm_batch_strided = f("hip_gemm_batch_strided", None, [vp, C.POINTER(GemmParam), C.c_size_t, ll, ll, ll])
            self.hip_gemm_ext_batch_strided = f("hip_gemm_ext_batch_strided", None, [vp, C.POINTER(GemmExtParam), C.c_size_t, ll, ll, ll, ll, ll])
            self.hip_gemm_batch_strided_2d = f("hip_gemm_batch_strided_2d", None, [vp, C.POINTER(GemmParam), C.c_size_t, C.c_size_t, ll, ll, ll, ll])
            self.hip_gemm_ext_batch_strided_2d = f("hip_gemm_ext_batch_strided_2d", None, [vp, C.POINTER(GemmExtParam), C.c_size_t, C.c_size_t, ll, ll, ll, ll, ll, ll, ll])
            self.hip_gemm_batch_pointers = f("hip_gemm_batch_pointers", None, [vp, C.POINTER(GemmParam), C.c_size_t, vp, vp, vp])
            self.hip_meltw_unary_batch_strided = f("hip_meltw_unary_batch_strided", None, [vp, C.POINTER(UnaryParam), C.c_size_t, ll, ll, ll])
            self.hip_meltw_binary_batch_strided = f("hip_meltw_binary_batch_strided", None, [vp, C.POINTER(BinaryParam), C.c_size_t, ll, ll, ll])
            self.hip_meltw_ternary_batch_strided = f("hip_meltw_ternary_batch_strided", None, [vp, C.POINTER(TernaryParam), C.c_size_t, ll, ll, ll, ll])
            pu = C.POINTER(C.POINTER(C.c_uint))
            self.hip_mtx_read = f("hip_mtx_read", C.c_int, [C.c_char_p, C.c_int, C.c_int, pu, pu, C.POINTER(vp), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)])
            self.hip_bcsc_from_dense = f("hip_bcsc_from_dense", C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, pu, pu, C.POINTER(vp), C.POINTER(C.c_uint)])
            self.free = f("free", None, [vp])
            self.hip_ipc_export = f("hip_ipc_export", C.c_int, [vp, vp])
            self.hip_gather_shards = f("hip_gather_shards", C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp])
            self.hip_malloc = f("hip_malloc", vp, [C.c_size_t])
            self.hip_free = f("hip_free", None, [vp])
            self.hip_memcpy_h2d = f("hip_memcpy_h2d", C.c_int, [vp, vp, C.c_size_t])
            self.hip_memcpy_d2h = f("hip_memcpy_d2h", C.c_int, [vp, vp, C.c_size_t])
            self.hip_memset = f("hip_memset", C.c_int, [vp, C.c_int, C.c_size_t])
            self.hip_launch_shards = f("hip_launch_shards", C.c_int, [C.POINTER(HipShard), C.c_int, C.c_int, vp])
            self.hip_gemm_batch_strided_sharded = f("hip_gemm_batch_strided_sharded", C.c_int, [vp, C.POINTER(GemmParam), C.c_size_t, ll, ll, ll, C.c_int, C.POINTER(C.c_int), C.c_int, vp])
            self.hip_gemm_ext_batch_strided_sharded = f("hip_gemm_ext_batch_strided_sharded", C.c_int,
                                                        [vp, C.POINTER(GemmExtParam), C.c_size_t, ll, ll, ll, ll, ll, C.c_int, C.POINTER(C.c_int), C.c_int, vp])
            pi = C.POINTER(C.c_int)
            self.hip_create_packed_spgemm_csr_sharded = f("hip_create_packed_spgemm_csr_sharded", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int, vp, vp, vp, C.c_int, pi])
            self.hip_create_packed_spgemm_csc_sharded = f("hip_create_packed_spgemm_csc_sharded", vp, [GemmShape, C.c_uint, C.c_uint, C.c_int, vp, vp, vp, C.c_int, pi])
            self.hip_create_packed_spgemm_bcsc_sharded = f("hip_create_packed_spgemm_bcsc_sharded", vp, [GemmShape, C.c_uint, C.c_uint, SpgemmConfig, C.c_int, pi])
            self.hip_fsspmdm_create_sharded = f("hip_fsspmdm_create_sharded", vp, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, pi])
            self.hip_sharded_count = f("hip_sharded_count", C.c_int, [vp])
            self.hip_sharded_range = f("hip_sharded_range", C.c_int, [vp, C.c_int, pi, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)])
            self.hip_sharded_handle = f("hip_sharded_handle", vp, [vp, C.c_int])
            self.hip_sharded_launch = f("hip_sharded_launch", C.c_int, [vp, C.POINTER(GemmParam), C.c_int, vp, C.c_size_t])
            self.hip_sharded_destroy = f("hip_sharded_destroy", None, [vp])
            self.hip_shard_range = f("hip_shard_range", None, [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)])

    # ---- calling a handle (a plain C function pointer) --------------------------------
    @staticmethod
    def call(handle: int, param, fntype=None):
        if not handle:
            raise RuntimeError("NULL kernel handle (dispatch refused the descriptor or no HIP device is present)")
        if fntype is None:
            fntype = {GemmParam: GEMM_FN, GemmExtParam: GEMM_EXT_FN, UnaryParam: UNARY_FN, BinaryParam: BINARY_FN, TernaryParam: TERNARY_FN, MeqnParam: MEQN_FN}[type(param)]
        fntype(handle)(C.byref(param))

    def check(self):
        """Raise if the calling thread's sticky error state is set (kernels have no error channel)."""
        if self.prefix == "libxsmm_" and self.hip_get_last_error() != 0:
            msg = self.hip_get_last_error_string().decode()
            self.hip_clear_last_error()
            raise RuntimeError("libxsmm_amd: " + msg)


_api = None


def load() -> Api:
    """The product library.  Raises when it has not been built -- never falls back."""
    global _api
    if _api is None:
        _api = Api(LIB_PATH, "libxsmm_")
    return _api


# --------------------------------------------------------------------------------------
# small constructors
# --------------------------------------------------------------------------------------
def gemm_shape(m, n, k, lda, ldb, ldc, a_type, b_type, out_type, comp_type) -> GemmShape:
    return GemmShape(m, n, k, lda, ldb, ldc, a_type, b_type, out_type, comp_type)


def br_config(br_type=BR_NONE, stride_a=0, stride_b=0, unroll=0) -> BrConfig:
    return BrConfig(br_type, stride_a, stride_b, unroll)


def no_argops() -> ExtUnaryArgops:
    return ExtUnaryArgops()


def argops_cp(ldc: int, cp_type: int, cp_flags: int = 0) -> ExtUnaryArgops:
    a = ExtUnaryArgops()
    a.ldcp, a.cp_unary_type, a.cp_unary_flags = ldc, cp_type, cp_flags
    return a


def postops_colbias(ldd: int, d_type: int) -> ExtBinaryPostops:
    return ExtBinaryPostops(ldd, d_type, BINARY.ADD, BINARY_FLAG.BCAST_COL_IN_0)


def no_postops() -> ExtBinaryPostops:
    return ExtBinaryPostops(0, DT.F32, BINARY.NONE, 0)
