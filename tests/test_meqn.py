"""Matrix equations: trees of TPPs behind one handle.

Oracle = the restatement of src/generator_matequation_reference_impl.c:105-227 as a bottom-up composition of the pinned
TPP restatements (oracle/oracle_meltw.c): every op node is the TPP of its type, comp/out type = the op's dtype,
intermediate shapes per src/libxsmm_matrixeqn.c:869-936, the root writes out_shape.  It is pinned against the reference's
own libxsmm_dispatch_meqn (CPU JIT, oracle/_ref) in the CPU tests; the GPU tests compare the product with it."""
import ctypes as C

import numpy as np
import pytest

from helpers import normf_rel, rand_values
from libxsmm_amd import capi
from libxsmm_amd.capi import BINARY, BINARY_FLAG, DT, GEMM_FLAG, TERNARY, TERNARY_FLAG, UNARY, UNARY_FLAG
from oracle import pyoracle

NPDT = {DT.F32: np.float32, DT.BF16: np.uint16}
SINGULAR = capi.MatrixArgAttributes(0, 0, 0, 0)
ALPHA = C.c_float(0.125)


# tree notation: ("arg", pos) | ("u", type, flags, child) | ("b", type, flags, l, r) | ("t", type, flags, a, b, c) | ("mm", BINARY.MATMUL, 0, A, B)
def build(api, tree, arg_shapes, comp=DT.F32):
    idx = api.meqn_create()

    def walk(t):
        if t[0] == "arg":
            m, n, ld, dt = arg_shapes[t[1]]
            assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, t[1]), capi.MeqnArgShape(m, n, ld, dt), SINGULAR) == 0
        elif t[0] == "u":
            assert api.meqn_push_back_unary_op(capi.MeqnMetadata(idx, 0 if t[1] == UNARY.LEAKY_RELU else -1), t[1], comp, t[2]) == 0
            walk(t[3])
        elif t[0] in ("b", "mm"):
            assert api.meqn_push_back_binary_op(capi.MeqnMetadata(idx, -1), t[1], comp, t[2]) == 0
            walk(t[3]); walk(t[4])
        else:
            assert api.meqn_push_back_ternary_op(capi.MeqnMetadata(idx, -1), t[1], comp, t[2]) == 0
            walk(t[3]); walk(t[4]); walk(t[5])
    walk(tree)
    return idx


REDUCES = (UNARY.REDUCE_X_OP_ADD, UNARY.REDUCE_X_OP_MAX, UNARY.REDUCE_X2_OP_ADD)


def evaluate(tree, arg_shapes, arrays, out_shape, comp=DT.F32):
    """Bottom-up composition of oracle TPPs; returns the output array (flat, ld*n elements of out type)."""
    orc = pyoracle.oracle()

    def ev(t, root):
        if t[0] == "arg":
            m, n, ld, dt = arg_shapes[t[1]]
            return arrays[t[1]], (m, n, ld, dt)
        kids = [ev(c, False) for c in t[3:]]
        (x0, (m0, n0, ld0, dt0)) = kids[0]
        if t[0] == "mm":       # A (m x k) times B (k x n) through the pinned GEMM restatement, beta = 0 (samples/equation/equation_matmul.c:37-62)
            (x1, (m1, n1, ld1, dt1)) = kids[1]
            assert n0 == m1
            ld, odt = (out_shape[2], out_shape[3]) if root else (m0, comp)
            out = np.zeros(ld * n1, dtype=NPDT[odt])
            p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = x0.ctypes.data, x1.ctypes.data, out.ctypes.data
            orc.gemm(p, pyoracle.GemmDesc(m0, n1, n0, ld0, ld1, ld, dt0, dt1, odt, DT.F32, GEMM_FLAG.BETA_0 | GEMM_FLAG.USE_XGEMM_ABI | (GEMM_FLAG.VNNI_A if t[1] == BINARY.MATMUL_A_VNNI else 0), 0, 0, 0, 0))
            return out, (m0, n1, ld, odt)
        if t[0] == "u":
            if t[1] in REDUCES:
                m, n = ((n0, 1) if t[2] & UNARY_FLAG.REDUCE_ROWS else (m0, 1)); dm, dn = m0, n0
            else:
                m, n = m0, n0; dm, dn = m0, n0
        elif t[0] == "b" and t[1] == BINARY.MUL_AND_REDUCE_TO_SCALAR_OP_ADD:
            m, n = 1, 1; dm, dn = m0, n0          # a dot product over the operands' extent [ref: mateltwise ref :2523-2542]
        else:
            m = max(k[1][0] for k in kids); n = max(k[1][1] for k in kids); dm, dn = m, n
        ld, odt = (out_shape[2], out_shape[3]) if root else (m, comp)
        out = np.zeros(ld * n, dtype=NPDT[odt])
        if t[0] == "u":
            p = capi.UnaryParam(); p.in_.primary, p.out.primary = x0.ctypes.data, out.ctypes.data
            if t[1] == UNARY.LEAKY_RELU:
                p.op.primary = C.addressof(ALPHA)
            d = pyoracle.MeltwDesc(dm, dn, ld0, ld, 0, 0, dt0, DT.UNSUPPORTED, DT.UNSUPPORTED, comp, odt, t[2], t[1], 1)
        elif t[0] == "b":
            (x1, (_, _, ld1, dt1)) = kids[1]
            p = capi.BinaryParam(); p.in0.primary, p.in1.primary, p.out.primary = x0.ctypes.data, x1.ctypes.data, out.ctypes.data
            d = pyoracle.MeltwDesc(dm, dn, ld0, ld, ld1, 0, dt0, dt1, DT.UNSUPPORTED, comp, odt, t[2], t[1], 2)
        else:
            (x1, (_, _, ld1, dt1)), (x2, (_, _, ld2, dt2)) = kids[1], kids[2]
            p = capi.TernaryParam(); p.in0.primary, p.in1.primary, p.in2.primary, p.out.primary = x0.ctypes.data, x1.ctypes.data, x2.ctypes.data, out.ctypes.data
            d = pyoracle.MeltwDesc(dm, dn, ld0, ld, ld1, ld2, dt0, dt1, dt2, comp, odt, t[2], t[1], 3)
        orc.meltw(p, d)
        return out, (m, n, ld, odt)
    return ev(tree, True)[0]


M, N, LD = 40, 24, 48
A = lambda i: ("arg", i)
CASES = {
    # equation_simple.c:516-538: (a0 + inc(a1)) * (x2(a2) + a3)
    "simple": (("b", BINARY.MUL, 0, ("b", BINARY.ADD, 0, A(0), ("u", UNARY.INC, 0, A(1))), ("b", BINARY.ADD, 0, ("u", UNARY.X2, 0, A(2)), A(3))),
               [(M, N, LD, DT.F32)] * 4, (M, N, LD, DT.F32)),
    # bias + ReLU chain of config #5, un-fused from the GEMM: relu(bias (column vector) + x), bf16 in and out
    "bias_relu_bf16": (("u", UNARY.RELU, 0, ("b", BINARY.ADD, BINARY_FLAG.BCAST_COL_IN_0, A(0), A(1))),
                       [(64, 1, 64, DT.BF16), (64, 64, 64, DT.BF16)], (64, 64, 64, DT.BF16)),
    # a column reduction feeding a broadcast: x * rsqrt-free scale = x * (sum over columns of x^2) broadcast back
    "reduce_bcast": (("b", BINARY.MUL, BINARY_FLAG.BCAST_COL_IN_1, A(0), ("u", UNARY.REDUCE_X2_OP_ADD, UNARY_FLAG.REDUCE_COLS, A(0))),
                     [(M, N, LD, DT.F32)], (M, N, M, DT.F32)),
    "ternary_muladd": (("t", TERNARY.MULADD, 0, A(0), ("u", UNARY.NEGATE, 0, A(1)), A(2)),
                       [(M, N, LD, DT.F32), (M, N, M, DT.F32), (M, N, LD, DT.F32)], (M, N, LD, DT.F32)),
    "tanh_sigmoid_chain": (("u", UNARY.TANH, 0, ("b", BINARY.MUL, 0, ("u", UNARY.SIGMOID, 0, A(0)), A(1))), [(M, N, LD, DT.F32), (M, N, LD, DT.F32)], (M, N, LD, DT.F32)),
    # equation_simple_layernorm.c:85-106: ((x * s + b) * gamma + beta) with the mean/variance factors as broadcast scalars
    "layernorm_affine": (("t", TERNARY.MULADD, TERNARY_FLAG.REUSE_IN_2_AS_OUT,
                          ("t", TERNARY.MULADD, TERNARY_FLAG.BCAST_SCALAR_IN_1 | TERNARY_FLAG.BCAST_SCALAR_IN_2 | TERNARY_FLAG.REUSE_IN_2_AS_OUT, A(0), A(1), A(2)), A(3), A(4)),
                         [(64, 32, 64, DT.BF16), (1, 1, 1, DT.F32), (1, 1, 1, DT.F32), (64, 32, 64, DT.BF16), (64, 32, 64, DT.BF16)], (64, 32, 64, DT.BF16)),
    # equation 0 of equation_matmul.c without the transcendentals: (C x D) * (A + B), the product as a BINARY_MATMUL node
    "matmul_mul": (("b", BINARY.MUL, 0, ("mm", BINARY.MATMUL, 0, A(2), A(3)), ("b", BINARY.ADD, 0, A(0), A(1))),
                   [(48, 24, 56, DT.F32), (48, 24, 48, DT.F32), (48, 40, 52, DT.F32), (40, 24, 42, DT.F32)], (48, 24, 56, DT.F32)),
    # a bf16 product at the head, A in VNNI-2 layout (what the reference's AMX / AVX-512 bf16 kernels take), relu on top
    "matmul_vnni_bf16": (("u", UNARY.RELU, 0, ("mm", BINARY.MATMUL_A_VNNI, 0, A(0), A(1))),
                         [(32, 16, 32, DT.BF16), (16, 24, 16, DT.BF16)], (32, 24, 40, DT.BF16)),
    # equation_layernorm.c:966-998: the db and ds sums of the backward pass, dot products that end in one number
    "dot_to_scalar": (("b", BINARY.MUL_AND_REDUCE_TO_SCALAR_OP_ADD, 0, A(0), A(1)),
                      [(64, 12, 128, DT.BF16), (64, 12, 64, DT.BF16)], (1, 1, 1, DT.F32)),
    "mul_dot_to_scalar": (("b", BINARY.MUL_AND_REDUCE_TO_SCALAR_OP_ADD, 0, ("b", BINARY.MUL, 0, A(0), A(1)), A(2)),
                          [(M, N, LD, DT.F32), (M, N, M, DT.F32), (M, N, LD, DT.F32)], (1, 1, 1, DT.F32)),
    # equation_softmax.c:527-538 without the DUMP: exp(x - max x) / sum exp(x - max x), both reductions (COLS then ROWS) to one number inside the tree
    "softmax_fwd": (("b", BINARY.MUL, BINARY_FLAG.BCAST_SCALAR_IN_1,
                     ("u", UNARY.EXP, 0, ("b", BINARY.SUB, BINARY_FLAG.BCAST_SCALAR_IN_1, A(0), ("u", UNARY.REDUCE_X_OP_MAX, UNARY_FLAG.REDUCE_ROWS, ("u", UNARY.REDUCE_X_OP_MAX, UNARY_FLAG.REDUCE_COLS, A(0))))),
                     ("u", UNARY.RECIPROCAL, 0, ("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_ROWS, ("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_COLS,
                      ("u", UNARY.EXP, 0, ("b", BINARY.SUB, BINARY_FLAG.BCAST_SCALAR_IN_1, A(0), ("u", UNARY.REDUCE_X_OP_MAX, UNARY_FLAG.REDUCE_ROWS, ("u", UNARY.REDUCE_X_OP_MAX, UNARY_FLAG.REDUCE_COLS, A(0))))))))),
                    [(64, 12, 128, DT.BF16)], (64, 12, 64, DT.BF16)),
    # equation_softmax.c:676-688: a1 - (sum a0) * a0... as NMULADD with the sum broadcast into operand 0
    "softmax_bwd": (("t", TERNARY.NMULADD, TERNARY_FLAG.BCAST_SCALAR_IN_0 | TERNARY_FLAG.REUSE_IN_2_AS_OUT,
                     ("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_ROWS, ("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_COLS, A(0))), A(0), A(1)),
                    [(M, N, M, DT.F32), (M, N, LD, DT.F32)], (M, N, LD, DT.F32)),
    # the sum of squares of an expression as the head: one number out
    "sum_of_squares": (("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_ROWS, ("u", UNARY.REDUCE_X2_OP_ADD, UNARY_FLAG.REDUCE_COLS, ("b", BINARY.SUB, 0, A(0), A(1)))),
                       [(M, N, LD, DT.F32), (M, N, M, DT.F32)], (1, 1, 1, DT.F32)),
    # a product reduced to ONE number: sum over all elements of (A0 x A1) -- a GEMM node below a 1 x 1 head (round-2 advisor: the nested
    # GEMM call used to rewind the staging scratch that holds a host-resident 1 x 1 result)
    "matmul_sum_to_scalar": (("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_ROWS, ("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_COLS, ("mm", BINARY.MATMUL, 0, A(0), A(1)))),
                             [(48, 40, 52, DT.F32), (40, 24, 42, DT.F32)], (1, 1, 1, DT.F32)),
    "mixed_precision": (("b", BINARY.SUB, 0, ("u", UNARY.X2, 0, A(0)), ("b", BINARY.MUL, BINARY_FLAG.BCAST_SCALAR_IN_1, A(1), A(2))),
                        [(M, N, LD, DT.BF16), (M, N, M, DT.F32), (1, 1, 1, DT.F32)], (M, N, LD, DT.BF16)),
}


def _inputs(shapes, seed):
    rng = np.random.default_rng(seed)
    return [rand_values(rng, ld * n, dt) for (m, n, ld, dt) in shapes]


def _call(api, handle, arrays_ptrs, out_ptr):
    inputs = (capi.MatrixArg * len(arrays_ptrs))()
    for i, ptr in enumerate(arrays_ptrs):
        inputs[i].primary = ptr
    ops = (capi.MatrixOpArg * 1)()
    ops[0].primary = C.addressof(ALPHA)
    p = capi.MeqnParam()
    p.inputs, p.ops_args = inputs, ops
    p.output.primary = out_ptr
    capi.Api.call(handle, p)


def _valid(x, shape):
    m, n, ld, _ = shape
    return x.reshape(n, ld)[:, :m]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_composition_matches_reference_meqn(reference, name):
    if name == "matmul_vnni_bf16":
        pytest.skip("the reference's CPU JIT corrupts the heap on this bf16 MATMUL tree on this host (NaN output, glibc abort): GPU-vs-oracle only")
    tree, shapes, out_shape = CASES[name]
    arrays = _inputs(shapes, 5)
    mine = evaluate(tree, shapes, arrays, out_shape)
    idx = build(reference, tree, shapes)
    h = reference.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    if not h:
        pytest.skip("the reference JIT declines this equation on this host")
    theirs = np.zeros(out_shape[2] * out_shape[1], dtype=NPDT[out_shape[3]])
    _call(reference, h, [a.ctypes.data for a in arrays], theirs.ctypes.data)
    assert normf_rel(_valid(theirs, out_shape), _valid(mine, out_shape), out_shape[3]) <= ((1e-3 if name == "tanh_sigmoid_chain" else 1e-6) if out_shape[3] == DT.F32 else 4e-3)   # the JIT approximates tanh/sigmoid


def test_incomplete_and_unsupported_equations_return_null(api):
    if api.hip_available() != 1:
        pytest.skip("needs a GPU: without one every dispatch is NULL anyway")
    idx = api.meqn_create()
    assert api.meqn_push_back_binary_op(capi.MeqnMetadata(idx, -1), BINARY.ADD, DT.F32, 0) == 0
    assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, 0), capi.MeqnArgShape(8, 8, 8, DT.F32), SINGULAR) == 0
    assert api.dispatch_meqn(idx, capi.MeqnArgShape(8, 8, 8, DT.F32)) is None          # second operand missing


# device tanhf / the matrix core's summation order / the tree-shaped sum of a dot product: not bit-identical
BY_NORM = {"reduce_bcast": 1e-5, "tanh_sigmoid_chain": 1e-6, "matmul_mul": 1e-6, "matmul_vnni_bf16": 8e-3, "dot_to_scalar": 2e-5, "mul_dot_to_scalar": 2e-5,
           "softmax_fwd": 8e-3, "softmax_bwd": 1e-5, "sum_of_squares": 1e-5, "matmul_sum_to_scalar": 1e-5}
# reductions that end in ONE number and (up to 2^14 elements, round 3) vector-valued ones: phases of the one-workgroup kernel
FUSABLE = {"simple", "bias_relu_bf16", "ternary_muladd", "mixed_precision", "tanh_sigmoid_chain", "layernorm_affine", "dot_to_scalar", "mul_dot_to_scalar",
           "softmax_fwd", "softmax_bwd", "sum_of_squares"}
if __import__("os").environ.get("LIBXSMM_HIP_MEQN_VECRED") != "0":      # LIBXSMM_HIP_MEQN_VECRED=0 keeps trees with vector-valued reductions a chain
    FUSABLE = FUSABLE | {"reduce_bcast"}


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [0, 2], ids=["tpp_chain", "fused_jit"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_meqn_matches_oracle_composition(name, jit):
    import torch
    api = capi.load()
    api.hip_set_jit(jit)
    tree, shapes, out_shape = CASES[name]
    arrays = _inputs(shapes, 7)
    ref = evaluate(tree, shapes, arrays, out_shape)
    idx = build(api, tree, shapes)
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    api.hip_set_jit(1)
    assert h
    assert api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape)) == h                  # cached per output shape
    kname = api.hip_kernel_name(h, 0).decode()
    assert kname.startswith("meqn_jit") == (jit == 2 and name in FUSABLE), kname
    view = lambda a: a.view(np.int16) if a.dtype == np.uint16 else a
    dev = [torch.from_numpy(view(a).copy()).to("cuda:0") for a in arrays]
    out = torch.zeros(out_shape[2] * out_shape[1], dtype=torch.int16 if out_shape[3] == DT.BF16 else torch.float32, device="cuda:0")
    _call(api, h, [d.data_ptr() for d in dev], out.data_ptr())
    api.hip_sync(); api.check()
    got = out.cpu().numpy().view(NPDT[out_shape[3]])
    # every node is a TPP kernel that is bit-identical to its oracle -> so is the composition (device tanhf differs from
    # the host's by ulps: the transcendental case is compared in norm)
    if name in BY_NORM:
        assert normf_rel(_valid(ref, out_shape), _valid(got, out_shape), out_shape[3]) < BY_NORM[name]
    else:
        assert np.array_equal(_valid(got, out_shape), _valid(ref, out_shape))
    # a second call with other inputs reuses handle and workspace
    arrays2 = _inputs(shapes, 8)
    ref2 = evaluate(tree, shapes, arrays2, out_shape)
    dev2 = [torch.from_numpy(view(a).copy()).to("cuda:0") for a in arrays2]
    _call(api, h, [d.data_ptr() for d in dev2], out.data_ptr())
    api.hip_sync(); api.check()
    got2 = out.cpu().numpy().view(NPDT[out_shape[3]])
    if name in BY_NORM:
        assert normf_rel(_valid(ref2, out_shape), _valid(got2, out_shape), out_shape[3]) < BY_NORM[name]
    else:
        assert np.array_equal(_valid(got2, out_shape), _valid(ref2, out_shape))


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [0, 2], ids=["tpp_chain", "fused_jit"])
def test_gpu_meqn_dump_writes_the_intermediate_to_the_op_argument(jit):
    """UNARY_DUMP inside a tree = identity whose value also lands in ops_args[op_arg_pos].primary (the intermediate's own
    leading dimension and type) [ref: src/generator_matequation_reference_impl.c:58-60, mateltwise ref :2478-2494] -- the
    mechanism equation_softmax.c uses to keep exp(x - max) for the backward pass."""
    import torch
    api = capi.load()
    shapes, out_shape = [(M, N, LD, DT.F32), (M, N, LD, DT.F32)], (M, N, LD, DT.F32)
    arrays = _inputs(shapes, 11)
    idx = api.meqn_create()
    assert api.meqn_push_back_binary_op(capi.MeqnMetadata(idx, -1), BINARY.MUL, DT.F32, 0) == 0
    assert api.meqn_push_back_unary_op(capi.MeqnMetadata(idx, 1), UNARY.DUMP, DT.F32, 0) == 0
    assert api.meqn_push_back_unary_op(capi.MeqnMetadata(idx, -1), UNARY.X2, DT.F32, 0) == 0
    assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, 0), capi.MeqnArgShape(*shapes[0]), SINGULAR) == 0
    assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, 1), capi.MeqnArgShape(*shapes[1]), SINGULAR) == 0
    api.hip_set_jit(jit)
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    api.hip_set_jit(1)
    assert h
    assert api.hip_kernel_name(h, 0).decode().startswith("meqn_jit") == (jit == 2)      # the generated kernel stores the second image itself
    dev = [torch.from_numpy(a.copy()).to("cuda:0") for a in arrays]
    out = torch.zeros(LD * N, dtype=torch.float32, device="cuda:0")
    dump = torch.full((M * N,), -7.0, dtype=torch.float32, device="cuda:0")          # the intermediate is M x N with ld = M
    inputs = (capi.MatrixArg * 2)()
    inputs[0].primary, inputs[1].primary = dev[0].data_ptr(), dev[1].data_ptr()
    ops = (capi.MatrixOpArg * 2)()
    ops[1].primary = dump.data_ptr()
    p = capi.MeqnParam()
    p.inputs, p.ops_args = inputs, ops
    p.output.primary = out.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    a0, a1 = _valid(arrays[0], shapes[0]), _valid(arrays[1], shapes[1])
    sq = a0 * a0
    assert np.array_equal(dump.cpu().numpy().reshape(N, M), sq)
    assert np.array_equal(_valid(out.cpu().numpy(), out_shape), sq * a1)
    ops[1].primary = None                                                            # a missing destination is an error, not a fault
    capi.Api.call(h, p)
    assert api.hip_get_last_error() != 0
    api.hip_clear_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [0, 2], ids=["tpp_chain", "fused_jit"])
def test_gpu_meqn_scalar_arguments_may_live_in_host_memory(jit):
    """1 x 1 arguments (a learning rate, a mean) are usually stack variables of the caller (samples/equation/equation_splitSGD.c:
    arg_array[2].primary = &lr): both evaluation paths stage them instead of faulting."""
    import torch
    api = capi.load()
    api.hip_set_jit(jit)
    tree, shapes, out_shape = CASES["mixed_precision"]
    arrays = _inputs(shapes, 13)
    ref = evaluate(tree, shapes, arrays, out_shape)
    idx = build(api, tree, shapes)
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    api.hip_set_jit(1)
    assert h
    view = lambda a: a.view(np.int16) if a.dtype == np.uint16 else a   # noqa: E731
    dev = [torch.from_numpy(view(a).copy()).to("cuda:0") for a in arrays[:2]]
    scalar = C.c_float(float(arrays[2][0]))                                        # plain host memory
    out = torch.zeros(out_shape[2] * out_shape[1], dtype=torch.int16, device="cuda:0")
    for _ in range(3):                                                             # the staging scratch is recycled call after call
        _call(api, h, [dev[0].data_ptr(), dev[1].data_ptr(), C.addressof(scalar)], out.data_ptr())
    api.hip_sync(); api.check()
    assert np.array_equal(_valid(out.cpu().numpy().view(np.uint16), out_shape), _valid(ref, out_shape))


@pytest.mark.gpu
def test_gpu_meqn_gemm_node_below_a_scalar_head_in_host_memory():
    """sum(A0 x A1) with the 1 x 1 result on the caller's stack (synchronous mode): the MATMUL step is a nested handle invocation and must
    not rewind the staging scratch or drop the pending copy-back of the staged scalar (round-2 advisor finding, csrc/meqn.cpp)."""
    import torch
    api = capi.load()
    tree, shapes, out_shape = CASES["matmul_sum_to_scalar"]
    idx = build(api, tree, shapes)
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    assert h
    for seed in (21, 22, 23):                                     # repeated calls: the scratch is recycled call after call
        arrays = _inputs(shapes, seed)
        ref = evaluate(tree, shapes, arrays, out_shape)
        dev = [torch.from_numpy(a.copy()).to("cuda:0") for a in arrays]
        result = C.c_float(-12345.0)                              # plain host memory
        _call(api, h, [d.data_ptr() for d in dev], C.addressof(result))
        api.check()
        assert abs(result.value - float(ref[0])) <= 1e-5 * max(1.0, abs(float(ref[0]))), (seed, result.value, float(ref[0]))


# ---- MATMUL / BRGEMM and GATHER nodes (samples/equation/equation_matmul.c, equation_gather_reduce.c) ------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("jit", [0, 2], ids=["tpp_chain", "fused_jit"])
@pytest.mark.parametrize("dt", [DT.F32, DT.BF16], ids=["f32", "bf16"])
def test_gpu_meqn_softmax_forward_reads_back_what_its_dump_node_wrote(dt, jit):
    """The forward tree of samples/equation/equation_softmax.c:527-538 as the driver builds it: the head multiplies ARGUMENT 0 -- the very buffer
    the DUMP node below the sum writes exp(x - max) to -- with the reciprocal of the sum.  One generated kernel: a max phase, a phase that
    writes the exponentials and sums them, and the scaling phase in which every thread reads back its own units."""
    import torch
    api = capi.load()
    m, n, ld = 64, 12, 128
    x = rand_values(np.random.default_rng(21), ld * n, dt)
    idx = api.meqn_create()
    OP, DUMP_AT = capi.MeqnMetadata(idx, -1), capi.MeqnMetadata(idx, 31)
    rows, cols = UNARY_FLAG.REDUCE_ROWS, UNARY_FLAG.REDUCE_COLS
    assert api.meqn_push_back_binary_op(OP, BINARY.MUL, DT.F32, BINARY_FLAG.BCAST_SCALAR_IN_1) == 0
    assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, 0), capi.MeqnArgShape(m, n, m, DT.F32), SINGULAR) == 0
    assert api.meqn_push_back_unary_op(OP, UNARY.RECIPROCAL, DT.F32, 0) == 0
    assert api.meqn_push_back_unary_op(OP, UNARY.REDUCE_X_OP_ADD, DT.F32, rows) == 0
    assert api.meqn_push_back_unary_op(OP, UNARY.REDUCE_X_OP_ADD, DT.F32, cols) == 0
    assert api.meqn_push_back_unary_op(DUMP_AT, UNARY.DUMP, DT.F32, 0) == 0
    assert api.meqn_push_back_unary_op(OP, UNARY.EXP, DT.F32, 0) == 0
    assert api.meqn_push_back_binary_op(OP, BINARY.SUB, DT.F32, BINARY_FLAG.BCAST_SCALAR_IN_1) == 0
    assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, 1), capi.MeqnArgShape(m, n, ld, dt), SINGULAR) == 0
    assert api.meqn_push_back_unary_op(OP, UNARY.REDUCE_X_OP_MAX, DT.F32, rows) == 0
    assert api.meqn_push_back_unary_op(OP, UNARY.REDUCE_X_OP_MAX, DT.F32, cols) == 0
    assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, 1), capi.MeqnArgShape(m, n, ld, dt), SINGULAR) == 0
    api.hip_set_jit(jit)
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(m, n, ld, dt))
    api.hip_set_jit(1)
    assert h
    assert api.hip_kernel_name(h, 0).decode().startswith("meqn_jit_r") == (jit == 2)
    view = lambda a: a.view(np.int16) if a.dtype == np.uint16 else a
    dx = torch.from_numpy(view(x).copy()).to("cuda:0")
    kept = torch.zeros(m * n, dtype=torch.float32, device="cuda:0")
    out = torch.zeros(ld * n, dtype=torch.int16 if dt == DT.BF16 else torch.float32, device="cuda:0")
    inputs = (capi.MatrixArg * 2)()
    inputs[0].primary, inputs[1].primary = kept.data_ptr(), dx.data_ptr()
    ops = (capi.MatrixOpArg * 32)()
    ops[31].primary = kept.data_ptr()
    p = capi.MeqnParam()
    p.inputs, p.ops_args = inputs, ops
    p.output.primary = out.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    xf = _valid(_f32(x, dt), (m, n, ld, dt)).astype(np.float64)
    ex = np.exp(xf - xf.max())
    want = ex / ex.sum()
    got = _valid(_f32(out.cpu().numpy().view(NPDT[dt]), dt), (m, n, ld, dt))
    assert np.abs(kept.cpu().numpy().reshape(n, m) - ex).max() < 1e-6 * ex.max()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < (4e-3 if dt == DT.BF16 else 1e-6)
    assert abs(float(got.sum()) - 1.0) < (2e-2 if dt == DT.BF16 else 1e-5)


def _f32(bits_or_f32, dt):
    return bits_or_f32 if dt == DT.F32 else (bits_or_f32.astype(np.uint32) << 16).view(np.float32)


def _mat(x, m, n, ld, dt, blocks=1):
    return _f32(x, dt).reshape(blocks, n, ld)[:, :, :m].astype(np.float64)      # [block][col][row]


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [DT.F32, DT.BF16], ids=["f32", "bf16"])
def test_gpu_meqn_matmul_node(dt):
    """out = tanh(C x D) * (A + B): a BINARY_MATMUL node below element-wise nodes (equation 0 of samples/equation/equation_matmul.c)."""
    import torch
    api = capi.load()
    m, n, k = 48, 24, 40
    shapes = [(m, n, m + 8, dt), (m, n, m, dt), (m, k, m + 4, dt), (k, n, k + 2, dt)]
    out_shape = (m, n, m + 8, dt)
    arrays = _inputs(shapes, 5)
    idx = api.meqn_create()
    md = lambda pos=-1: capi.MeqnMetadata(idx, pos)    # noqa: E731
    assert api.meqn_push_back_binary_op(md(), BINARY.MUL, DT.F32, 0) == 0
    assert api.meqn_push_back_unary_op(md(), UNARY.TANH, DT.F32, 0) == 0
    assert api.meqn_push_back_binary_op(md(), BINARY.MATMUL, DT.F32, 0) == 0
    assert api.meqn_push_back_arg(md(2), capi.MeqnArgShape(*shapes[2]), SINGULAR) == 0
    assert api.meqn_push_back_arg(md(3), capi.MeqnArgShape(*shapes[3]), SINGULAR) == 0
    assert api.meqn_push_back_binary_op(md(), BINARY.ADD, DT.F32, 0) == 0
    assert api.meqn_push_back_arg(md(0), capi.MeqnArgShape(*shapes[0]), SINGULAR) == 0
    assert api.meqn_push_back_arg(md(1), capi.MeqnArgShape(*shapes[1]), SINGULAR) == 0
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    assert h
    dev = [torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to("cuda:0") for a in arrays]
    out = torch.zeros(out_shape[2] * n, dtype=torch.float32 if dt == DT.F32 else torch.int16, device="cuda:0")
    _call(api, h, [d.data_ptr() for d in dev], out.data_ptr())
    api.hip_sync(); api.check()
    A, B, Cm, D = (_mat(arrays[i], *shapes[i])[0] for i in range(4))
    gold = np.tanh(D @ Cm) * (A + B)                                              # [col][row] storage: (C x D)^T = D^T-major product
    got = _mat(out.cpu().numpy().view(NPDT[dt]), *out_shape)[0]
    assert np.sqrt(((got - gold) ** 2).sum() / (gold ** 2).sum()) < (2e-5 if dt == DT.F32 else 8e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["binary", "ternary_arg", "ternary_tmp", "vnni_bf16"])
def test_gpu_meqn_brgemm_node(variant):
    """BRGEMM nodes: A and B are strided sets of `blocks` matrices, the count arrives in ops_args[pos].tertiary; the ternary form
    accumulates into its third operand in place -- an argument (equation 1 of equation_matmul.c: that input is overwritten) or an
    intermediate (equation 3); A_VNNI takes bf16 A blocks in VNNI-2 layout."""
    import torch
    api = capi.load()
    m, n, k, blocks = 32, 16, 24, 5
    dt = DT.BF16 if variant == "vnni_bf16" else DT.F32
    es = 2 if dt == DT.BF16 else 4
    shapes = [(m, n, m, dt), (m, n, m + 4, DT.F32), (m, k, m, dt), (k, n, k, dt)]
    rng = np.random.default_rng(9)
    arrays = [rand_values(rng, shapes[0][2] * n, dt), rand_values(rng, shapes[1][2] * n, DT.F32),
              rand_values(rng, m * k * blocks, dt), rand_values(rng, k * n * blocks, dt)]
    set_a = capi.MatrixArgAttributes(1, 3, blocks, m * k * es)                   # TYPE_SET, STRIDE_BASE, cardinality, byte stride
    set_b = capi.MatrixArgAttributes(1, 3, blocks, k * n * es)
    idx = api.meqn_create()
    md = lambda pos=-1: capi.MeqnMetadata(idx, pos)    # noqa: E731
    out_shape = (m, n, m, dt)
    assert api.meqn_push_back_binary_op(md(), BINARY.ADD, DT.F32, 0) == 0
    assert api.meqn_push_back_arg(md(0), capi.MeqnArgShape(*shapes[0]), SINGULAR) == 0
    if variant in ("binary", "vnni_bf16"):
        assert api.meqn_push_back_binary_op(md(3), BINARY.BRGEMM_A_VNNI if variant == "vnni_bf16" else BINARY.BRGEMM, DT.F32, 0) == 0
    else:
        assert api.meqn_push_back_ternary_op(md(3), TERNARY.BRGEMM, DT.F32, TERNARY_FLAG.REUSE_IN_2_AS_OUT) == 0
    assert api.meqn_push_back_arg(md(2), capi.MeqnArgShape(*shapes[2]), set_a) == 0
    assert api.meqn_push_back_arg(md(3), capi.MeqnArgShape(*shapes[3]), set_b) == 0
    if variant == "ternary_arg":
        assert api.meqn_push_back_arg(md(1), capi.MeqnArgShape(*shapes[1]), SINGULAR) == 0
    elif variant == "ternary_tmp":
        assert api.meqn_push_back_unary_op(md(), UNARY.X2, DT.F32, 0) == 0
        assert api.meqn_push_back_arg(md(1), capi.MeqnArgShape(*shapes[1]), SINGULAR) == 0
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    assert h
    A0, Cacc = _mat(arrays[0], *shapes[0])[0], _mat(arrays[1], *shapes[1])[0]
    Ab, Bb = _mat(arrays[2], m, k, m, dt, blocks), _mat(arrays[3], k, n, k, dt, blocks)
    prod = sum(Bb[r] @ Ab[r] for r in range(blocks))                             # [col][row] storage
    a_dev = arrays[2]
    if variant == "vnni_bf16":                                                   # [block][k/2][m][2] from [block][k][m]
        a_dev = np.ascontiguousarray(arrays[2].reshape(blocks, k // 2, 2, m).transpose(0, 1, 3, 2)).reshape(-1)
    host = [arrays[0], arrays[1], a_dev, arrays[3]]
    dev = [torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a.copy()).to("cuda:0") for a in host]
    out = torch.zeros(out_shape[2] * n, dtype=torch.float32 if dt == DT.F32 else torch.int16, device="cuda:0")
    inputs = (capi.MatrixArg * 4)()
    for i, d in enumerate(dev):
        inputs[i].primary = d.data_ptr()
    count = C.c_ulonglong(blocks)
    ops = (capi.MatrixOpArg * 4)()
    ops[3].tertiary = C.addressof(count)
    p = capi.MeqnParam()
    p.inputs, p.ops_args = inputs, ops
    p.output.primary = out.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    gold = A0 + {"binary": prod, "vnni_bf16": prod, "ternary_arg": Cacc + prod, "ternary_tmp": Cacc * Cacc + prod}[variant]
    got = _mat(out.cpu().numpy().view(NPDT[dt]), *out_shape)[0]
    assert np.sqrt(((got - gold) ** 2).sum() / (gold ** 2).sum()) < (2e-6 if dt == DT.F32 else 8e-3)
    after = _mat(dev[1].cpu().numpy(), *shapes[1])[0]
    if variant == "ternary_arg":                                                 # the accumulator argument now holds C + sum A_r x B_r
        assert np.sqrt((((Cacc + prod) - after) ** 2).sum() / ((Cacc + prod) ** 2).sum()) < 2e-6
    else:
        assert np.array_equal(after, Cacc)
    ops[3].tertiary = None                                                       # a missing block count is an error, not a fault
    capi.Api.call(h, p)
    assert api.hip_get_last_error() != 0
    api.hip_clear_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("idx_bytes", [4, 8])
@pytest.mark.parametrize("dt", [DT.F32, DT.BF16], ids=["f32", "bf16"])
def test_gpu_meqn_gather_node_above_an_argument(dt, idx_bytes):
    """reduce_cols_add(gather_cols(X, idx)): the GATHER node reads its column indices from inputs[pos].secondary
    (samples/equation/equation_gather_reduce.c:146-166)."""
    import torch
    api = capi.load()
    m, n, ld, big_n = 37, 21, 40, 21 * 5
    rng = np.random.default_rng(3)
    X = rand_values(rng, ld * big_n, dt)
    cols = rng.permutation(big_n)[:n].astype(np.uint32 if idx_bytes == 4 else np.uint64)
    idx = api.meqn_create()
    md = capi.MeqnMetadata(idx, -1)
    gflags = UNARY_FLAG.GS_COLS | (UNARY_FLAG.IDX_SIZE_4BYTES if idx_bytes == 4 else UNARY_FLAG.IDX_SIZE_8BYTES)
    assert api.meqn_push_back_unary_op(md, UNARY.REDUCE_X_OP_ADD, dt, UNARY_FLAG.REDUCE_COLS) == 0
    assert api.meqn_push_back_unary_op(md, UNARY.GATHER, dt, gflags) == 0
    assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, 0), capi.MeqnArgShape(m, n, ld, dt), SINGULAR) == 0
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(m, 1, ld, dt))
    assert h
    xd = torch.from_numpy(X.view(np.int16) if X.dtype == np.uint16 else X).to("cuda:0")
    idd = torch.from_numpy(cols.view(np.int32 if idx_bytes == 4 else np.int64)).to("cuda:0")
    out = torch.zeros(ld, dtype=torch.float32 if dt == DT.F32 else torch.int16, device="cuda:0")
    inputs = (capi.MatrixArg * 1)()
    inputs[0].primary, inputs[0].secondary = xd.data_ptr(), idd.data_ptr()
    p = capi.MeqnParam()
    p.inputs = inputs
    p.output.primary = out.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    Xf = _f32(X, dt).reshape(big_n, ld)[:, :m].astype(np.float64)
    gold = Xf[cols.astype(np.int64)].sum(axis=0)
    got = _f32(out.cpu().numpy().view(NPDT[dt]), dt)[:m].astype(np.float64)
    assert np.sqrt(((got - gold) ** 2).sum() / (gold ** 2).sum()) < (1e-6 if dt == DT.F32 else 8e-3)
    inputs[0].secondary = None
    capi.Api.call(h, p)
    assert api.hip_get_last_error() != 0
    api.hip_clear_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("in_place", [True, False], ids=["in_place", "separate_output"])
def test_gpu_meqn_accumulating_gemm_as_the_head(in_place):
    """C += A x B as the WHOLE equation: a TERNARY_MATMUL head with REUSE_IN_2_AS_OUT.  Its result belongs in the caller's output: with the accumulator
    passed as input 2 AND as the output nothing is copied; with a separate output the accumulator is copied there first and stays untouched."""
    import torch
    api = capi.load()
    m, n, k = 32, 48, 24
    shapes = [(m, k, m, DT.F32), (k, n, k + 2, DT.F32), (m, n, m + 4, DT.F32)]
    arrays = _inputs(shapes, 11)
    idx = api.meqn_create()
    md = lambda pos=-1: capi.MeqnMetadata(idx, pos)    # noqa: E731
    assert api.meqn_push_back_ternary_op(md(), TERNARY.MATMUL, DT.F32, TERNARY_FLAG.REUSE_IN_2_AS_OUT) == 0
    for i in range(3):
        assert api.meqn_push_back_arg(md(i), capi.MeqnArgShape(*shapes[i]), SINGULAR) == 0
    out_shape = shapes[2] if in_place else (m, n, m + 8, DT.F32)
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    assert h
    dev = [torch.from_numpy(a.copy()).to("cuda:0") for a in arrays]
    out = dev[2] if in_place else torch.zeros(out_shape[2] * n, dtype=torch.float32, device="cuda:0")
    _call(api, h, [d.data_ptr() for d in dev], out.data_ptr())
    api.hip_sync(); api.check()
    A, B, Cacc = (_mat(arrays[i], *shapes[i])[0] for i in range(3))
    gold = Cacc + B @ A                                                          # [col][row] storage
    got = _mat(out.cpu().numpy(), *out_shape)[0]
    assert np.sqrt(((got - gold) ** 2).sum() / (gold ** 2).sum()) < 2e-6
    if not in_place:
        assert np.array_equal(dev[2].cpu().numpy(), arrays[2])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [DT.F32, DT.BF16], ids=["f32", "bf16"])
def test_gpu_meqn_scatter_as_the_head(dt):
    """scatter_cols(A + B, idx): SCATTER exists as the head of an equation only and takes its index list from output.secondary
    (generator_matequation_reference_impl.c:41-56); the columns that no index names keep what the output held."""
    import torch
    api = capi.load()
    m, n, ld, big_n = 40, 13, 48, 40
    shapes = [(m, n, ld, dt), (m, n, m, dt)]
    arrays = _inputs(shapes, 13)
    rng = np.random.default_rng(4)
    cols = rng.permutation(big_n)[:n].astype(np.uint32)
    idx = api.meqn_create()
    md = lambda pos=-1: capi.MeqnMetadata(idx, pos)    # noqa: E731
    assert api.meqn_push_back_unary_op(md(), UNARY.SCATTER, dt, UNARY_FLAG.GS_COLS | UNARY_FLAG.IDX_SIZE_4BYTES) == 0
    assert api.meqn_push_back_binary_op(md(), BINARY.ADD, dt, 0) == 0                # SCATTER copies elements of its operand's width: the sum is produced in the output type
    assert api.meqn_push_back_arg(md(0), capi.MeqnArgShape(*shapes[0]), SINGULAR) == 0
    assert api.meqn_push_back_arg(md(1), capi.MeqnArgShape(*shapes[1]), SINGULAR) == 0
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(m, big_n, ld, dt))
    assert h
    dev = [torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to("cuda:0") for a in arrays]
    before = rand_values(rng, ld * big_n, dt)
    out = torch.from_numpy(before.view(np.int16) if before.dtype == np.uint16 else before.copy()).to("cuda:0")
    idd = torch.from_numpy(cols.view(np.int32)).to("cuda:0")
    inputs = (capi.MatrixArg * 2)()
    for i, d in enumerate(dev):
        inputs[i].primary = d.data_ptr()
    p = capi.MeqnParam()
    p.inputs = inputs
    p.output.primary, p.output.secondary = out.data_ptr(), idd.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    A, B = (_mat(arrays[i], *shapes[i])[0] for i in range(2))
    gold = _mat(before, m, big_n, ld, dt)[0].copy()
    gold[cols.astype(np.int64)] = A + B
    got = _mat(out.cpu().numpy().view(NPDT[dt]), m, big_n, ld, dt)[0]
    untouched = np.setdiff1d(np.arange(big_n), cols)
    assert np.array_equal(got[untouched], gold[untouched])
    assert np.sqrt(((got - gold) ** 2).sum() / (gold ** 2).sum()) < (1e-6 if dt == DT.F32 else 8e-3)
    p.output.secondary = None
    capi.Api.call(h, p)
    assert api.hip_get_last_error() != 0
    api.hip_clear_last_error()
