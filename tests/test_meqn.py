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
from libxsmm_amd.capi import BINARY, BINARY_FLAG, DT, TERNARY, TERNARY_FLAG, UNARY, UNARY_FLAG
from oracle import pyoracle

NPDT = {DT.F32: np.float32, DT.BF16: np.uint16}
SINGULAR = capi.MatrixArgAttributes(0, 0, 0, 0)
ALPHA = C.c_float(0.125)


# tree notation: ("arg", pos) | ("u", type, flags, child) | ("b", type, flags, l, r) | ("t", type, flags, a, b, c)
def build(api, tree, arg_shapes, comp=DT.F32):
    idx = api.meqn_create()

    def walk(t):
        if t[0] == "arg":
            m, n, ld, dt = arg_shapes[t[1]]
            assert api.meqn_push_back_arg(capi.MeqnMetadata(idx, t[1]), capi.MeqnArgShape(m, n, ld, dt), SINGULAR) == 0
        elif t[0] == "u":
            assert api.meqn_push_back_unary_op(capi.MeqnMetadata(idx, 0 if t[1] == UNARY.LEAKY_RELU else -1), t[1], comp, t[2]) == 0
            walk(t[3])
        elif t[0] == "b":
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
        if t[0] == "u":
            if t[1] in REDUCES:
                m, n = ((n0, 1) if t[2] & UNARY_FLAG.REDUCE_ROWS else (m0, 1)); dm, dn = m0, n0
            else:
                m, n = m0, n0; dm, dn = m0, n0
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


FUSABLE = {"simple", "bias_relu_bf16", "ternary_muladd", "mixed_precision", "tanh_sigmoid_chain", "layernorm_affine"}     # no reduction inside


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
    if name == "tanh_sigmoid_chain":
        assert normf_rel(_valid(ref, out_shape), _valid(got, out_shape), out_shape[3]) < 1e-6
    else:
        assert np.array_equal(_valid(got, out_shape), _valid(ref, out_shape))
    # a second call with other inputs reuses handle and workspace
    arrays2 = _inputs(shapes, 8)
    ref2 = evaluate(tree, shapes, arrays2, out_shape)
    dev2 = [torch.from_numpy(view(a).copy()).to("cuda:0") for a in arrays2]
    _call(api, h, [d.data_ptr() for d in dev2], out.data_ptr())
    api.hip_sync(); api.check()
    got2 = out.cpu().numpy().view(NPDT[out_shape[3]])
    if name == "tanh_sigmoid_chain":
        assert normf_rel(_valid(ref2, out_shape), _valid(got2, out_shape), out_shape[3]) < 1e-6
    else:
        assert np.array_equal(_valid(got2, out_shape), _valid(ref2, out_shape))


@pytest.mark.gpu
def test_gpu_meqn_dump_writes_the_intermediate_to_the_op_argument():
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
    h = api.dispatch_meqn(idx, capi.MeqnArgShape(*out_shape))
    assert h
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
