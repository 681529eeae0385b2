"""Committed golden vectors produced by the reference itself (tests/golden/make_golden.py).  They pin
  * the oracle restatement on boxes without /root/reference or oracle/_ref   (CPU, not gpu)
  * the product library through the C-ABI                                      (-m gpu)
against reference outputs.  GEMM inputs are rebuilt from (kwargs, seed); a stored probe detects RNG drift."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import GemmCase, TOL_BF16, TOL_F32, normf_rel
from libxsmm_amd import capi
from libxsmm_amd.capi import BINARY, BINARY_FLAG, DT, GEMM_FLAG, TERNARY, UNARY, UNARY_FLAG
from oracle import pyoracle

HERE = os.path.dirname(os.path.abspath(__file__))
G = dict(np.load(os.path.join(HERE, "golden", "reference_vectors.npz")))   # materialised: NpzFile re-reads per access
import importlib.util
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
_mg = importlib.util.module_from_spec(_spec)
try:
    _spec.loader.exec_module(_mg)
    GEMM = _mg.GEMM
except Exception:      # pragma: no cover
    GEMM = {}
SEED = 20260923


def test_seeded_inputs_have_not_drifted():
    probe = GemmCase(seed=SEED, **GEMM["cfg2_f32_32_strd"])
    assert np.array_equal(probe.A, G["gemm_probe_A"]) and np.array_equal(probe.B, G["gemm_probe_B"])


@pytest.mark.parametrize("name", sorted(GEMM))
def test_oracle_gemm_reproduces_reference_vectors(name):
    case = GemmCase(seed=SEED, **GEMM[name])
    c, mask = case.run_oracle()
    assert np.array_equal(case.valid_region(c), case.valid_region(G[f"gemm_{name}_C"]))
    if mask is not None:
        assert np.array_equal(case.valid_mask_bits(mask), case.valid_mask_bits(G[f"gemm_{name}_mask"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GEMM))
def test_gpu_gemm_matches_reference_vectors(name):
    case = GemmCase(seed=SEED, **GEMM[name])
    got, _, handle = case.run_gpu()
    ref = G[f"gemm_{name}_C"]
    tol = TOL_BF16 if case.c_type == DT.BF16 else (1e-12 if case.c_type == DT.F64 else TOL_F32)
    if GEMM[name].get("act") == 3:
        tol = 7e-4                                    # fused sigmoid bound, samples/xgemm/gemm_kernel.c:5396
    if case.c_type == DT.I32:                         # integer accumulation: bit-exact
        assert np.array_equal(case.valid_region(ref), case.valid_region(got))
        return
    assert normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type) < tol


UNARY_GOLD = {
    "relu_mask_bf16": (UNARY.RELU, 70, 9, 72, 72, DT.BF16, DT.BF16, UNARY_FLAG.BITMASK_2BYTEMULT),
    "transpose_f32": (UNARY.TRANSFORM_NORM_TO_NORMT, 37, 19, 40, 19, DT.F32, DT.F32, 0),
    "vnni2_bf16": (UNARY.TRANSFORM_NORM_TO_VNNI2, 32, 16, 32, 32, DT.BF16, DT.BF16, 0),
    "gather_cols_f32": (UNARY.GATHER, 24, 10, 40, 24, DT.F32, DT.F32, UNARY_FLAG.GS_COLS | UNARY_FLAG.IDX_SIZE_4BYTES),
    "sigmoid_f32_bf16": (UNARY.SIGMOID, 33, 7, 40, 35, DT.F32, DT.BF16, 0),
    "reduce_rows_add": (UNARY.REDUCE_X_OP_ADD, 75, 33, 80, 33, DT.F32, DT.F32, UNARY_FLAG.REDUCE_ROWS),
}


def _unary_param(tag, X, Y, aux):
    p = capi.UnaryParam()
    p.in_.primary, p.out.primary = X, Y
    if f"tpp_{tag}_idx" in G:
        p.in_.secondary = aux
    elif f"tpp_{tag}_aux" in G:
        p.out.secondary = aux
    return p


@pytest.mark.parametrize("tag", sorted(UNARY_GOLD))
def test_oracle_tpp_reproduces_reference_vectors(tag):
    typ, m, n, ldi, ldo, in_dt, out_dt, flags = UNARY_GOLD[tag]
    X, Y = G[f"tpp_{tag}_in"].copy(), G[f"tpp_{tag}_out0"].copy()
    aux = G[f"tpp_{tag}_idx"].copy() if f"tpp_{tag}_idx" in G else (np.zeros_like(G[f"tpp_{tag}_aux"]) if f"tpp_{tag}_aux" in G else None)
    p = _unary_param(tag, X.ctypes.data, Y.ctypes.data, aux.ctypes.data if aux is not None else 0)
    pyoracle.oracle().meltw(p, pyoracle.MeltwDesc(m, n, ldi, ldo, 0, 0, in_dt, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, out_dt, flags, typ, 1))
    assert np.array_equal(Y, G[f"tpp_{tag}_out"])
    if f"tpp_{tag}_aux" in G:
        assert np.array_equal(aux, G[f"tpp_{tag}_aux"])


def _dev(x):
    import torch
    v = {np.uint16: np.int16, np.uint32: np.int32, np.uint64: np.int64}.get(x.dtype.type)
    return torch.from_numpy(np.ascontiguousarray(x.view(v) if v else x)).to("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(UNARY_GOLD))
def test_gpu_tpp_matches_reference_vectors(tag):
    api = capi.load()
    typ, m, n, ldi, ldo, in_dt, out_dt, flags = UNARY_GOLD[tag]
    X, Y = _dev(G[f"tpp_{tag}_in"]), _dev(G[f"tpp_{tag}_out0"].copy())
    aux = None
    if f"tpp_{tag}_idx" in G:
        aux = _dev(G[f"tpp_{tag}_idx"])
    elif f"tpp_{tag}_aux" in G:
        aux = _dev(np.zeros_like(G[f"tpp_{tag}_aux"]))
    h = api.dispatch_meltw_unary(typ, capi.UnaryShape(m, n, ldi, ldo, in_dt, out_dt, DT.F32), flags)
    assert h
    capi.Api.call(h, _unary_param(tag, X.data_ptr(), Y.data_ptr(), aux.data_ptr() if aux is not None else 0))
    api.hip_sync(); api.check()
    got, exp = Y.cpu().numpy().view(G[f"tpp_{tag}_out"].dtype), G[f"tpp_{tag}_out"]
    if tag == "sigmoid_f32_bf16":
        assert normf_rel(exp, got, out_dt) < 7e-3
    elif tag == "reduce_rows_add":
        assert normf_rel(exp[:n], got[:n], DT.F32) < 1e-5
    else:
        assert np.array_equal(exp, got)
    if f"tpp_{tag}_aux" in G:
        bits = lambda a: np.unpackbits(a.reshape(n, -1), axis=1, bitorder="little")[:, :m]
        assert np.array_equal(bits(aux.cpu().numpy()), bits(G[f"tpp_{tag}_aux"]))


def test_oracle_binary_ternary_reproduce_reference_vectors():
    orc = pyoracle.oracle()
    X0, X1 = G["tpp_biasadd_in0"].copy(), G["tpp_biasadd_in1"].copy()
    Y = np.zeros(64 * 64, dtype=np.uint16)
    p = capi.BinaryParam(); p.in0.primary, p.in1.primary, p.out.primary = X0.ctypes.data, X1.ctypes.data, Y.ctypes.data
    orc.meltw(p, pyoracle.MeltwDesc(64, 64, 64, 64, 64, 0, DT.BF16, DT.BF16, DT.UNSUPPORTED, DT.F32, DT.BF16, BINARY_FLAG.BCAST_COL_IN_0, BINARY.ADD, 2))
    assert np.array_equal(Y, G["tpp_biasadd_out"])
    A0, A1, bits = G["tpp_select_in0"].copy(), G["tpp_select_in1"].copy(), G["tpp_select_bits"].copy()
    Y = np.zeros(48 * 9, dtype=np.float32)
    p = capi.TernaryParam(); p.in0.primary, p.in1.primary, p.in2.primary, p.out.primary = A0.ctypes.data, A1.ctypes.data, bits.ctypes.data, Y.ctypes.data
    orc.meltw(p, pyoracle.MeltwDesc(40, 9, 48, 48, 48, 48, DT.F32, DT.F32, DT.F32, DT.F32, DT.F32, 0, TERNARY.SELECT, 3))
    assert np.array_equal(Y, G["tpp_select_out"])


def test_oracle_sparse_reproduces_reference_vectors():
    orc = pyoracle.oracle()
    rp, ci, va, B, C0 = (G[f"spcsr_edge_{k}"] for k in ("rowptr", "colidx", "vals", "B", "C0"))
    for beta0 in (0, 1):
        c = C0.copy()
        orc.lib.oracle_packed_spgemm_csr_asparse(DT.F32, 35, 16, 35, 16, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, B.ctypes.data, 16, c.ctypes.data, 16, beta0)
        assert normf_rel(G[f"spcsr_edge_C_beta0_{beta0}"], c, DT.F32) <= 1e-5
    a, Bf, Cf = G["fsspmdm_pyfr_A"], G["fsspmdm_pyfr_B"], G["fsspmdm_pyfr_C0"]
    M, K = a.shape
    rowptr = np.zeros(M + 1, dtype=np.uint32); colidx = []; vals = []
    for i in range(M):
        nz = np.nonzero(a[i])[0]; colidx += list(nz); vals += list(a[i, nz]); rowptr[i + 1] = len(colidx)
    colidx, vals = np.array(colidx, dtype=np.uint32), np.array(vals, dtype=np.float64)
    c = Cf.copy()
    orc.lib.oracle_fsspmdm(DT.F64, M, 96, K, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, Bf.ctypes.data, 96, c.ctypes.data, 96, 0)
    assert normf_rel(G["fsspmdm_pyfr_C"], c, DT.F64) <= 1e-12
    if "bcsc_C" in G:
        c = np.zeros_like(G["bcsc_C"])
        orc.lib.oracle_packed_spgemm_bcsc(DT.BF16, DT.BF16, 64, 64, 256, 3, 32, 32, 1, G["bcsc_A"].ctypes.data, G["bcsc_bvals"].ctypes.data,
                                          G["bcsc_colptr"].ctypes.data, G["bcsc_rowidx"].ctypes.data, c.ctypes.data, 1)
        assert normf_rel(G["bcsc_C"], c, DT.BF16) <= 5e-3


@pytest.mark.gpu
def test_gpu_sparse_matches_reference_vectors():
    api = capi.load()
    rp, ci, va, B, C0 = (G[f"spcsr_edge_{k}"] for k in ("rowptr", "colidx", "vals", "B", "C0"))
    for beta0 in (0, 1):
        h = api.create_packed_spgemm_csr(capi.gemm_shape(35, 16, 35, 0, 16, 16, DT.F32, DT.F32, DT.F32, DT.F32), GEMM_FLAG.BETA_0 if beta0 else 0, 0, 16,
                                         rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        assert h
        dv, dB, dC = _dev(va), _dev(B), _dev(C0.copy())
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = dv.data_ptr(), dB.data_ptr(), dC.data_ptr()
        capi.Api.call(h, p); api.hip_sync(); api.check()
        assert normf_rel(G[f"spcsr_edge_C_beta0_{beta0}"], dC.cpu().numpy(), DT.F32) <= 1e-5
        api.release_kernel(h)
    a, Bf, Cf = np.ascontiguousarray(G["fsspmdm_pyfr_A"]), G["fsspmdm_pyfr_B"], G["fsspmdm_pyfr_C0"]
    M, K = a.shape
    al, be = C.c_double(1.0), C.c_double(1.0)
    h = api.fsspmdm_create(DT.F64, M, 96, K, K, 96, 96, C.addressof(al), C.addressof(be), a.ctypes.data, 0, None)
    assert h
    dB, dC = _dev(Bf), _dev(Cf.copy())
    api.fsspmdm_execute(h, dB.data_ptr(), dC.data_ptr()); api.hip_sync(); api.check()
    assert normf_rel(G["fsspmdm_pyfr_C"], dC.cpu().numpy(), DT.F64) <= 1e-12
    api.fsspmdm_destroy(h)
    if "bcsc_C" in G:
        h = api.create_packed_spgemm_bcsc(capi.gemm_shape(3, 0, 256, 256, 0, 64, DT.BF16, DT.BF16, DT.BF16, DT.F32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0, capi.SpgemmConfig(64, 32, 32))
        assert h
        dA, dV, dcp, dri = _dev(G["bcsc_A"]), _dev(G["bcsc_bvals"]), _dev(G["bcsc_colptr"]), _dev(G["bcsc_rowidx"])
        dC = _dev(np.zeros_like(G["bcsc_C"]))
        nblk = C.c_ulonglong(2)
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.b.secondary, p.b.tertiary, p.b.quaternary, p.c.primary = dA.data_ptr(), dV.data_ptr(), dcp.data_ptr(), dri.data_ptr(), C.addressof(nblk), dC.data_ptr()
        capi.Api.call(h, p); api.hip_sync(); api.check()
        assert normf_rel(G["bcsc_C"], dC.cpu().numpy().view(np.uint16), DT.BF16) <= 5e-3
        api.release_kernel(h)
