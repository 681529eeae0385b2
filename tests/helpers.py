"""Shared test plumbing: seeded inputs shaped like the reference's drivers produce them, and the
glue that lets ONE problem description be run through three executors:

  * the oracle restatement (oracle/liboracle.so)            -- host numpy buffers
  * the reference itself (oracle/_ref/libxsmm_ref.so)        -- host numpy buffers, when built
  * the product (libxsmm_amd/lib/libxsmm_amd.so)             -- torch CUDA tensors, -m gpu tests only

Values follow samples/xgemm/gemm_kernel.c:837-865 of the reference: multiples of 0.1 in [-0.4, 0.5],
bf16 data produced by truncation of the fp32 pattern (:1000-1002).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from libxsmm_amd import capi  # noqa: E402
from libxsmm_amd.capi import DT, GEMM_FLAG  # noqa: E402
from oracle import pyoracle  # noqa: E402

NP_OF = {DT.F32: np.float32, DT.F64: np.float64, DT.BF16: np.uint16, DT.I32: np.int32, DT.U32: np.uint32,
         DT.I16: np.int16, DT.U16: np.uint16, DT.I8: np.int8, DT.U8: np.uint8, DT.I64: np.int64, DT.U64: np.uint64, DT.BF8: np.uint8, DT.HF8: np.uint8, DT.BF32: np.float32, DT.F16: np.uint16}

UPLOAD_HOOK = None    # tests/guard.py: device buffers that touch unmapped address space instead of torch tensors (x -> object with data_ptr() / cpu().numpy())
FP8_WIDE = False      # tests flip this to draw 8-bit floats over (almost) the whole exponent range

# the reference's own acceptance bounds [samples/xgemm/gemm_kernel.c:5312-5414]
TOL_F32 = 1.2e-5
TOL_BF16 = 5e-3
TOL_F64 = 1e-12


def f32_to_bf16_trunc(x: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def bf16_to_f32(x: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(x, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def _fp8_table(hf8: bool) -> np.ndarray:
    """value of every byte of E5M2 (BF8: the upper byte of an IEEE half) / E4M3 (HF8: bias 7, no infinities, 0x7f = NaN) [ref: src/libxsmm_math.c:546-585]"""
    b = np.arange(256, dtype=np.uint16)
    if not hf8:
        return (b << 8).view(np.float16).astype(np.float64)
    sign = np.where(b & 0x80, -1.0, 1.0)
    e, m = ((b >> 3) & 15).astype(np.int32), (b & 7).astype(np.float64)
    v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    v = np.where((e == 15) & ((b & 7) == 7), np.nan, v)
    return sign * v


_FP8_VALUES = {False: _fp8_table(False), True: _fp8_table(True)}


def as_float(x: np.ndarray, dt: int) -> np.ndarray:
    if dt in (DT.BF8, DT.HF8):
        return _FP8_VALUES[dt == DT.HF8][np.ascontiguousarray(x).view(np.uint8)]
    if dt == DT.F16:
        return np.ascontiguousarray(x).view(np.float16).astype(np.float64) if x.dtype in (np.uint16, np.int16) else x.astype(np.float64)
    return bf16_to_f32(x).astype(np.float64) if dt == DT.BF16 else x.astype(np.float64)


def rand_values(rng: np.random.Generator, count: int, dt: int) -> np.ndarray:
    if dt in (DT.I8, DT.I16, DT.I32):
        return rng.integers(-9, 10, count).astype(NP_OF[dt])
    if dt in (DT.U8, DT.U16, DT.U32):
        return rng.integers(0, 19, count).astype(NP_OF[dt])
    if dt == DT.MXFP4X2:                    # two E2M1 codes per byte, all 256 combinations
        return rng.integers(0, 256, count).astype(np.uint8)
    if dt in (DT.MXBF8, DT.MXHF8):          # the element encodings are plain E5M2 / E4M3
        return rand_values(rng, count, DT.BF8 if dt == DT.MXBF8 else DT.HF8)
    if dt in (DT.BF8, DT.HF8):              # finite 8-bit floats in [1/8, 2) with random sign, plus a few zeros: the magnitude range of the
        bias, mbits = (15, 2) if dt == DT.BF8 else (7, 3)     # reference driver's data (multiples of 0.1 in [-0.5, 0.5]); FP8_WIDE widens it
        lo = max(-14 if FP8_WIDE else -3, -bias)           # exponent field 0 = subnormals / zero
        e = rng.integers(bias + lo, bias + (4 if FP8_WIDE else 1), count)
        v = ((rng.integers(0, 2, count) << 7) | (e << mbits) | rng.integers(0, 1 << mbits, count)).astype(np.uint8)
        v[rng.random(count) < 0.05] = 0
        return v
    if dt == DT.BF32:                       # f32 storage, not bf16-representable: the rounding of the operands matters
        return ((rng.random(count).astype(np.float32) - 0.5) * 1.37).astype(np.float32)
    v = (np.floor(rng.random(count) * 10.0) - 4.0) / 10.0
    if dt == DT.F16:
        return v.astype(np.float16).view(np.uint16)
    if dt == DT.BF16:
        return f32_to_bf16_trunc(v.astype(np.float32))
    return v.astype(NP_OF[dt])


def normf_rel(ref: np.ndarray, tst: np.ndarray, dt: int) -> float:
    r, t = as_float(ref.ravel(), dt), as_float(tst.ravel(), dt)
    den = float(np.sum(r * r))
    num = float(np.sum((r - t) ** 2))
    return float(np.sqrt(num / den)) if den > 0 else float(np.sqrt(num))


def ptr(a) -> int:
    """Address of a numpy array or a torch tensor."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()


class GemmCase:
    """One (BR)GEMM problem: descriptor inputs + host buffers."""

    def __init__(self, m, n, k, a_type=DT.F32, c_type=None, lda=None, ldb=None, ldc=None, flags=0,
                 br_type=capi.BR_NONE, br_count=1, colbias=False, act=0, batch=1, seed=0, beta=0, shared_b=False, b_type=None, scf=None):
        rng = np.random.default_rng(seed)
        self.m, self.n, self.k = m, n, k
        self.a_type = a_type
        self.b_type = a_type if b_type is None else b_type
        self.c_type = (DT.F32 if a_type == DT.BF32 else a_type) if c_type is None else c_type
        self.comp_type = DT.F64 if a_type == DT.F64 else (DT.I32 if a_type in (DT.I8, DT.U8) else DT.F32)   # fp8: f32
        self.scf = None if scf is None else C.c_float(scf)           # 8-bit GEMM with f32 output: scale read from c.tertiary
        ta, tb = bool(flags & GEMM_FLAG.TRANS_A), bool(flags & GEMM_FLAG.TRANS_B)
        self.lda = lda if lda is not None else (k if ta else m)
        self.ldb = ldb if ldb is not None else (n if tb else k)
        self.ldc = ldc if ldc is not None else m
        self.flags = flags | (GEMM_FLAG.BETA_0 if beta == 0 else 0)
        self.br_type, self.br_count, self.batch = br_type, br_count, batch
        self.colbias, self.act = colbias, act
        self.ext = colbias or act != 0
        self.a_elems = self.lda * (m if ta else k)
        self.mxmx = a_type in (DT.MXFP4X2, DT.MXBF8, DT.MXHF8) and self.b_type == a_type     # both operands microscaled
        self.mx = a_type == DT.MXFP4X2 and not self.mxmx   # packed E2M1 pairs: lda bytes per k-pair, one E8M0 scale per (32 k, row)
        if self.mx or self.mxmx:
            assert k % 32 == 0 and not ta
            self.a_elems = self.lda * k // (2 if a_type == DT.MXFP4X2 else 1)
            self.s_elems = self.lda * (k // 32)
        self.b_elems = self.ldb * (k if tb else n)
        if self.mxmx:                           # B in A's layout, indexed by the column: [k-group][ldb][4 bytes]
            assert tb and self.ldb >= n
            self.b_elems = self.ldb * k // (2 if a_type == DT.MXFP4X2 else 1)
            self.sb_elems = self.ldb * (k // 32)
        vf = 4 if capi.DT_SIZE[self.c_type] == 1 else 2                 # VNNI_C re-lays 8-bit results as VNNI-4, 16-bit ones as VNNI-2: room for the pad columns
        self.c_elems = self.ldc * (((n + vf - 1) // vf) * vf if flags & GEMM_FLAG.VNNI_C else n)
        asz, csz, bsz = capi.DT_SIZE[a_type], capi.DT_SIZE[self.c_type], capi.DT_SIZE[self.b_type]
        nbr = br_count if br_type != capi.BR_NONE else 1
        self.nbr = nbr
        # every batch element owns nbr A blocks and nbr B blocks (B optionally shared across the batch)
        self.A = rand_values(rng, batch * nbr * self.a_elems, a_type)
        self.B = rand_values(rng, (1 if shared_b else batch) * nbr * self.b_elems, self.b_type)
        self.C0 = rand_values(rng, batch * self.c_elems, self.c_type)
        self.D = rand_values(rng, batch * m, self.c_type) if colbias else None
        self.mask_ld = ((self.ldc + 15) // 16) * 16
        self.mask_bytes = (self.mask_ld // 8) * n
        self.bs_a = nbr * self.a_elems * asz
        self.bs_b = 0 if shared_b else nbr * self.b_elems * bsz
        self.bs_c = self.c_elems * csz
        self.bs_d = m * csz if colbias else 0
        self.br_stride_a = self.a_elems * asz
        if self.mx or self.mxmx:                # scales in a narrow band around 1.0 with a few exact zeros (scale byte 0 decodes to 0.0f)
            self.S = rng.integers(124, 131, batch * nbr * self.s_elems).astype(np.uint8)
            if self.mx:
                self.S[rng.random(self.S.size) < 0.02] = 0
            self.bs_s = nbr * self.s_elems
        if self.mxmx:
            self.SB = rng.integers(124, 131, (1 if shared_b else batch) * nbr * self.sb_elems).astype(np.uint8)
            self.bs_sb = 0 if shared_b else nbr * self.sb_elems
        self.br_stride_b = self.b_elems * bsz
        # OFFSET mode: a permutation of the nbr blocks (byte offsets), shared by the batch
        perm = rng.permutation(nbr)
        self.offs_a = (perm * self.br_stride_a).astype(np.int64)
        self.offs_b = (perm[::-1].copy() * self.br_stride_b).astype(np.int64)

    # ---- descriptor pieces ------------------------------------------------------------------
    def shape(self) -> capi.GemmShape:
        return capi.gemm_shape(self.m, self.n, self.k, self.lda, self.ldb, self.ldc, self.a_type, self.b_type, self.c_type, self.comp_type)

    def brcfg(self) -> capi.BrConfig:
        if self.br_type == capi.BR_STRIDE:
            return capi.br_config(capi.BR_STRIDE, self.br_stride_a, self.br_stride_b, 0)
        return capi.br_config(self.br_type, 0, 0, 0)

    def argops(self) -> capi.ExtUnaryArgops:
        if self.act == 0:
            return capi.no_argops()
        t = capi.UNARY.SIGMOID if self.act == 3 else capi.UNARY.RELU
        return capi.argops_cp(self.ldc, t, capi.UNARY_FLAG.BITMASK_2BYTEMULT if self.act == 2 else 0)

    def postops(self) -> capi.ExtBinaryPostops:
        return capi.postops_colbias(self.m, self.c_type) if self.colbias else capi.no_postops()

    def oracle_desc(self) -> pyoracle.GemmDesc:
        f = self.flags
        f |= {capi.BR_ADDRESS: GEMM_FLAG.BATCH_REDUCE_ADDRESS, capi.BR_OFFSET: GEMM_FLAG.BATCH_REDUCE_OFFSET,
              capi.BR_STRIDE: GEMM_FLAG.BATCH_REDUCE_STRIDE}.get(self.br_type, 0)
        f |= GEMM_FLAG.USE_XGEMM_EXT_ABI if self.ext else GEMM_FLAG.USE_XGEMM_ABI
        return pyoracle.GemmDesc(self.m, self.n, self.k, self.lda, self.ldb, self.ldc, self.a_type, self.b_type, self.c_type,
                                 self.comp_type, f, self.br_stride_a, self.br_stride_b, int(self.colbias), self.act)

    # ---- param construction over arbitrary buffers ----------------------------------------------
    def make_param(self, A, B, Cbuf, D=None, mask=None, offs=None, addr=None, brc=None, batch_index=0, S=None, SB=None):
        """A, B, Cbuf, D, mask: numpy arrays or torch tensors; returns (param, keepalive)."""
        p = capi.GemmExtParam() if self.ext else capi.GemmParam()
        keep = []
        if self.mxmx:
            p.a.tertiary = ptr(self.S if S is None else S) + batch_index * self.bs_s
            p.b.tertiary = ptr(self.SB if SB is None else SB) + batch_index * self.bs_sb
        if self.mx:                             # a.tertiary: the scales, or (ADDRESS mode) a list of per-block scale pointers
            sc = self.S if S is None else S
            if self.br_type == capi.BR_ADDRESS:
                p.a.tertiary = ptr(addr[2]) + batch_index * self.nbr * 8
            else:
                p.a.tertiary = ptr(sc) + batch_index * self.bs_s
        pa = ptr(A) + batch_index * self.bs_a
        pb = ptr(B) + batch_index * self.bs_b
        if self.br_type == capi.BR_ADDRESS:
            la, lb = addr[0], addr[1]
            p.a.primary = ptr(la) + batch_index * self.nbr * 8
            p.b.primary = ptr(lb) + batch_index * self.nbr * 8
        else:
            p.a.primary, p.b.primary = pa, pb
        if self.br_type == capi.BR_OFFSET:
            oa, ob = offs
            p.a.secondary, p.b.secondary = ptr(oa), ptr(ob)
        if self.br_type != capi.BR_NONE:
            cnt = brc if brc is not None else C.c_ulonglong(self.br_count)
            keep.append(cnt)
            p.op.tertiary = C.addressof(cnt)
        p.c.primary = ptr(Cbuf) + batch_index * self.bs_c
        if self.scf is not None:
            p.c.tertiary = C.addressof(self.scf)
        if self.ext:
            if self.colbias:
                p.d.primary = ptr(D) + batch_index * self.bs_d
            if self.act == 2:
                p.c.secondary = ptr(mask) + batch_index * self.mask_bytes
        return p, keep

    def host_address_lists(self, A: np.ndarray, B: np.ndarray, S=None):
        """Pointer lists for ADDRESS mode over host (or device, given base addresses) buffers."""
        la = np.zeros(self.batch * self.nbr, dtype=np.uint64)
        lb = np.zeros(self.batch * self.nbr, dtype=np.uint64)
        ls = np.zeros(self.batch * self.nbr, dtype=np.uint64)
        for b in range(self.batch):
            for r in range(self.nbr):
                la[b * self.nbr + r] = ptr(A) + b * self.bs_a + r * self.br_stride_a
                lb[b * self.nbr + r] = ptr(B) + b * self.bs_b + (self.nbr - 1 - r) * self.br_stride_b
                if self.mx:
                    ls[b * self.nbr + r] = ptr(self.S if S is None else S) + b * self.bs_s + r * self.s_elems
        return (la, lb, ls) if self.mx else (la, lb)

    # ---- executors ---------------------------------------------------------------------------------
    def run_oracle(self, fma=False):
        """Returns (C, mask) computed by the CPU restatement, batch element by batch element."""
        orc = pyoracle.oracle()
        Cbuf = self.C0.copy()
        mask = np.zeros(self.batch * self.mask_bytes, dtype=np.uint8) if self.act == 2 else None
        addr = self.host_address_lists(self.A, self.B) if self.br_type == capi.BR_ADDRESS else None
        d = self.oracle_desc()
        for b in range(self.batch):
            p, keep = self.make_param(self.A, self.B, Cbuf, self.D, mask, (self.offs_a, self.offs_b), addr, batch_index=b)
            orc.gemm(p, d, fma=fma)
        return Cbuf, mask

    def run_reference(self, jit=False):
        """Same through the real reference: its C reference kernel, or its CPU JIT when jit=True."""
        ref = pyoracle.reference()
        Cbuf = self.C0.copy()
        mask = np.zeros(self.batch * self.mask_bytes, dtype=np.uint8) if self.act == 2 else None
        addr = self.host_address_lists(self.A, self.B) if self.br_type == capi.BR_ADDRESS else None
        handle = None
        if jit:
            handle = (ref.dispatch_brgemm_ext(self.shape(), self.flags, 0, self.brcfg(), self.argops(), self.postops()) if self.ext
                      else ref.dispatch_brgemm(self.shape(), self.flags, 0, self.brcfg()))
            if not handle:
                return None, None
        for b in range(self.batch):
            p, keep = self.make_param(self.A, self.B, Cbuf, self.D, mask, (self.offs_a, self.offs_b), addr, batch_index=b)
            if jit:
                capi.Api.call(handle, p)
            elif self.ext:
                rc = ref.lib.xref_reference_gemm_ext(C.byref(p), self.shape(), self.flags, 0, self.brcfg(), self.argops(), self.postops())
                assert rc == 0
            else:
                rc = ref.lib.xref_reference_gemm(C.byref(p), self.shape(), self.flags, 0, self.brcfg())
                assert rc == 0
        return Cbuf, mask

    def dispatch(self, api: capi.Api) -> int:
        if self.ext:
            return api.dispatch_brgemm_ext(self.shape(), self.flags, 0, self.brcfg(), self.argops(), self.postops())
        if self.br_type == capi.BR_NONE:
            return api.dispatch_gemm(self.shape(), self.flags, 0)
        return api.dispatch_brgemm(self.shape(), self.flags, 0, self.brcfg())

    def run_gpu(self, batched=True):
        """Through the product library on cuda:0 (torch is plumbing for device memory only)."""
        import torch
        api = capi.load()
        dev = torch.device("cuda:0")

        def up(x):
            if UPLOAD_HOOK is not None:
                return None if x is None else UPLOAD_HOOK(x)
            return None if x is None else torch.from_numpy(x.view(np.int16) if x.dtype == np.uint16 else x).to(dev)
        A, B, Cbuf, D = up(self.A), up(self.B), up(self.C0.copy()), up(self.D)
        S = up(self.S) if (self.mx or self.mxmx) else None
        SB = up(self.SB) if self.mxmx else None
        mask = (up(np.zeros(self.batch * self.mask_bytes, dtype=np.uint8)) if UPLOAD_HOOK is not None else torch.zeros(self.batch * self.mask_bytes, dtype=torch.uint8, device=dev)) if self.act == 2 else None
        offs = (up(self.offs_a), up(self.offs_b))
        addr = None
        if self.br_type == capi.BR_ADDRESS:
            lists = self.host_address_lists(A, B, S)
            addr = tuple(up(x.view(np.int64)) for x in lists)
        handle = self.dispatch(api)
        assert handle, "dispatch returned NULL"
        if batched and self.batch > 1:
            p, keep = self.make_param(A, B, Cbuf, D, mask, offs, addr, S=S, SB=SB)
            sa = self.nbr * 8 if self.br_type == capi.BR_ADDRESS else self.bs_a
            sb = (self.nbr * 8 if self.br_type == capi.BR_ADDRESS else self.bs_b)
            if self.ext:
                api.hip_gemm_ext_batch_strided(handle, C.byref(p), self.batch, sa, sb, self.bs_c, self.bs_d, self.mask_bytes)
            else:
                api.hip_gemm_batch_strided(handle, C.byref(p), self.batch, sa, sb, self.bs_c)
        else:
            for b in range(self.batch):
                p, keep = self.make_param(A, B, Cbuf, D, mask, offs, addr, batch_index=b, S=S, SB=SB)
                capi.Api.call(handle, p)
        api.hip_sync()
        api.check()
        out = Cbuf.cpu().numpy()
        if self.c_type in (DT.BF16, DT.F16):
            out = out.view(np.uint16)
        return out, (mask.cpu().numpy() if mask is not None else None), handle

    def valid_region(self, Cbuf: np.ndarray) -> np.ndarray:
        """The m x n part of every batch element's C (padding rows between ldc and m are don't-care)."""
        c = Cbuf.reshape(self.batch, -1)[:, : self.ldc * self.n].reshape(self.batch, self.n, self.ldc)
        return c[:, :, : self.m]

    def valid_mask_bits(self, mask: np.ndarray) -> np.ndarray:
        bits = np.unpackbits(mask.reshape(self.batch, self.n, self.mask_ld // 8), axis=2, bitorder="little")
        return bits[:, :, : self.m]


def compress_by_bitmask(a_mem: np.ndarray):
    """LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK operands from a dense A in MEMORY order (f32: [k][m]; 16-bit: the VNNI image [k/2][m][2]):
    (non-zeros in that order, one bit per element LSB first) -- what samples/xgemm/gemm_kernel.c:107-212 builds.  -0.0 counts as zero there
    (the driver compares the VALUE with 0) and so it does here."""
    if a_mem.dtype == np.float32:
        nz = a_mem != 0.0
    else:
        nz = (a_mem & 0x7fff) != 0
    return np.ascontiguousarray(a_mem[nz]), np.packbits(nz, bitorder="little")


def sparsify(rng: np.random.Generator, a_mem: np.ndarray, frac: float) -> np.ndarray:
    out = a_mem.copy()
    out[rng.random(out.size) < frac] = 0
    return out


def mx6_operands(rng: np.random.Generator, k: int, ld: int, nbr: int):
    """An MXBF6 / MXHF6 operand: nbr blocks of [k/4][ld][3 bytes] (four 6-bit values of a row per k-group; every bit pattern is a number) and
    its E8M0 scales [k/32][ld] in a narrow band around 1 [ref: generator_gemm_reference_impl.c:2680-2727]."""
    assert k % 32 == 0 and (ld * 6) % 8 == 0
    return rng.integers(0, 256, nbr * (k // 4) * ld * 3).astype(np.uint8), rng.integers(124, 131, nbr * (k // 32) * ld).astype(np.uint8)
