"""The reference's OWN sample drivers as the integration test-suite.

oracle/Makefile target `drivers` compiles the unmodified sources under /root/reference/samples (where they lie) against THIS
repository's include/libxsmm.h + include/libxsmm_utils.h and links them to libxsmm_amd.so; the binaries (oracle/_ref/drivers,
git-ignored, shipped to the GPU box like every other built file) generate their own data, dispatch through the public API, run
the kernels on the MI355X (their buffers come from libxsmm_aligned_malloc = pinned, device-visible memory) and compare against
their built-in gold loops.  A driver is its own judge: exit code 0 and no failure marker in its output.
"""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DRV = os.path.join(HERE, "..", "oracle", "_ref", "drivers")
FAIL_MARKERS = ("FAILED", "ERROR", "JIT failed", "failed. Bailing", "not supported")


def run(binary, *args, timeout=120, env_extra=None):
    exe = os.path.join(DRV, binary)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (make -C oracle drivers needs /root/reference)")
    env = dict(os.environ, LIBXSMM_VERBOSE="0", **(env_extra or {}))
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env)
    out = r.stdout + r.stderr
    if os.environ.get("DRIVERS_SHOW"):
        print(f"\n$ {binary} {' '.join(str(a) for a in args)} -> rc={r.returncode}\n" + "\n".join(out.splitlines()[-12:]))
    return r.returncode, out


def check(binary, *args, **kw):
    rc, out = run(binary, *args, **kw)
    assert rc == 0, f"{binary} {args}: exit {rc}\n{out[-2000:]}"
    bad = [m for m in FAIL_MARKERS if m in out]
    assert not bad, f"{binary} {args}: {bad}\n{out[-2000:]}"
    return out


def write_mtx(path, rowptr, colidx, vals, ncols, by_column=False):
    ent = [(r, int(colidx[z]), float(vals[z])) for r in range(len(rowptr) - 1) for z in range(rowptr[r], rowptr[r + 1])]
    if by_column:
        ent.sort(key=lambda e: (e[1], e[0]))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%\n")
        f.write(f"{len(rowptr) - 1} {ncols} {len(ent)}\n")
        for r, c, v in ent:
            f.write(f"{r + 1} {c + 1} {v!r}\n")


@pytest.fixture(scope="module")
def edge_mtx(tmp_path_factory):
    """The 35x35 / 108 non-zero EDGE stiffness operator of the committed golden vectors, as .mtx files (row- and column-sorted)."""
    d = dict(np.load(os.path.join(HERE, "golden", "reference_vectors.npz")))
    base = tmp_path_factory.mktemp("mtx")
    csr, csc = str(base / "edge_csr.mtx"), str(base / "edge_csc.mtx")
    write_mtx(csr, d["spcsr_edge_rowptr"], d["spcsr_edge_colidx"], d["spcsr_edge_vals"], 35)
    write_mtx(csc, d["spcsr_edge_rowptr"], d["spcsr_edge_colidx"], d["spcsr_edge_vals"], 35, by_column=True)
    return csr, csc


# samples/xgemm/gemm_kernel.c -- A B comp C  M N K lda ldb ldc  alpha beta  alignA alignC  transA transB  vnniA vnniB vnniC  prefetch  br  brsize brunroll reps tilecfg
GEMM_CASES = [
    "F32 F32 F32 F32 23 23 23 23 23 23 1 0 0 0 0 0 0 0 0 nopf nobr 1 0 2 0",           # BASELINE config #1
    "F32 F32 F32 F32 32 32 32 32 32 32 1 0 0 0 0 0 0 0 0 nopf strdbr 8 0 2 0",         # config #2's kernel as a stride-BRGEMM
    "F32 F32 F32 F32 32 32 32 32 32 32 1 1 0 0 0 0 0 0 0 nopf addrbr 4 0 2 0",
    "F32 F32 F32 F32 64 48 40 64 40 64 1 1 0 0 0 0 0 0 0 nopf offsbr 3 0 2 0",
    "F32 F32 F32 F32 17 9 31 33 33 24 1 1 0 0 1 0 0 0 0 nopf nobr 1 0 2 0",            # A transposed, padded leading dimensions
    "F32 F32 F32 F32 16 16 16 16 16 16 1 0 0 0 0 1 0 0 0 nopf nobr 1 0 2 0",           # B transposed
    "F64 F64 F64 F64 23 23 23 23 23 23 1 1 0 0 0 0 0 0 0 nopf nobr 1 0 2 0",
    "BF16 BF16 F32 F32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf strdbr 4 0 2 0",       # VNNI-2 A
    "BF16 BF16 F32 BF16 64 64 64 64 64 64 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "BF16 BF16 F32 BF16 32 32 32 32 32 32 1 0 0 0 0 0 1 0 1 nopf nobr 1 0 2 0",        # VNNI-2 C
    "F16 F16 F32 F16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf strdbr 4 0 2 0",        # IEEE halves, f32 accumulation
    "F16 F16 F32 F32 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "I8 I8 I32 I32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "U8 I8 I32 I32 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf strdbr 2 0 2 0",
    "BF8 BF8 F32 F32 32 32 64 32 64 32 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "HF8 HF8 F32 F32 64 64 64 64 64 64 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    # low-bit weights x 8-bit activations: 4-bit minus zero points (interleaved), 2-bit (interleaved), 1-bit
    "U4 U8 I32 I32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "I2 U8 I32 I32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "I2 I8 I32 I32 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "I1 I8 I32 I32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "I1 U8 I32 I32 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf strdbr 3 0 2 0",
    # microscaling formats: both operands MX (E2M1, E5M2, E4M3, the 6-bit E3M2 / E2M3) with f32 or MX-typed C (the driver checks C's data AND scales),
    # MXFP4 weights x bf16, MXFP4 weights (interleaved) x i8
    "MXFP4 MXFP4 F32 F32 64 64 64 64 64 64 1 0 0 0 0 1 1 1 0 nopf nobr 1 0 2 0",
    "MXFP4 MXFP4 F32 MXFP4 64 64 64 64 64 64 1 0 0 0 0 1 1 1 0 nopf nobr 1 0 2 0",
    "MXBF8 MXBF8 F32 F32 64 64 64 64 64 64 1 0 0 0 0 1 1 1 0 nopf nobr 1 0 2 0",
    "MXBF8 MXBF8 F32 MXBF8 64 64 64 64 64 64 1 0 0 0 0 1 1 1 0 nopf nobr 1 0 2 0",
    "MXHF8 MXHF8 F32 F32 32 32 64 32 32 32 1 1 0 0 0 1 1 1 0 nopf strdbr 2 0 2 0",
    "MXBF6 MXBF6 F32 F32 64 64 64 64 64 64 1 0 0 0 0 1 1 1 0 nopf nobr 1 0 2 0",
    "MXHF6 MXHF6 F32 F32 32 32 64 32 32 32 1 1 0 0 0 1 1 1 0 nopf nobr 1 0 2 0",
    "MXFP4 BF16 F32 BF16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "MXFP4 I8 I32 F32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "MXFP4 I8 I32 BF16 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    # further types of the dense loop: BF32 (f32 storage, bf16 precision), 16-bit integers, 8-bit floats with a result of their own type,
    # 8-bit float weights x bf16, row-scaled i8 weights x bf16
    "BF32 BF32 F32 F32 64 64 64 64 64 64 1 0 0 0 0 0 0 0 0 nopf nobr 1 0 2 0",
    "I16 I16 I32 I32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "BF8 BF8 F32 BF8 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "HF8 HF8 F32 HF8 64 64 64 64 64 64 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "BF8 BF16 F32 BF16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",
    "HF8 BF16 F32 F32 64 64 64 64 64 64 1 1 0 0 0 0 1 0 0 nopf strdbr 2 0 2 0",
    "I8 BF16 F32 BF16 64 64 64 64 64 64 1 0 0 0 0 0 0 0 0 nopf nobr 1 0 2 0",
    "I8 BF16 F32 F32 64 64 64 64 64 64 1 1 0 0 0 0 0 0 0 nopf nobr 1 0 2 0",
    "F16 F16 F16 F16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",          # comp_type F16: a rounding after every product
    # round 6: VNNI_C on IEEE-half results (VNNI-2) and on results of the operands' 8-bit float type (VNNI-4); 8-bit integers -> f32 WITHOUT the VNNI_A flag
    "F16 F16 F32 F16 32 32 32 32 32 32 1 0 0 0 0 0 1 0 1 nopf nobr 1 0 2 0",
    "BF8 BF8 F32 BF8 32 32 64 32 64 32 1 0 0 0 0 0 1 0 1 nopf nobr 1 0 2 0",
    "HF8 HF8 F32 HF8 32 16 64 32 64 32 1 0 0 0 0 0 1 0 1 nopf strdbr 2 0 2 0",
    "I8 I8 I32 F32 32 32 64 32 64 32 1 0 0 0 0 0 0 0 0 nopf nobr 1 0 2 0",
    "U8 I8 I32 F32 64 32 64 64 64 64 1 1 0 0 0 0 0 0 0 nopf strdbr 2 0 2 0",
    "F16 F16 F16 F32 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf strdbr 2 0 2 0",
    "F16 F16 IMPLICIT F16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0",     # IMPLICIT = f32 sums here (the driver asks this library's libxsmm_cpuid)
    # "spmm": A sparsified to the given fraction and handed over as (non-zeros, bitmask) -- LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK
    "F32 F32 F32 F32 64 64 64 64 64 64 1 0 0 0 0 0 0 0 0 nopf spmm 0.5 0 2 0",
    "F32 F32 F32 F32 128 48 256 128 256 128 1 1 0 0 0 0 0 0 0 nopf spmm 0.9 0 2 0",
    "BF16 BF16 F32 BF16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf spmm 0.5 0 2 0",
    "BF16 BF16 F32 F32 128 32 128 128 128 128 1 1 0 0 0 0 1 0 0 nopf spmm 0.75 0 2 0",
]


@pytest.mark.parametrize("args", GEMM_CASES)
def test_reference_gemm_driver(args):
    check("gemm_kernel", *args.split())


# the same source built with -DUSE_GEMM_EXT_FRONTEND (the reference's gemm_kernel_fused): ... reps tilecfg binary_postop(1 = column bias add)
# unary_postop(1 ReLU, 2 ReLU + bitmask, 3 sigmoid)
@pytest.mark.parametrize("args", [
    "BF16 BF16 F32 BF16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf strdbr 4 0 2 0 1 1",     # BASELINE config #5's kernel: bias + ReLU fused
    "BF16 BF16 F32 BF16 64 64 64 64 64 64 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0 1 2",       # ReLU with bitmask output
    "BF16 BF16 F32 BF16 32 32 32 32 32 32 1 0 0 0 0 0 1 0 0 nopf addrbr 3 0 2 0 0 3",     # sigmoid
    "F32 F32 F32 F32 64 48 40 64 40 64 1 1 0 0 0 0 0 0 0 nopf strdbr 3 0 2 0 1 1",
    "F32 F32 F32 F32 23 23 23 23 23 23 1 0 0 0 0 0 0 0 0 nopf nobr 1 0 2 0 1 2",
    "F32 F32 F32 F32 32 32 32 32 32 32 1 0 0 0 0 0 0 0 0 nopf offsbr 4 0 2 0 0 3",
    "BF16 BF16 F32 F32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0 1 0",
    # round 6: the seven further precisions the reference's kernel tests keep fusion ON for (samples/xgemm/kernel_test/generate_gemm_test_scripts.tpl:77)
    "F16 F16 F32 F16 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf strdbr 4 0 2 0 1 1",
    "F16 F16 F32 F32 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf nobr 1 0 2 0 1 2",
    "BF8 BF8 F32 F32 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0 1 1",
    "BF8 BF8 F32 BF8 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0 1 2",
    "HF8 HF8 F32 F32 32 32 64 32 64 32 1 1 0 0 0 0 1 0 0 nopf strdbr 2 0 2 0 1 3",
    "HF8 HF8 F32 HF8 64 64 64 64 64 64 1 0 0 0 0 0 1 0 0 nopf nobr 1 0 2 0 1 1",
    "BF32 BF32 F32 F32 64 64 64 64 64 64 1 0 0 0 0 0 0 0 0 nopf nobr 1 0 2 0 1 2",
    "BF32 BF32 F32 F32 23 23 23 23 23 23 1 1 0 0 0 0 0 0 0 nopf nobr 1 0 2 0 0 3",
])
def test_reference_fused_gemm_driver(args):
    check("gemm_kernel_fused", *args.split())


# samples/xgemm_sparse/spmm_kernel.c -- A B comp C  M N K m_blocks sparsity bk bn beta transA transB vnniA vnniB vnniC reps
@pytest.mark.parametrize("args", [
    "BF16 BF16 F32 BF16 64 64 256 16 0.75 32 16 0 0 0 1 0 0 2",            # BASELINE config #4's kernel (random instead of 2:8 pattern)
    "BF16 BF16 F32 BF16 64 64 256 8 0.5 32 32 1 0 0 1 0 0 2",
    "F32 F32 F32 F32 32 32 64 4 0.75 16 4 0 0 0 0 0 0 2",
    "U8 I8 I32 I32 64 64 256 16 0.75 32 16 0 0 0 1 0 0 2",                  # 8-bit integers on v_mfma_i32_16x16x32_i8
    "I8 U8 I32 I32 64 64 256 8 0.5 32 32 1 0 0 1 0 0 2",
    "U8 I8 I32 I32 32 32 64 4 0.75 8 8 0 0 0 1 0 0 2",                      # small blocks: generic kernel
    "BF16 BF16 F32 BF16 64 64 256 8192 0.75 32 16 0 0 0 1 0 0 2",            # BASELINE config #4 at full size (m_blocks = 8192)
    "U8 I8 I32 I32 64 64 256 8192 0.75 32 16 0 0 0 1 0 0 2",
])
def test_reference_bcsc_driver(args):
    check("spmm_kernel", *args.split())


# samples/xgemm_norm_packed/*.c -- M N K N_CRUNS reps file
@pytest.mark.parametrize("binary,which", [("asparse_packed_csr", 0), ("asparse_packed_csr_f32", 0), ("bsparse_packed_csr", 0), ("bsparse_packed_csr_f32", 0),
                                          ("bsparse_packed_csc", 1), ("bsparse_packed_csc_f32", 1)])
def test_reference_packed_drivers(binary, which, edge_mtx):
    mnk = ("35", "9", "35") if binary.startswith("asparse") else ("9", "35", "35")
    out = check(binary, *mnk, "8", "2", edge_mtx[which])
    # these drivers print their error but do not assert it ("parity only weakly pinned"): assert it here
    errs = [float(x) for x in re.findall(r"max error: ([0-9.eE+-]+)", out)]
    assert errs and max(errs) <= (1e-5 if binary.endswith("_f32") else 1e-6), out[-1500:]


# BASELINE config #3 at its full sizes: the 35 x 35 operator, N = 35 quantities, packed width P = 4096 and 65 536
@pytest.mark.parametrize("binary,P", [("asparse_packed_csr_f32", 4096), ("asparse_packed_csr_f32", 65536), ("asparse_packed_csr", 4096)])
def test_reference_packed_driver_at_baseline_size(binary, P, edge_mtx):
    out = check(binary, "35", "35", "35", str(P), "2", edge_mtx[0])
    errs = [float(x) for x in re.findall(r"max error: ([0-9.eE+-]+)", out)]
    assert errs and max(errs) <= (1e-5 if binary.endswith("_f32") else 1e-6), out[-1500:]


# samples/xgemm_sparse_Ainregs/pyfr_driver_asp_reg.c -- file N reps [beta]
@pytest.mark.parametrize("beta,N", [(0, 4800), (1, 4800), (0, 1 << 20)])
def test_reference_fsspmdm_driver(beta, N, edge_mtx):
    # The driver walks N in steps of FSSPMDM_NBLOCK (default 48) WITHOUT a remainder step [pyfr_driver_asp_reg.c:235-237, 383-385]: with
    # N % block != 0 its last call reads and writes past the end of B and C (on any backend).  2^20 is not a multiple of 48, so the
    # BASELINE-size line runs with the driver's own knob set to 64; 4800 = 100 * 48 keeps the default.
    env = {"FSSPMDM_NBLOCK": "64"} if N % 48 else None
    out = check("pyfr_driver_asp_reg", edge_mtx[0], str(N), "2", beta, env_extra=env, timeout=600)
    errs = [float(x) for x in re.findall(r"\(libxsmm vs\. gold\): abs=([0-9.eE+-]+)", out)]
    assert errs and max(errs) <= 1e-6, out[-1500:]


# samples/eltwise/*.c
@pytest.mark.parametrize("args", [
    "1 0 F32 F32 F32 64 48 64 64", "1 0 BF16 F32 BF16 64 48 70 72", "1 0 F32 F32 BF16 33 17 40 36", "3 0 F32 F32 F32 64 48 64 64",
    "4 0 F32 F32 F32 32 32 32 32", "7 0 F32 F32 F32 64 48 64 64", "9 0 BF16 F32 BF16 64 48 64 64", "11 0 F32 F32 F32 64 48 64 64",
    "13 0 F32 F32 F32 64 48 64 64", "14 0 F32 F32 F32 64 48 64 64", "15 0 F32 F32 F32 64 48 64 64", "16 0 F32 F32 F32 64 48 64 64",
    "17 0 F32 F32 F32 64 48 64 64", "1 0 F16 F32 F16 64 48 64 64", "13 0 BF8 F32 BF8 64 48 64 64", "3 0 HF8 F32 HF8 33 17 40 36", "1 0 F32 F32 HF8 64 48 64 64",
    "9 0 F16 F32 F16 64 48 64 64", "8 0 F32 F32 F32 64 48 64 64", "10 0 F32 F32 F32 64 48 64 64", "12 0 F32 F32 F32 64 48 64 64", "27 0 F32 F32 F32 64 48 64 64", "27 0 BF16 F32 BF16 64 48 64 64",
    "7 0 BF16 F32 BF16 64 48 64 64", "11 0 BF16 F32 BF16 64 48 64 64", "17 0 BF8 F32 BF8 64 48 64 64", "15 0 HF8 F32 HF8 64 48 64 64", "1 1 F32 F32 F32 64 48 64 64", "1 2 F32 F32 F32 64 48 64 64", "1 3 F32 F32 F32 64 48 64 64", "2 0 F32 F32 F32 64 48 64 64",
    "64 0 F32 F32 BF16 64 48 64 64", "65 0 F32 F32 BF16 64 48 64 64", "65 0 F32 F32 BF16 33 17 40 36",      # round 6: DECOMP_FP32_TO_BF16X2 / X3
])
def test_reference_unary_driver(args):
    check("eltwise_unary_simple", *args.split())


@pytest.mark.parametrize("args", [
    "1 0 F32 F32 F32 F32 64 48 64 64", "2 0 BF16 BF16 F32 BF16 64 48 64 64", "3 1 F32 F32 F32 F32 64 48 64 64", "4 2 F32 F32 F32 F32 64 48 64 64",
    "5 0 F32 F32 F32 F32 64 48 64 64", "1 0 F16 F16 F32 F16 64 48 64 64", "2 0 BF8 BF8 F32 BF8 64 48 64 64", "1 2 HF8 HF8 F32 HF8 33 17 40 36", "9 4 F32 F32 F32 F32 33 17 40 36", "10 5 F32 F32 F32 F32 64 48 64 64", "27 0 F32 F32 F32 F32 64 48 64 64", "28 0 F32 F32 F32 F32 64 48 64 64", "29 0 BF16 BF16 F32 BF16 64 48 64 64",
    "30 0 F32 F32 F32 F32 64 48 64 64", "31 0 F32 F32 F32 F32 64 48 64 64", "32 0 F32 F32 F32 F32 64 48 64 64", "5 3 BF16 BF16 F32 BF16 64 48 64 64", "4 0 F16 F16 F32 F16 64 48 64 64", "1 6 BF16 F32 F32 F32 64 48 64 64", "1 3 F32 F32 F32 BF16 64 48 64 64",
])
def test_reference_binary_driver(args):
    check("eltwise_binary_simple", *args.split())


@pytest.mark.parametrize("args", ["D F 0 F32 F32 F32 64 48 64 64", "D F 1 F32 F32 F32 64 48 64 64", "D F 1 BF16 F32 BF16 64 48 64 64", "D B 1 F32 F32 F32 64 48 64 64",
                                  "L F 0 F32 F32 F32 64 48 64 64", "E F 0 F32 F32 F32 64 48 64 64", "L F 1 F32 F32 F32 64 48 64 64", "L B 1 F32 F32 F32 64 48 64 64",
                                  "E F 0 BF16 F32 BF16 64 48 64 64", "E B 0 F32 F32 F32 64 48 64 64", "D B 1 BF16 F32 BF16 64 48 64 64", "D F 1 F16 F32 F16 64 48 64 64",
                                  "D F 0 BF8 F32 BF8 64 48 64 64", "D F 1 HF8 F32 HF8 64 48 64 64", "L F 1 BF16 F32 BF16 64 48 64 64"])
def test_reference_relu_driver(args):
    check("eltwise_unary_relu", *args.split())


# every transform letter of the driver (T transpose; R / S / F / V / W / G / Q / H / B / C / D / I / N / M the NORM <-> VNNI2 / VNNI4 / VNNI8 and VNNI <-> VNNI-T
# forms; X / Y / Z the padded ones), in each element width it accepts
@pytest.mark.parametrize("args", ["T F32 64 48 64 48", "T F32 33 17 40 20", "T BF16 64 48 64 48", "T F64 16 24 16 24", "V BF16 64 48 64 64", "R BF16 64 48 64 64",
                                  "T I8 64 48 64 48", "T BF8 64 48 64 48", "T I16 64 48 64 48", "T F16 64 48 64 48", "T I32 64 48 64 48", "T I64 64 48 64 48", "R F16 64 48 64 64",
                                  "S I8 64 48 64 64", "S BF16 64 48 64 64", "F I8 64 48 64 64", "F BF16 64 48 64 64", "V I16 64 48 64 64", "W I8 64 48 64 64", "W BF16 64 48 64 64",
                                  "G I8 64 48 64 64", "G BF16 64 48 64 64", "Q BF16 64 48 64 64", "H BF16 64 48 64 64", "B BF16 64 48 64 64", "C BF16 64 48 64 64", "D BF16 64 48 64 64",
                                  "I BF16 64 48 64 64", "N I8 64 48 64 64", "M I8 64 48 64 64", "X BF16 64 48 64 64", "Y BF16 64 48 64 64", "Z BF16 64 48 64 64", "X I8 64 48 64 64",
                                  "Y I8 64 48 64 64", "Z I8 64 48 64 64"])
def test_reference_transform_driver(args):
    check("eltwise_unary_transform", *args.split())


# M N ldi ldo gather(0)/scatter(1) rows(0)/cols(1)/offs(2) 16-bit-dtype 64-bit-index iters
@pytest.mark.parametrize("args", ["64 48 64 64 0 1 0 0 1", "64 48 64 64 0 0 0 1 1", "64 48 64 64 1 1 0 0 1", "64 48 64 64 0 2 0 1 1", "64 48 64 64 0 1 1 0 1"])
def test_reference_gather_scatter_driver(args):
    check("eltwise_unary_gather_scatter", *args.split())


# samples/eltwise/eltwise_unary_dropout.c -- F/B bitmask prec_in prec_out M N ldi ldo: its gold loop draws libxsmm_cpuid_vlen32() rows at a time
@pytest.mark.parametrize("args", ["F 1 F32 F32 64 64 64 64", "F 0 F32 F32 40 13 48 40", "F 1 BF16 BF16 64 48 64 64", "F 1 F32 BF16 33 17 40 36", "B 1 F32 F32 64 64 64 64", "B 1 BF16 BF16 50 20 56 52",
                                  "F 1 F16 F16 64 48 64 64", "F 1 BF8 BF8 64 48 64 64", "B 1 HF8 HF8 64 48 64 64", "F 0 BF16 F32 64 48 64 64"])
def test_reference_dropout_driver(args):
    check("eltwise_unary_dropout", *args.split())


# samples/eltwise/eltwise_unary_reduce.c allocates with plain malloc(): synchronous reductions stage host operands
# M N ldi reduce_x reduce_x2 reduce_rows op(0 add, 1 max) dtype n_cols_idx idx_type record_idx reduce_on_outputs iters
@pytest.mark.parametrize("args", ["64 48 64 1 0 1 0 F32 0 0 0 0 1", "64 48 64 1 0 0 0 F32 0 0 0 0 1", "64 48 64 1 1 1 0 F32 0 0 0 0 1", "64 48 64 1 0 1 1 F32 0 0 0 0 1",
                                  "64 48 64 1 0 0 0 BF16 0 0 0 0 1", "33 17 40 1 1 0 0 F32 0 0 0 0 1",
                                  "64 48 64 1 0 0 0 F32 12 0 0 0 1", "64 48 64 1 0 0 1 F32 12 1 1 0 1", "64 48 64 1 0 0 2 F32 12 0 1 0 1", "64 48 64 1 0 0 1 BF16 9 1 0 0 1",
                                  "64 48 64 1 0 0 1 F32 0 0 1 0 1", "64 48 64 0 1 1 0 F32 0 0 0 0 1", "64 48 64 0 1 0 0 F32 0 0 0 0 1", "64 48 64 1 0 1 1 BF16 0 0 0 0 1",
                                  "64 48 64 1 0 0 2 F32 0 0 0 0 1", "64 48 64 1 0 1 2 F32 0 0 0 0 1", "64 48 64 1 0 0 0 F16 0 0 0 0 1", "64 48 64 1 0 0 0 BF8 0 0 0 0 1",
                                  "64 48 64 1 0 0 0 F32 0 0 0 1 1", "64 48 64 1 1 1 0 F32 0 0 0 1 1"])          # listed columns (n_cols_idx), 4 / 8-byte indices, recorded arg-max / arg-min
def test_reference_reduce_driver(args):
    check("eltwise_unary_reduce", *args.split())


@pytest.mark.parametrize("args", ["1 0 F32 F32 IMPLICIT F32 F32 64 48 64 64", "1 0 BF16 BF16 IMPLICIT F32 BF16 64 48 64 64", "1 4 F32 F32 IMPLICIT F32 F32 33 17 40 36"])
def test_reference_ternary_driver(args):
    check("eltwise_ternary_simple", *args.split())


# samples/equation/equation_simple.c -- M N ld datatype_mode(0 f32, 1 bf16) iters: five-argument element-wise + reduce/broadcast trees
# (not run: equation_simple_layernorm is BF8-only; equation_bf16_x3_split_f32 reads, as arguments, buffers that DUMP nodes of the same tree
# write -- its result depends on the reference's node scheduling)
@pytest.mark.parametrize("args", ["64 48 64 0 2", "64 48 64 1 2", "33 17 40 0 2"])
def test_reference_equation_driver(args):
    check("equation_simple", *args.split())


# samples/equation/equation_softmax.c -- S1 S2 S3 datatype_mode(0 f32, 1 bf16) pass(1 fwd, 2 bwd, 3 both) iters: reductions to a scalar, scalar
# broadcasts, exp, reciprocal and UNARY_DUMP (the forward keeps exp(x - max) for the backward) in one tree.  The driver prints its norms
# without judging them: the check-norm (relative L2 against its intrinsics implementation) is asserted here.
@pytest.mark.parametrize("args,bound", [("4 8 16 0 3 2", 1e-5), ("8 16 32 0 1 2", 1e-5), ("8 16 32 1 1 2", 2e-2), ("4 8 16 1 2 2", 2e-2)])
def test_reference_softmax_equation_driver(args, bound):
    out = check("equation_softmax", *args.split())
    norms = [float(x) for x in re.findall(r"Check-norm\s*:\s*([0-9.eE+-]+)", out)]
    assert norms and max(norms) <= bound, out[-2000:]


# samples/equation/equation_layernorm.c -- S1 S2 S3 datatype_mode(0 f32, 1 bf16) pass(1 fwd, 2 bwd, 3 both) iters: per row block, X / X^2 column
# reductions + a row reduction as plain TPPs, then six equations -- the affine forward, dgamma / dbeta accumulated in place, the ds / db dot
# products (BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD heads, results on the caller's stack) and din.  Mean, variance and the scalar factors live on
# the driver's stack: every call is synchronous.  The driver judges itself (check-norm against its scalar loops, bound 0.007).  S3 = 20: rows that
# are no multiple of 8, the TPP chain instead of the generated kernels.
@pytest.mark.parametrize("args", ["4 8 64 0 3 1", "4 8 64 1 3 1", "2 5 48 0 3 1", "3 4 20 0 3 1", "3 4 20 1 3 1", "8 3 256 1 2 1"])
def test_reference_layernorm_equation_driver(args):
    out = check("equation_layernorm", *args.split())
    norms = [float(x) for x in re.findall(r"Check-norm\s*:\s*([0-9.eE+-]+)", out)]
    assert len(norms) >= (1 if args.split()[4] == "1" else 3) and max(norms) <= 0.007, out[-2000:]


# equation_relu.c (M N ld datatype_mode): ReLU with bitmask as the HEAD of a tree (mask through output.secondary); equation_splitSGD.c
# (M N ld iters): UNZIP(MULADD(grad, lr, ZIP(lo, hi))) -- 16-bit halves zipped to f32, updated, split again; lr is a host scalar
@pytest.mark.parametrize("binary,args", [("equation_relu", "64 48 64 0"), ("equation_relu", "64 48 64 1"), ("equation_relu", "33 17 40 0"),
                                         ("equation_splitSGD", "64 48 64 2"), ("equation_splitSGD", "33 17 40 2")])
def test_reference_side_channel_equation_drivers(binary, args):
    out = check(binary, *args.split())
    norms = [float(x) for x in re.findall(r"Check-norm\s*:\s*([0-9.eE+-]+)", out)]
    assert norms and max(norms) == 0.0, out[-2000:]


# samples/xgemm_norm_packed/dense_packed{ac,bc}rm.c (the reference's tests/packed.sh) -- M N K beta reps: libxsmm_create_packed_gemm_ac_rm / _bc_rm
@pytest.mark.parametrize("binary", ["dense_packedacrm", "dense_packedacrm_f32", "dense_packedbcrm", "dense_packedbcrm_f32"])
@pytest.mark.parametrize("args", ["9 81 35 0.0 2", "9 81 35 1.0 2", "9 35 81 1.0 2"])
def test_reference_dense_packed_drivers(binary, args):
    out = check(binary, *args.split())
    errs = [float(x) for x in re.findall(r"max\. error: ([0-9.eE+-]+)", out)]
    assert errs and max(errs) <= (1e-4 if binary.endswith("_f32") else 1e-6), out[-1500:]


# samples/xgemm_packed/gemm_packed_kernel.c -- A B comp C  M N K lda ldb ldc  R  alpha beta  transA transB  reps: libxsmm_create_packed_gemm
@pytest.mark.parametrize("args", ["F32 F32 F32 F32 9 9 9 9 9 9 64 1 0 0 0 2", "F32 F32 F32 F32 4 7 5 4 5 4 24 1 1 0 0 2", "F64 F64 F64 F64 9 9 9 9 9 9 16 1 1 0 0 2"])
def test_reference_packed_gemm_driver(args):
    check("gemm_packed_kernel", *args.split())


def test_reference_hello_world():
    """samples/hello/hello.c: 1000 f64 13x5x7 products accumulated into one C, every matrix from plain malloc() -- synchronous GEMM
    calls stage host operands, so LIBXSMM's first example runs as it is."""
    check("hello")


def test_reference_threadsafety_test():
    """tests/threadsafety.c: concurrent dispatch of random shapes from all OpenMP threads, registry queries, init/finalize cycles."""
    env_threads = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "8"
    try:
        check("threadsafety", timeout=300)
    finally:
        if env_threads is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = env_threads


def test_reference_matdiff_test():
    check("matdiff")


# samples/eltwise/eltwise_unary_quantization.c -- F32 I8|I16|I32 M N ldi ldo skip_scf_cvt signed_sat: QUANT and DEQUANT TPPs
@pytest.mark.parametrize("args", ["F32 I8 64 48 64 64 0 0", "F32 I8 64 48 64 64 0 1", "F32 I16 33 17 40 36 0 1", "F32 I32 64 48 64 64 1 0", "F32 I8 64 48 64 64 1 1"])
def test_reference_quantization_driver(args):
    out = check("eltwise_unary_quantization", *args.split())
    assert out.count("SUCCESS") == 2, out[-1500:]


# samples/equation/equation_gather_reduce.c -- M N ld datatype_mode idx_type(0: 32-bit, 1: 64-bit) iters: reduce_cols(gather_cols(X)) as one
# equation, the GATHER node reading its indices from inputs[0].secondary (argument lists of equation_test/equation_gather_reduce.sh)
@pytest.mark.parametrize("args", ["37 21 40 0 0 2", "64 32 64 0 1 2", "64 32 64 1 0 2", "17 5 20 1 1 2"])
def test_reference_gather_reduce_equation_driver(args):
    check("equation_gather_reduce", *args.split())


# samples/equation/equation_gather_dot.c, equation_gather_bcstmul_add.c -- cols M numidx idxblk iters (equation_test/*.sh: "<cols> <M> 256 16 0"):
# KV-cache style look-ups as TPP sequences, a two-vector dot-product equation whose head reduces to 1 x 1, gather + GEMM
@pytest.mark.parametrize("exe", ["equation_gather_dot", "equation_gather_bcstmul_add"])
@pytest.mark.parametrize("args", ["1024 48 256 16 0", "2048 80 64 16 2"])
def test_reference_gather_kvcache_equation_drivers(exe, args):
    check(exe, *args.split())


# samples/equation/equation_matmul.c -- n_tensors {m n ld blocks} x n_tensors datatype_mode(0 f32, 1 bf16) fusion(0 none, 1 relu, 2 sigmoid) iters:
# MATMUL / BRGEMM nodes below and above element-wise nodes (binary and accumulate-in-place ternary forms, strided argument sets, bf16 A in
# VNNI layout, bias + activation on top of a BRGEMM).  The driver prints its norms without judging them: asserted here.
# (the driver sizes the scratch of its own gold code as m_C * m_D BYTES for m_C * n_D floats -- equation_matmul.c:345 -- so the shapes keep k >= 4 n)
@pytest.mark.parametrize("args,bound", [("5 32 16 32 1 32 16 32 1 32 64 32 4 64 16 64 4 32 16 32 1 0 0 2", 1e-5),
                                        ("5 32 16 32 1 32 16 32 1 32 64 32 4 64 16 64 4 32 16 32 1 0 2 2", 1e-5),
                                        ("5 64 16 64 1 64 16 64 1 64 64 64 3 64 16 64 3 64 16 64 1 1 1 2", 1e-2)])
def test_reference_matmul_equation_driver(args, bound):
    out = check("equation_matmul", *args.split())
    norms = [float(x) for x in re.findall(r"Check-norm\s*:\s*([0-9.eE+-]+)", out)]
    assert len(norms) >= 3 and max(norms) <= bound, out[-3000:]


# samples/eltwise/eltwise_unary_quantization_to_mxfp4.c / _to_mxbf8.c -- M N ldi ldo: bf16 -> block-scaled 4-bit / 8-bit floats with E8M0
# scales in out.secondary, compared byte by byte with the drivers' gold code (data and scales)
@pytest.mark.parametrize("exe", ["eltwise_unary_quantization_to_mxfp4", "eltwise_unary_quantization_to_mxbf8"])
@pytest.mark.parametrize("args", ["64 16 64 64", "128 33 160 256", "1024 64 1024 1024"])
def test_reference_mx_quantization_drivers(exe, args):
    out = check(exe, *args.split())
    assert out.count("SUCCESS") == 2 and "FAILURE" not in out, out[-1500:]


# samples/eltwise/eltwise_unary_quantization_to_nvfp4.c -- M N ldi ldo: bf16 -> NVFP4X2 (16-row blocks, E4M3 scales, bf16-rounded arithmetic)
def test_reference_nvfp4_quantization_driver():
    out = check("eltwise_unary_quantization_to_nvfp4", "128", "33", "160", "256")
    assert out.count("SUCCESS") == 2 and "FAILURE" not in out, out[-1500:]
