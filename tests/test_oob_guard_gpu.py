"""Out-of-bounds proof for the kernels that load "unconditionally" (round-3 advisor note, round-4 review item 7): the masked 8-bit and ragged 16-bit kernels
issue neighbour-clamped loads, the k % 64 == 32 half chunks issue full-width step-1 loads that are discarded, the LDS-DMA forms fetch whole 16-byte slots, the
FsSpMDM / packed kernels have tail columns.  torch-allocator memory has slack on every side, so an over-read there cannot fault.  Here the SAME parity tests run
in a pytest subprocess with LIBXSMM_TEST_GUARD set: tests/conftest.py then places every uploaded operand flush against UNMAPPED address space (tests/guard.py,
tests/guard_alloc.c: hipMemAddressReserve / hipMemMap), `end`: the operand's last byte is the last mapped byte, `front`: its first byte is the first mapped byte.
One load or store outside an operand is a GPU page fault and kills the subprocess -- the assertion below then names the test that was running."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

GROUPS = {
    # masked 8-bit (40^3, odd leading dimensions), half chunks k = 32 / 96 on the 1x1 and 2x2 streaming kernels, 8-bit floats of their own type
    "int8_fp8": ("tests/test_gemm_gpu.py", "int8_gemm_is_bit_identical or fp8_gemm_matches_oracle or fp8_results_of_their_own_type"),
    # ragged 16-bit (B panel by LDS-DMA = the BL forms, and B in registers), whole-tile bf16 / f16, fused epilogues
    "bf16_f16": ("tests/test_gemm_gpu.py", "ragged_16bit or bf16_gemm_matches_oracle or f16_gemm_matches_oracle or fused_epilogue"),
    # f32: headline kernel, ragged f32 (23^3, 17x9x31 with padded leading dimensions), transposes
    "f32": ("tests/test_gemm_gpu.py", "f32_gemm_matches_oracle"),
    # 8-bit weights x bf16, BF32, I16, low-bit and MX types
    "lowp": ("tests/test_gemm_gpu.py", "more_gemm_types or mxfp4_gemm_matches or mxmx_gemm_matches or low_bit_weight or interleaved_4bit"),
    # packed CSR / CSC / BCSC, FsSpMDM (tail panels), dense packed GEMMs
    "sparse": ("tests/test_sparse_gpu.py", "packed_csr_asparse or packed_bsparse or fsspmdm or bcsc"),
}


@pytest.mark.parametrize("side", ["end", "front"])
@pytest.mark.parametrize("group", sorted(GROUPS))
def test_parity_tests_with_operands_flush_against_unmapped_memory(group, side):
    path, expr = GROUPS[group]
    if side == "front" and group in ("lowp", "sparse"):
        pytest.skip("under-reads: covered for the clamped-load kernels (int8_fp8, bf16_f16, f32)")
    env = dict(os.environ, LIBXSMM_TEST_GUARD=side)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, path), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", "-k", expr, "-v", "--no-header"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, f"guarded run ({side}) of {path} -k '{expr}' ended with {r.returncode} (negative / 134: the GPU faulted on an out-of-bounds access):\n{tail}"
    assert " passed" in r.stdout
