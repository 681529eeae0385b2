"""Register / scratch budget of the compiled gfx950 kernels, read from the code objects inside libxsmm_amd.so (no GPU needed).

A kernel that needs scratch has spilled registers or keeps its argument block in memory: for the streaming kernels of this library that is a
performance bug (round 2 found two of them this way: the M-block streaming BCSC kernel before its lambdas were force-inlined, and the signed-A
int8 BCSC variant under the three-waves register cap).  The table itself is committed as profiles/r06_kernel_resources.txt."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr   # noqa: E402

LIB = os.path.join(ROOT, "libxsmm_amd", "lib", "libxsmm_amd.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(kr.LLVM, "llvm-readelf"))),
                                reason="needs the built library and the ROCm LLVM tools")

# the 512-register bf16 macro-tile kernel, f32 / element-wise bf16 C stores of 64-tiles only: a few registers of the EPILOGUE's 64-bit address
# arithmetic are spilled after the main loop and reloaded inside the epilogue (nothing inside the loop; the packed-bf16 form has no scratch)
SCRATCH_ALLOWED = {r"xamd::gemm_bf16_macro_kernel<[02], 64, false, 4, 4, 4, 0, (false|true)>": 72}

# kernel family -> least waves per SIMD that DESIGN.md's occupancy statements rely on (registers and static LDS together)
OCCUPANCY = {
    r"xamd::gemm_f32_stream_kernel_lean<.*>": 4,                 # headline: four workgroups of 32 KiB LDS per CU
    r"xamd::gemm_f32_stream_kernel<.*>": 4,
    r"xamd::gemm_f32_wg64_kernel<.*>": 4,
    r"xamd::gemm_bf16_wg64_kernel<.*>": 4,
    r"xamd::gemm_p16_kernel<1, .*>": 8,
    r"xamd::bcsc_mfma_bf16_stream_kernel<.*>": 2,                # 2048 waves = one round at two waves per SIMD
    r"xamd::bcsc_mfma_bf16_stream_full_kernel<.*>": 2,           # round 6: the same 2048 waves, one record per chunk
    r"xamd::bcsc_mfma_bf16_dma_kernel<.*>": 3,
    r"xamd::bcsc_mfma_i8_stream_full_kernel<.*>": 2,
    r"xamd::bcsc_mfma_i8_dma_kernel<., true, .*>": 3,
    r"xamd::bcsc_mfma_i8_dma_kernel<., false, .*>": 2,
    r"xamd::bcsc_mfma_f32_kernel<.*>": 4,
    r"xamd::spmm_stream_kernel<.*>": 8,
    r"xamd::gemm_f64_stream_kernel<false, false, .*>": 4,        # round 4, f64: 4096 problems of 32^3 are one round of waves
    r"xamd::gemm_f64_stream64_kernel<.*>": 2,
    r"xamd::gemm_f64_blocked_kernel<.*>": 2,                     # two workgroups per CU: one computes while the other waits at its barrier
    r"xamd::gemm_f64_p16_kernel<.*>": 8,
}


@pytest.fixture(scope="module")
def table():
    return kr.collect(LIB)


def test_every_translation_unit_is_found(table):
    names = {t["name"] for t in table}
    assert len(table) > 200
    for family in ("xamd::gemm_f32_stream_kernel_lean", "xamd::bcsc_mfma_bf16_stream_kernel", "xamd::spmm_stream_kernel", "xamd::mx_out_quant_kernel"):
        assert any(n.startswith(family) for n in names), family          # gemm_kernels.hip, sparse_kernels.hip, meltw_kernels.hip


def test_no_kernel_spills_to_scratch(table):
    bad = {}
    for t in table:
        allowed = max([v for k, v in SCRATCH_ALLOWED.items() if re.fullmatch(k, t["name"])], default=0)
        if t["scratch"] > allowed:
            bad[t["name"]] = (t["scratch"], t["spills"])
    assert not bad, f"kernels with scratch (bytes, spilled registers): {bad}"


def test_hot_kernels_keep_their_occupancy(table):
    low, seen = {}, set()
    for t in table:
        for pattern, least in OCCUPANCY.items():
            if re.fullmatch(pattern, t["name"]):
                seen.add(pattern)
                if t["waves"] < least:
                    low[t["name"]] = (t["waves"], least, t["vgpr"], t["lds"])
    assert seen == set(OCCUPANCY), f"no kernel matches {set(OCCUPANCY) - seen}"
    assert not low, f"(waves per SIMD, expected at least, registers, LDS bytes): {low}"


def test_committed_table_is_current(table):
    """profiles/r06_kernel_resources.txt is the table of THIS build (regenerate with tools/kernel_resources.py --out ...)."""
    path = os.path.join(ROOT, "profiles", "r06_kernel_resources.txt")
    committed = {}
    for line in open(path).read().splitlines()[2:]:
        cols = line.split(None, 8)
        committed[cols[8]] = (int(cols[3]), int(cols[4]))                 # static LDS and scratch; register counts may move with the compiler
    built = {t["name"]: (t["lds"], t["scratch"]) for t in table}
    assert committed == built


def test_hand_written_waits_of_the_bcsc_full_tile_kernels_stay_the_only_ones():
    """DESIGN.md section 4, decision 36: next to LDS-DMA requests in flight the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS access it can see, which lands
    the whole operand ring before every chunk.  The full-tile streaming kernels write their in-loop LDS traffic as instructions for that reason: no such wait may be
    left in them (tools/dma_wait_scan.py, from the code objects of the built library)."""
    import dma_wait_scan
    table = dma_wait_scan.scan(LIB)
    full = {k: v for k, v in table.items() if "stream_full_kernel" in k}
    assert len(full) >= 36                                   # bf16 12, f32 12, 8-bit integers 24 instances (some share a symbol prefix cut by c++filt)
    for fam in ("bcsc_mfma_bf16_stream_full_kernel", "bcsc_mfma_i8_stream_full_kernel"):
        assert any(fam in k for k in full), fam
    bad = {k: v for k, v in full.items() if v[1]}
    assert not bad, bad


def test_non_temporal_instance_of_the_elementwise_kernel_loads_non_temporally():
    """The f32 copy's non-temporal instance lost the bit on its load once (a second, cacheable load in the same function made the compiler fold the two): 0.755 -> 0.716."""
    import dma_wait_scan
    ins = dma_wait_scan.instructions(LIB, "meltw_ew8_kernelILi1ELb1ELi4E")
    assert ins, "meltw_ew8_kernel<1, true, 4> not found"
    wide = [x for x in ins if x.startswith("global_load_dwordx4") or x.startswith("global_store_dwordx4")]
    assert wide and all(x.endswith(" nt") for x in wide), wide
