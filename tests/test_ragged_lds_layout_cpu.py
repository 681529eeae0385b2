"""A host model of the data movement of the ragged 16-bit matrix-core kernel's LDS forms (gemm_mfma_bf16_kernel<.., BL> and the prepared <.., BL, BND>), no GPU.

The kernel moves operand panels into a wave-private LDS image with one DWORD per lane and instruction (lane-linear destination, the permutation on the SOURCE side) and
reads MFMA operands back with the swizzle undone.  This model restates exactly that index arithmetic -- which global dword lands in which LDS slot, which slot a lane reads
for operand register e of step s -- and multiplies what the lanes hold the way v_mfma_f32_32x32x16 does (lane = row / column, lane half = k half).  Everything a request
must NOT deliver (beyond the block's extent: dropped by the bounded descriptor; left-over LDS contents) is NaN, so a leak into a stored result shows.  It pins the layout
algebra (swizzle, the 32-row A image, tail selects), not the silicon: the device parity tests are tests/test_gemm_gpu.py::test_ragged_16bit_shapes_*."""
import numpy as np
import pytest


def _wave(A2, B, m, n, k, lda, ldb, i0, j0, MT, NT, bounded):
    """One wave's 32 MT x 32 NT tile.  A2: VNNI-2 image as float64 'dwords' [kp * lda + i] -> (a_even, a_odd) pairs, B: [j * ldb + kk].  Returns acc[MT][NT][32 rows][32 cols]."""
    kpl = k // 2 - 1
    kchunks = (k + 31) // 32
    a_extent = kpl * lda + m                 # dwords (bounded descriptor of A)
    b_extent = ((n - 1) * ldb + k) // 2      # dwords (k, ldb even)
    acc = np.zeros((MT, NT, 32, 32))
    lanes = np.arange(64)
    li, h = lanes & 31, lanes >> 5
    d, fb = lanes & 15, lanes >> 4
    for kc in range(kchunks):
        img_b = np.full((NT * 8 * 64, 2), np.nan)                  # LDS dword slots of the B image, each dword = two halves
        for x in range(NT * 8):
            f = fb + 4 * x
            pc = (d >> 2) ^ ((f >> 1) & 3)
            kp = kc * 16 + pc * 4 + (d & 3)
            jc = j0 + f
            for l in lanes:
                if bounded:
                    dw = (jc[l] * ldb) // 2 + kp[l]                 # no clamp: the descriptor drops what lies beyond the block
                    if dw < b_extent and jc[l] * ldb % 2 == 0:
                        img_b[64 * x + l] = B[2 * dw:2 * dw + 2]
                else:
                    dw = (min(jc[l], n - 1) * ldb) // 2 + min(kp[l], kpl)
                    img_b[64 * x + l] = B[2 * dw:2 * dw + 2]
        if bounded:
            img_a = np.full((MT * 8 * 64, 2), np.nan)
            for x in range(MT * 8):
                for l in lanes:
                    if MT == 2:
                        kp_, row = 16 * kc + x, i0 + l
                    else:
                        kp_, row = 16 * kc + 2 * x + h[l], i0 + li[l]
                    dw = kp_ * lda + row
                    if dw < a_extent:
                        img_a[64 * x + l] = A2[dw]
        for s in range(2):
            af = np.zeros((MT, 64, 4, 2)); bf = np.zeros((NT, 64, 4, 2))
            for l in lanes:
                for e in range(4):
                    kp = kc * 16 + 8 * h[l] + 4 * s + e
                    kok = kp <= kpl
                    for nt in range(NT):
                        f = 32 * nt + li[l]
                        slot16 = f * 4 + ((2 * h[l] + s) ^ ((f >> 1) & 3))          # 16-byte slot index; dword e of it
                        v = img_b[slot16 * 4 + e]
                        bf[nt, l, e] = v if kok else 0.0
                    for mt in range(MT):
                        if bounded:
                            v = img_a[(8 * h[l] + 4 * s + e) * (32 * MT) + 32 * mt + li[l]]
                        else:
                            v = A2[min(kp, kpl) * lda + min(i0 + 32 * mt + li[l], m - 1)]
                        af[mt, l, e] = v if kok else 0.0
            # v_mfma_f32_32x32x16: D[row][col] += sum over the two lane halves and a lane's 8 values of A(row, .) * B(., col); A operand lane = row, B operand lane = column
            for mt in range(MT):
                for nt in range(NT):
                    a = af[mt].reshape(2, 32, 8); b = bf[nt].reshape(2, 32, 8)
                    acc[mt, nt] += np.einsum("hik,hjk->ij", a, b)
    return acc


CASES = [(40, 40, 40, 40, 40, 2, 2), (24, 24, 24, 24, 24, 1, 1), (40, 33, 200, 44, 202, 2, 2), (7, 5, 2, 7, 2, 1, 1), (65, 31, 34, 66, 34, 2, 2), (72, 72, 72, 72, 72, 2, 2)]


@pytest.mark.parametrize("bounded", [False, True], ids=["clamped", "bounded"])
@pytest.mark.parametrize("m,n,k,lda,ldb,MT,NT", CASES)
def test_lds_image_and_operand_reads_reproduce_the_product(m, n, k, lda, ldb, MT, NT, bounded):
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    Ad = rng.integers(-4, 5, size=(m, k)).astype(np.float64)
    Bd = rng.integers(-4, 5, size=(k, n)).astype(np.float64)
    A2 = np.full(((k // 2) * lda, 2), 7.0)                         # padding rows hold real (finite) numbers, as in a caller's buffer
    for kp in range(k // 2):
        A2[kp * lda:kp * lda + m, 0] = Ad[:, 2 * kp]; A2[kp * lda:kp * lda + m, 1] = Ad[:, 2 * kp + 1]
    B = np.full(n * ldb + 64, 5.0)
    for j in range(n):
        B[j * ldb:j * ldb + k] = Bd[:, j]
    ref = Ad @ Bd
    tm, tn = 32 * MT, 32 * NT
    for i0 in range(0, m, tm):
        for j0 in range(0, n, tn):
            acc = _wave(A2, B, m, n, k, lda, ldb, i0, j0, MT, NT, bounded)
            for mt in range(MT):
                for nt in range(NT):
                    r0, c0 = i0 + 32 * mt, j0 + 32 * nt
                    rows, cols = max(0, min(32, m - r0)), max(0, min(32, n - c0))
                    if rows and cols:
                        assert np.array_equal(acc[mt, nt, :rows, :cols], ref[r0:r0 + rows, c0:c0 + cols]), (i0, j0, mt, nt)
