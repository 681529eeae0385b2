"""A host model of the data movement of the one-problem-per-workgroup kernel for ragged 16-bit shapes (libxsmm_amd/csrc/gemm_wgp.hpp), no GPU.

The kernel brings a problem's two operand blocks into LDS as 16-byte pieces -- piece P of A = k-pair row P / ppr, rows 4 (P % ppr) .. + 3; piece P of B = column P / ppc,
k 8 (P % ppc) .. + 7 -- with the lane-linear destination of a global -> LDS request (request x fills bytes 1024 x .. of its image, lane l the 16 bytes at 16 l), and
reads MFMA operands back as `arow[(4 kg + e) * rp]` (A: four k pairs of one row) and `bcol + 16 kg` (B: eight k of one column), kg = the lane's 8-deep k group.
This model restates exactly that index arithmetic on a byte image whose untouched bytes are NaN patterns, multiplies the fragments the way v_mfma_f32_32x32x16 does
(lane & 31 = row / column, lane >> 5 = k half) and compares with the plain product: it pins the layout algebra (piece <-> address, image pitch, the zeroing of absent k
groups, which tiles a wave owns), not the silicon -- the device parity tests are tests/test_gemm_gpu.py::test_ragged_16bit_shapes_*."""
import numpy as np
import pytest


def _model(m, n, k, lda, ldb, seed=0):
    assert m % 4 == 0 and k % 8 == 0 and lda % 4 == 0 and ldb % 8 == 0
    rng = np.random.default_rng(seed)
    A = rng.integers(-4, 5, (m, k)).astype(np.float64); B = rng.integers(-4, 5, (k, n)).astype(np.float64)
    # memory images: A VNNI-2 [k/2][lda][2] halves, B [n][ldb] halves; padding = NaN (must never reach a stored result)
    a_mem = np.full((k // 2) * lda * 2, np.nan); b_mem = np.full(n * ldb, np.nan)
    for kp in range(k // 2):
        for i in range(m):
            a_mem[(kp * lda + i) * 2 + 0] = A[i, 2 * kp]; a_mem[(kp * lda + i) * 2 + 1] = A[i, 2 * kp + 1]
    for j in range(n):
        b_mem[j * ldb:j * ldb + k] = B[:, j]
    ppr, rp, ppc = m // 4, m, k // 8
    a_pieces, b_pieces = (k // 2) * ppr, n * ppc
    a_img_halves = ((a_pieces + 63) // 64) * 512; b_img_halves = ((b_pieces + 63) // 64) * 512          # 1 KiB request slots, in halves
    img_a = np.full(a_img_halves, np.nan); img_b = np.full(b_img_halves, np.nan)
    for P in range(a_pieces):                                     # request x = P // 64, lane = P % 64: destination = 1024 x + 16 lane = 16 P bytes = 8 P halves
        kp, pc = divmod(P, ppr)
        src = (kp * lda + 4 * pc) * 2                             # halves: dword (kp * lda + 4 pc)
        img_a[8 * P:8 * P + 8] = a_mem[src:src + 8]
    for P in range(b_pieces):
        col, pc = divmod(P, ppc)
        src = col * ldb + 8 * pc
        img_b[8 * P:8 * P + 8] = b_mem[src:src + 8]
    tiles_m, tiles_n = (m + 31) // 32, (n + 31) // 32
    ntiles = tiles_m * tiles_n
    kchunks, kgroups = (k + 31) // 32, k // 8
    C = np.full((m, n), np.nan)
    owners = {}
    for w in range(4):
        for t in range((ntiles + 3) // 4):
            tid = w + 4 * t
            if tid >= ntiles:
                continue
            owners[tid] = owners.get(tid, 0) + 1
            tj, ti = divmod(tid, tiles_m)
            acc = np.zeros((32, 32))
            for kc in range(kchunks):
                for s in range(2):
                    for h in range(2):
                        kg = 4 * kc + 2 * s + h
                        ok = kg < kgroups
                        kgc = kg if ok else 0
                        a_frag = np.zeros((32, 8)); b_frag = np.zeros((32, 8))
                        for li in range(32):
                            for e in range(4):                    # dword index arow[(4 kgc + e) * rp], arow = img_a dwords + 32 ti + li
                                dw = (4 * kgc + e) * rp + 32 * ti + li
                                pair = img_a[2 * dw:2 * dw + 2] if 2 * dw + 2 <= img_a.size else np.array([0.0, 0.0])      # beyond the allocation: zeros by definition
                                a_frag[li, 2 * e:2 * e + 2] = pair
                            off = (32 * tj + li) * (ppc * 8) + 8 * kgc                        # halves: bcol + 16 kgc bytes
                            b_frag[li] = img_b[off:off + 8] if off + 8 <= img_b.size else 0.0
                        if not ok:
                            a_frag[:] = 0.0; b_frag[:] = 0.0
                        with np.errstate(invalid="ignore"):
                            acc += a_frag @ b_frag.T              # rows i of the tile x columns j of the tile, over this lane half's eight k
            for li in range(32):
                for lj in range(32):
                    i, j = 32 * ti + li, 32 * tj + lj
                    if i < m and j < n:
                        C[i, j] = acc[li, lj]
    assert sorted(owners) == list(range(ntiles)) and set(owners.values()) == {1}              # every tile has exactly one wave
    return C, A @ B


@pytest.mark.parametrize("m,n,k,lda,ldb", [(40, 40, 40, 40, 40), (72, 72, 72, 72, 72), (24, 40, 8, 24, 8), (96, 96, 96, 96, 96), (44, 100, 16, 48, 24), (72, 40, 48, 76, 56), (128, 96, 32, 128, 32)])
def test_pieces_land_where_the_fragments_read_them(m, n, k, lda, ldb):
    got, ref = _model(m, n, k, lda, ldb)
    assert not np.isnan(got).any(), "padding or untouched LDS reached a stored result"
    assert np.array_equal(got, ref)


def _model_w8(m, n, k, flat, seed=1):
    """8-bit WEIGHTS (gemm_wgp16_kernel<.., AK>): the A block is a packed BYTE image -- pairs [k/2][m][2] (AK 0 / 1) or flat [k][m] (AK 2..4), lda == m -- that comes in as a
    linear copy (piece P = bytes 16 P .. of the block) and is read back as the two bytes of (row i, k pair): ushort index (4 kg + e) * m + i, resp. bytes (8 kg + 2 e) * m + i
    and + m.  Values stand for themselves here (the conversion to bf16 is a function of the byte alone)."""
    assert m % 4 == 0 and k % 8 == 0 and (m * k) % 16 == 0
    rng = np.random.default_rng(seed)
    A = rng.integers(1, 100, (m, k)).astype(np.float64)
    img = np.full(((m * k + 15) // 16) * 16, np.nan)
    mem = np.zeros(m * k)
    for kk in range(k):
        for i in range(m):
            mem[(kk * m + i) if flat else ((kk // 2) * m + i) * 2 + (kk % 2)] = A[i, kk]
    for P in range(m * k // 16):
        img[16 * P:16 * P + 16] = mem[16 * P:16 * P + 16]               # request x = P // 64, lane P % 64 -> LDS byte 16 P
    got = np.full((m, k), np.nan)
    tiles_m = (m + 31) // 32
    for ti in range(tiles_m):
        for kg in range(k // 8):
            for li in range(32):
                i = 32 * ti + li
                if i >= m:
                    continue
                for e in range(4):
                    if flat:
                        b0 = (8 * kg + 2 * e) * m + i
                        lo, hi = img[b0], img[b0 + m]
                    else:
                        u = (4 * kg + e) * m + i
                        lo, hi = img[2 * u], img[2 * u + 1]
                    got[i, 8 * kg + 2 * e], got[i, 8 * kg + 2 * e + 1] = lo, hi
    return got, A


@pytest.mark.parametrize("flat", [False, True])
@pytest.mark.parametrize("m,k", [(40, 40), (72, 72), (96, 32), (44, 16)])
def test_packed_byte_images_of_8bit_weights(m, k, flat):
    got, ref = _model_w8(m, 8, k, flat)
    assert np.array_equal(got, ref)


# ---- which wave owns which tile (wgp_deal / wgp_waves / wgp_tile_of of gemm_wgp.hpp) ---------------------------------------------------------------------------------
def _deal(tiles_m, tiles_n, tpw):
    if tiles_m == 2 and tiles_n == 2:
        return 1                # two waves, a tile row of two each (the launcher raises tpw to 2)
    if tpw < 2:
        return 0
    if tiles_m in (3, 4) and tiles_n == tpw:
        return 1                # a tile row per wave
    if tiles_n in (3, 4) and tiles_m == tpw:
        return 2                # a tile column per wave
    return 0


def _waves(tiles_m, tiles_n, deal):
    n = tiles_m if deal == 1 else tiles_n if deal == 2 else tiles_m * tiles_n
    return min(n, 4)


def _tile_of(deal, w, nw, t, tiles_m, tiles_n):
    if deal == 1:
        return (w, t) if w < tiles_m and t < tiles_n else None
    if deal == 2:
        return (t, w) if t < tiles_m and w < tiles_n else None
    tid = w + nw * t
    return (tid % tiles_m, tid // tiles_m) if tid < tiles_m * tiles_n else None


def test_every_tile_has_exactly_one_owner_and_strips_are_never_longer_than_round_robin():
    for tiles_m in range(1, 13):
        for tiles_n in range(1, 13):
            tiles = tiles_m * tiles_n
            if tiles < 2 or (tiles > 12 and not (tiles_m == 4 and tiles_n == 4)):
                continue
            tpw = (tiles + 3) // 4
            deal = _deal(tiles_m, tiles_n, tpw)
            if tiles_m == 2 and tiles_n == 2:
                tpw = 2
            nw = _waves(tiles_m, tiles_n, deal)
            if deal == 0 and tpw > 1:
                assert nw == 4          # the kernel's compile-time wave count of the round-robin deal with several tiles per wave
            owners = {}
            for w in range(nw):
                mine = [_tile_of(deal, w, nw, t, tiles_m, tiles_n) for t in range(tpw)]
                assert sum(x is not None for x in mine) <= tpw
                for x in mine:
                    if x is not None:
                        assert x not in owners, (tiles_m, tiles_n, x)
                        owners[x] = w
            assert len(owners) == tiles, (tiles_m, tiles_n, deal, nw)
            if deal:                    # a strip is as long as the round-robin deal's longest wave, and a strip wave has ALL its tpw tiles (the kernels multiply whole strips)
                assert (tiles_n if deal == 1 else tiles_m) == tpw


def test_the_shapes_the_strips_were_measured_on():
    assert _deal(3, 3, 3) == 1 and _waves(3, 3, 1) == 3           # 72^3, 80^3, 96^3: three waves, a tile row each
    assert _deal(2, 2, 1) == 1 and _waves(2, 2, 1) == 2           # 40^3 .. 64^3: two waves, two tiles each
    assert _deal(2, 3, 2) == 2 and _waves(2, 3, 2) == 3           # 64 x 96: a tile column per wave
    assert _deal(4, 3, 3) == 1 and _deal(3, 4, 3) == 2 and _deal(2, 5, 3) == 0
    assert _deal(4, 4, 4) == 1 and _waves(4, 4, 1) == 4           # 104^3 .. 120^3: four waves, a tile row of four each


# ---- the f32 form (gemm_wgp_f32_kernels.hip): wave grid and K chunks, restated on the host -------------------------------------------------------------------------
def _f32_grid(tm, tn):
    best, best_sum, pick = 1 << 30, 1 << 30, (1, 1, 1, 1)
    for wr, wc in ((2, 2), (1, 4), (4, 1), (1, 3), (3, 1), (1, 2), (2, 1), (1, 1)):
        if wr > tm or wc > tn:
            continue
        a, b = -(-tm // wr), -(-tn // wc)
        if a > 4 or b > 4:
            continue
        if a * b < best or (a * b == best and a + b < best_sum):
            best, best_sum, pick = a * b, a + b, (wr, wc, a, b)
    return pick


def _f32_img_bytes(m, n, kc):
    return (-(-(kc * (m // 4)) // 64)) * 1024 + (-(-(n * (kc // 4)) // 64)) * 1024


def test_f32_wave_grid_covers_every_tile_with_blocks_of_at_most_four_by_four():
    for m in range(4, 129, 4):
        for n in range(1, 129, 7):
            tm, tn = -(-m // 16), -(-n // 16)
            wr, wc, rm, rn = _f32_grid(tm, tn)
            assert wr * wc <= 4 and 1 <= rm <= 4 and 1 <= rn <= 4
            assert wr * rm >= tm and wc * rn >= tn, (m, n, wr, wc, rm, rn)          # the launcher refuses otherwise; for m, n <= 128 it never has to
            owners = {}
            for w in range(wr * wc):
                r0, c0 = (w // wc) * rm, (w % wc) * rn
                for a in range(rm):
                    for b in range(rn):
                        if r0 + a < tm and c0 + b < tn:
                            assert (r0 + a, c0 + b) not in owners
                            owners[(r0 + a, c0 + b)] = w
            assert len(owners) == tm * tn


def test_f32_k_chunks_fit_the_budget_and_cover_k():
    budget = 48 * 1024
    for m, n, k in ((72, 72, 72), (128, 120, 200), (96, 96, 52), (112, 112, 112), (24, 128, 300), (128, 128, 1024)):
        kc, chunks = k, 1
        if _f32_img_bytes(m, n, k) > budget:
            for nch in range(2, 17):
                kc = (-(-k // nch) + 3) & ~3
                if _f32_img_bytes(m, n, kc) <= budget:
                    break
            else:
                continue                                    # the launcher declines (k too deep for sixteen chunks): another kernel takes the shape
            chunks = -(-k // kc)
        assert kc % 4 == 0 and chunks * kc >= k and (chunks - 1) * kc < k
        assert _f32_img_bytes(m, n, kc) <= budget
