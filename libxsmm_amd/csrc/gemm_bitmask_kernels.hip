// gemm_bitmask_kernels.hip -- GEMM with A compressed by bitmask, round 4: the expansion happens in REGISTERS, a wave ballot away from the matrix core.
//
// Semantics [ref: src/generator_gemm_reference_impl.c:535-556 (operand decoding), :857-948 (the loops); driver samples/xgemm/gemm_kernel.c:107-212]:
// LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK -- a.primary holds only the non-zeros of A in memory order (16-bit types: the VNNI-2 image
// [k / 2][m][2]), a.secondary one bit per element of that image (LSB first); C = beta * C + A B, f32 accumulation.
//
// Round 3's kernel (gemm_bitmask16_kernel, gemm_kernels.hip) expanded through LDS -- value windows staged, picked with a nibble table, written into a dense
// image, read back as MFMA operands: 512 bytes of LDS traffic per lane and 64-deep chunk, half of the LDS cycles bank conflicts, 46 us for 8192 x 8192 @50 % x
// 64 columns plus 30 us of helper kernels (two table passes, a 12-slice reduce): 0.12 of the roofline of the compressed bytes.  Here:
//   * bit row r (a k pair) of a wave's 32 rows is ONE 64-bit mask M (bit 2i + t = element (row i, k = 2r + t)): wave-uniform, an SGPR pair.  The row's
//     non-zeros inside the tile are a contiguous window of <= 64 values at a known offset: lane l loads value l of the window (one 2-byte buffer load, the
//     offset a scalar).  The dense position d = lane then needs value number rank(d) = popcount(M below d) = v_mbcnt(M): two instructions, and
//     ds_bpermute_b32 (the LDS crossbar, no LDS memory, no bank conflicts) fetches it; v_cndmask with M itself as the mask zeroes the clear positions.
//     FOUR vector instructions and one crossbar operation per 64 elements.
//   * that leaves lane 2i + t holding (row i, k = 2r + t); the matrix core wants lanes (i, k group).  K slots are labels: with k group = parity t, a lane
//     supplies the eight k = 2 (r0 + e) + t, e = 0..7 of eight consecutive bit rows -- eight registers packed in pairs -- and a constant-index ds_bpermute
//     per packed dword de-interleaves the lanes (2i + t -> i + 32 t).  B is re-laid once per call to match ([step of 16 k][parity][column][8 k]: a lane's
//     operand is 16 contiguous bytes, a wave's a 512-byte run) by the pre-pass kernel.
//   * no dense image, no staging, no barrier in the loop, no LDS memory at all before the epilogue.  A workgroup is SIXTEEN waves on one 32-row tile, each
//     with its own slice of k: the chip is filled without k-slices across workgroups, the sixteen partial tiles are added through LDS in wave order
//     (deterministic), C is written once.  No partial tiles in memory, no reduce kernel.
//   * one pre-pass kernel (bitmask_prepass_kernel): per bit row its total and the exclusive prefix per 32-row tile (16-bit entries: 2.5 % of the compressed
//     bytes), and the re-laid B.  Where a bit row's values start -- the scan over the row totals -- is computed by every wave for its own slice (<= 16 KiB of
//     totals, L2 resident), not by a third kernel.
//   * round 4, second session (profiles/r04b_bitmask_ablation.jsonl, timing-only switches in an EXPERIMENTS build): of 72 us the value loads cost 34, the expansion 14,
//     the meta data 11, B loads and MFMAs nothing; ONE 16-byte request per lane for all eight windows of a step cost what the eight 2-byte loads cost (at any
//     alignment): the bytes, not the instructions.  Tiles were dealt T -> XCD T % 8, so neighbouring tiles -- which share the lines of every bit row's values and of the
//     bitmap -- sat behind different L2s.  Now every XCD owns a contiguous eighth of the tiles (72 -> 59 us) and the pre-pass re-lays the masks tile by tile (a
//     block's 64 masks are 512 contiguous bytes instead of 64 lines 2 KiB apart: 59 -> 56.5 us).  Not adopted: lanes beyond a window's count masked off, B requested
//     before the values, values requested three steps ahead instead of two (no change each: the kernel is not waiting for a latency).  What is left: 34 us without any value load, 22 us for the loads (64 MB of values at 3 TB/s) -- the two do not overlap.
// m % 32 == 0, m <= 32768, k % 16 == 0 with at least 16 steps of 16, 16-bit operands (bf16 / IEEE half), C f32 / bf16 / f16.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "internal.hpp"
#include "gemm_device.hpp"

namespace xamd {

typedef unsigned long long u64b;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8b __attribute__((ext_vector_type(8)));
#define CA4 __attribute__((address_space(4)))        // wave-uniform read-only data: scalar loads

struct BitmaskArgs {
  const unsigned short* vals; const unsigned char* bitmap;     // the caller's operands
  const char* b; char* c;
  unsigned int* tot; unsigned short* tpre; unsigned short* bp;   // workspace: row totals, per-tile exclusive prefixes, the re-laid B
  unsigned long long* mt;                                         // workspace: the bit rows' masks tile by tile ([tile][bit row]: a block of 64 rows of one tile = 512 contiguous bytes)
  int m, n, k, ldb, ldc, c_type, beta0;
  int rows, row_bytes, tiles, n_pad, steps, steps_per_wave;      // bit rows (k / 2), bytes per bit row (m / 4), 32-row tiles, n rounded up to 32, k / 16
};

// ------------------------------------------------------------------------------------------------
// pre-pass: one wave per bit row (totals + tile prefixes), then the blocks that re-lay B
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void bitmask_prepass_kernel(BitmaskArgs p, unsigned int row_blocks) {
  // table layout: tpre[tile][bit row] -- the eight rows of a step are ONE 16-byte record per tile (a single scalar load in the main kernel)
  __shared__ unsigned short ex[8][64];
  __shared__ u64b mk[8][64];
  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (blockIdx.x < row_blocks) {
    const unsigned int r0 = blockIdx.x * 8u, r = r0 + wave;                         // rows is a multiple of 8: every wave has a row
    GM const u64b* row = (GM const u64b*)((GM const unsigned char*)p.bitmap + (size_t)r * (size_t)p.row_bytes);
    unsigned int carry = 0;
    for (unsigned int t0 = 0; t0 < (unsigned int)p.tiles; t0 += 64u) {
      const unsigned int t = t0 + lane;
      const u64b mword = t < (unsigned int)p.tiles ? row[t] : 0ull;
      const unsigned int c = (unsigned int)__builtin_popcountll(mword);
      mk[wave][lane] = mword;
      unsigned int incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned int v = (unsigned int)__shfl_up((int)incl, o); if (lane >= (unsigned int)o) incl += v; }
      ex[wave][lane] = (unsigned short)(carry + incl - c);
      carry += (unsigned int)__shfl((int)incl, 63);
      __syncthreads();
      if (wave == 0 && t < (unsigned int)p.tiles) {
        u32x4 rec;
#pragma unroll
        for (int q = 0; q < 4; ++q) rec[q] = (unsigned int)ex[2 * q][lane] | ((unsigned int)ex[2 * q + 1][lane] << 16);
        *(GM u32x4*)((GM unsigned short*)p.tpre + (size_t)t * (size_t)p.rows + r0) = rec;
      }
      {   // the eight rows' masks of tile t0 + tid / 8 leave as one 64-byte run: thread = (tile tid / 8, row tid % 8)
        const unsigned int tt = t0 + (threadIdx.x >> 3), q = threadIdx.x & 7u;
        if (tt < (unsigned int)p.tiles) ((GM u64b*)p.mt)[(size_t)tt * (size_t)p.rows + r0 + q] = mk[q][threadIdx.x >> 3];
      }
      __syncthreads();
    }
    if (lane == 0) ((GM unsigned int*)p.tot)[r] = carry;
    return;
  }
  // B [n][ldb] (k contiguous) -> bp [step][parity][n_pad][8]: element e of (step S, parity t, column j) is B(k = 16 S + 2 e + t, j); columns >= n are zeros
  const unsigned int id = (blockIdx.x - row_blocks) * 512u + threadIdx.x;
  const unsigned int j = id % (unsigned int)p.n_pad, S = id / (unsigned int)p.n_pad;
  if (S >= (unsigned int)p.steps) return;
  unsigned short ev[8] = {0, 0, 0, 0, 0, 0, 0, 0}, od[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (j < (unsigned int)p.n) {
    GM const unsigned short* col = (GM const unsigned short*)p.b + (size_t)j * (size_t)p.ldb + 16u * S;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ev[e] = col[2 * e]; od[e] = col[2 * e + 1]; }
  }
  u32x4 ve, vo;
#pragma unroll
  for (int q = 0; q < 4; ++q) { ve[q] = (unsigned int)ev[2 * q] | ((unsigned int)ev[2 * q + 1] << 16); vo[q] = (unsigned int)od[2 * q] | ((unsigned int)od[2 * q + 1] << 16); }
  GM u32x4* dst = (GM u32x4*)p.bp;
  dst[(size_t)(2u * S) * (unsigned int)p.n_pad + j] = ve;
  dst[(size_t)(2u * S + 1u) * (unsigned int)p.n_pad + j] = vo;
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
// v_cndmask_b32 with the bit row's mask as the selector: lane d keeps x where bit d of M is set, 0 elsewhere
__device__ __forceinline__ unsigned int keep_where_set(unsigned int x, u64b mask) {
  unsigned int r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
  return r;
}

// ABL: timing-only experiments (WRONG results; LIBXSMM_HIP_BITMASK_ABL): 1 no value loads, 2 no expansion (ballot ranks / crossbar), 4 no B loads, 8 no MFMA, 16 no meta-data blocks
template <bool F16, int NT, int KS, int ABL = 0>
__global__ __launch_bounds__(64 * KS) void gemm_bitmask_reg_kernel(BitmaskArgs p) {
  __shared__ __attribute__((aligned(16))) float red[KS / 2][NT * 1024];
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Hardware workgroup g runs on XCD g % 8 (its own L2).  Neighbouring tiles share cache lines -- a bit row's values for tiles T and T + 1 are adjacent in
  // memory (a window is ~64 of a line's 128 bytes) and a 128-byte line of the bitmap holds the masks of 16 tiles -- so every XCD takes a CONTIGUOUS eighth
  // of the tiles: what one of its waves over-fetches is what the same wave of the next tile (the next CU of the same XCD, at the same time) needs.
  // Round 4 ablation (profiles/r04b_bitmask_ablation.jsonl): with tile T on XCD T % 8 the value loads cost 34 of 72 us and ONE 16-byte request per lane for all
  // eight windows of a step cost the same as the eight 2-byte loads -- the lines, not the instructions.
  unsigned int T = blockIdx.x;
  if constexpr (!(ABL & 128)) { if ((gridDim.x & 7u) == 0u) T = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }
  const unsigned int j0 = blockIdx.y * 32u * NT;
  const unsigned int s_begin = wave * (unsigned int)p.steps_per_wave;
  const unsigned int s_end = std::min<unsigned int>((unsigned int)p.steps, s_begin + (unsigned int)p.steps_per_wave);
  f32x16 acc[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
  if (s_begin < s_end) {
    // where this wave's first bit row starts in the value array, and where the array ends: sums over the row totals (every lane 16 bytes per trip)
    unsigned int part = 0, all = 0;
    {
      const unsigned int r_first = 8u * s_begin;
      GM const u32x4* t4 = (GM const u32x4*)p.tot;                   // rows is a multiple of 8: whole 16-byte pieces
      for (unsigned int q = lane; q < (unsigned int)p.rows / 4u; q += 64u) {
        const u32x4 v = t4[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) { all += v[e]; part += (4u * q + e < r_first) ? v[e] : 0u; }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { part += (unsigned int)__shfl_xor((int)part, o); all += (unsigned int)__shfl_xor((int)all, o); }
    }
    unsigned int S = (unsigned int)__builtin_amdgcn_readfirstlane((int)part);                   // packed offset (in values) of the next bit row to be requested
    const unsigned int nnz = (unsigned int)__builtin_amdgcn_readfirstlane((int)all);
    // values: lane l = value l of the window; reads past the last non-zero of the array return 0 (buffer bounds)
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)p.vals, (short)0, (int)(2u * nnz), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bp, (short)0, -1, 0x00020000);
    const unsigned int vval = lane * 2u;
    const unsigned int vb = ((lane >> 5) * (unsigned int)p.n_pad + j0 + (lane & 31u)) * 16u;     // (parity, column) inside a step's block of the re-laid B
    const unsigned int permaddr = (2u * (lane & 31u) + (lane >> 5)) * 4u;                        // lane i + 32 t takes the packed dword of lane 2 i + t
    // Meta data (masks, tile prefixes, row totals) travel by VECTOR loads, 64 bit rows = 8 steps at a time, one row per lane, two blocks ahead of their
    // use; a step reads its eight rows' values out of the lanes with v_readlane (compile-time lane numbers).  Scalar loads were the first form: their
    // misses (every bit row's mask is another cache line, 2 KiB apart) cannot be requested far enough ahead -- a step's masks are 16 scalar registers -- and
    // while one is in flight every wait for a crossbar result is a full lgkmcnt(0): 2 us per step, 63 us for the kernel (rocprofv3, 8192 x 8192 @50 %).
    GM const u64b* mtv = (GM const u64b*)p.mt + (size_t)T * (size_t)p.rows;                          // this tile's masks, re-laid by the pre-pass: 64 rows = 512 contiguous bytes
    GM const unsigned short* tpv = (GM const unsigned short*)p.tpre + (size_t)T * (size_t)p.rows;      // this tile's column of the table
    GM const unsigned int* ttv = (GM const unsigned int*)p.tot;
    struct Block { unsigned int mlo, mhi, off, cnt; };               // per lane: bit row (first row of the block + lane): mask, packed offset of its window
    struct Raw { u32x2 m; unsigned int tot; unsigned int pre; };
    const unsigned int r_end = 8u * s_end;
    auto block_load = [&](unsigned int rb0) __attribute__((always_inline)) {
      Raw w;
      const unsigned int r = rb0 + lane < r_end ? rb0 + lane : r_end - 1u;                        // past the slice: the last row again (never used)
      w.m = *(GM const u32x2*)(mtv + r);
      w.tot = ttv[r];
      w.pre = tpv[r];
      return w;
    };
    // offsets of a block: running position S + exclusive scan of the row totals + tile prefix; S moves on by the block's total
    auto block_make = [&](const Raw& w, unsigned int rb0) __attribute__((always_inline)) {
      Block b; b.mlo = w.m[0]; b.mhi = w.m[1]; b.cnt = (unsigned int)__builtin_popcount(b.mlo) + (unsigned int)__builtin_popcount(b.mhi);
      const unsigned int t = rb0 + lane < r_end ? w.tot : 0u;
      unsigned int incl = t;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned int v = (unsigned int)__shfl_up((int)incl, o); if (lane >= (unsigned int)o) incl += v; }
      b.off = S + incl - t + w.pre;
      S += (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
      return b;
    };
    // Value windows: lane l loads value l of the window (one 2-byte buffer load per bit row; past the array: zero), the dense position d = lane fetches value
    // rank(d) with ds_bpermute_b32.  Measured alternatives (rocprofv3 kernel durations, 8192 x 8192 @50 % x 64 columns, profiles/r04_bitmask_reg_*.txt):
    //   scalar loads for masks / prefixes / totals, one step ahead:                          63 us (their misses cannot be requested further ahead: 16 SGPRs per step)
    //   this form (meta data by vector loads two blocks ahead, v_readlane):                  61 us
    //   one unaligned 16-byte request per lane for a whole step's windows, parked in LDS,
    //   gather by ds_read_u16 instead of ds_bpermute:                                        60 us
    // i.e. neither the texture path (~20 cycles per 64-lane load instruction whatever its width, tools/bperm_probe.hip) nor the crossbar (6.5 cycles per
    // ds_bpermute and CU) is what bounds the kernel: the counters show no unit above 25 % busy (vector ALU 24 %, LDS 20 %, texture addresser 60 % of the
    // cycles with ONE request in flight or more) and the waves waiting 87 % of their life -- 41 % on a counter, 46 % for an issue slot.
    unsigned int val[4][8]; u32x4 bfr[2][NT];
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
      for (int b_ = 0; b_ < 8; ++b_) val[a_][b_] = 0u;             // (lanes beyond a window's count never load: their registers keep whatever they hold)
    auto fetch = [&](auto slotc, auto lanec, const Block& b) __attribute__((always_inline)) {
      constexpr int slot = decltype(slotc)::value, l0 = decltype(lanec)::value;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (ABL & 1) val[slot][e] = vval + (unsigned int)e;
        else if constexpr (ABL & 32) {      // ONE 2-byte load per step (the other seven rows reuse it)
          if (e == 0) val[slot][0] = (unsigned int)__builtin_amdgcn_raw_buffer_load_b16(rv, (int)vval, (int)(2u * (unsigned int)__builtin_amdgcn_readlane((int)b.off, l0)), 0);
          else val[slot][e] = val[slot][0];
        } else if constexpr (ABL & 64) {    // ONE 16-byte load per step at a 2-byte aligned address: lane = (window lane / 8, piece lane % 8)
          if (e == 0) {
            const unsigned int myoff = (unsigned int)__builtin_amdgcn_ds_bpermute((int)((l0 + (lane >> 3)) << 2), (int)b.off);
            const unsigned int al = (ABL & 1024) ? (myoff & ~7u) : ((ABL & 512) ? (myoff & ~1u) : myoff);      // 16- / 4- / 2-byte aligned requests
            const u32x4 w4 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(2u * al + 16u * (lane & 7u)), 0, 0));
            val[slot][0] = w4[0]; val[slot][1] = w4[1]; val[slot][2] = w4[2]; val[slot][3] = w4[3];
          } else if (e >= 4) val[slot][e] = val[slot][e - 4];
        }
        else if constexpr (ABL & 256) {
          // only the lanes that hold a value of this window ask for one (the others fetch the next tile's values: half of the requested bytes at 50 %): no gain once
          // the neighbouring tile runs on the same XCD -- 57.3 against 56.4 us
          const unsigned int cnt = (unsigned int)__builtin_amdgcn_readlane((int)b.cnt, l0 + e);
          if (lane < cnt) val[slot][e] = (unsigned int)__builtin_amdgcn_raw_buffer_load_b16(rv, (int)vval, (int)(2u * (unsigned int)__builtin_amdgcn_readlane((int)b.off, l0 + e)), 0);
        }
        else val[slot][e] = (unsigned int)__builtin_amdgcn_raw_buffer_load_b16(rv, (int)vval, (int)(2u * (unsigned int)__builtin_amdgcn_readlane((int)b.off, l0 + e)), 0);
      }
    };
    auto fetch_b = [&](auto slotc, unsigned int s_in) __attribute__((always_inline)) {
      constexpr int slot = decltype(slotc)::value;
      const unsigned int sc = s_in < s_end ? s_in : s_end - 1u;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if constexpr (ABL & 4) bfr[slot][jt] = u32x4{sc, vb, 0x3f803f80u, 0x3f803f80u};
        else bfr[slot][jt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(vb + 512u * jt), (int)(sc * 2u * (unsigned int)p.n_pad * 16u), 0));
      }
    };
    auto expand = [&](auto slotc, auto lanec, const Block& b, u32x4& a4) __attribute__((always_inline)) {
      constexpr int slot = decltype(slotc)::value, l0 = decltype(lanec)::value;
      unsigned int dense[8];
      if constexpr (ABL & 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] = val[slot][2 * q] | (val[slot][2 * q + 1] << 16);
        return;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b.mlo, l0 + e), hi = (unsigned int)__builtin_amdgcn_readlane((int)b.mhi, l0 + e);
        const unsigned int rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
        const unsigned int x = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(rank << 2), (int)val[slot][e]);
        dense[e] = keep_where_set(x, ((u64b)hi << 32) | lo);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) a4[q] = (unsigned int)__builtin_amdgcn_ds_bpermute((int)permaddr, (int)(dense[2 * q] | (dense[2 * q + 1] << 16)));
    };
    auto mfma = [&](auto slotc, const u32x4& a4) __attribute__((always_inline)) {
      constexpr int slot = decltype(slotc)::value;
      if constexpr (ABL & 8) { acc[0][0] += __uint_as_float(a4[0] ^ a4[1] ^ a4[2] ^ a4[3] ^ bfr[slot][0][0] ^ bfr[slot][NT - 1][3]); return; }
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if constexpr (F16) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8b, bfr[slot][jt]), __builtin_bit_cast(f16x8b, a4), acc[jt], 0, 0, 0);
        else acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[slot][jt]), __builtin_bit_cast(bf16x8, a4), acc[jt], 0, 0, 0);
      }
    };
    // blocks of 8 steps: `cur` is being multiplied, `nxt` supplies the offsets of the steps requested across the block boundary, `pre` is in flight
    const unsigned int rb_first = 8u * s_begin;
    Raw w0 = block_load(rb_first), w1 = block_load(rb_first + 64u);
    Block cur = block_make(w0, rb_first);
    Block nxt = block_make(w1, rb_first + 64u);
    Raw pre = block_load(rb_first + 128u);
    // values of the block's first two steps, B operands of its first step
    fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, cur);
    fetch(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{}, cur);
    if constexpr (ABL & 4096) fetch(std::integral_constant<int, 2>{}, std::integral_constant<int, 16>{}, cur);       // (experiment: values three steps ahead)
    fetch_b(std::integral_constant<int, 0>{}, s_begin);
    for (unsigned int sb = s_begin; sb < s_end; sb += 8u) {
      static_for<8>([&](auto stc) {
        constexpr int st = stc.value;
        using SlotV = std::integral_constant<int, st & 3>; using SlotB = std::integral_constant<int, st & 1>;
        constexpr int AH = (ABL & 4096) ? 3 : 2;
        using NextV = std::integral_constant<int, (st + AH) & 3>; using NextB = std::integral_constant<int, (st + 1) & 1>;
        // requests: the values of step st + 2 (the next block's first steps from `nxt`), the B operands of step st + 1
        if constexpr (!(ABL & 2048)) fetch_b(NextB{}, sb + st + 1u);       // B first: the counter retires in order, so waiting for this step's B (issued a step ago) then leaves the younger value loads alone
        if constexpr (st + AH < 8) fetch(NextV{}, std::integral_constant<int, 8 * (st + AH)>{}, cur);
        else fetch(NextV{}, std::integral_constant<int, 8 * (st + AH - 8)>{}, nxt);
        if constexpr (ABL & 2048) fetch_b(NextB{}, sb + st + 1u);
        if (sb + st < s_end) {                               // wave-uniform
          u32x4 a4;
          expand(SlotV{}, std::integral_constant<int, 8 * st>{}, cur, a4);
          mfma(SlotB{}, a4);
        }
      });
      if constexpr (!(ABL & 16)) {
      cur = nxt;
      nxt = block_make(pre, 8u * sb + 128u);               // the block after next: its raw data was requested a block ago
      pre = block_load(8u * sb + 192u);
      }
    }
  }
  // the KS partial tiles, added in wave order through LDS (halving: waves [h, 2h) hand their tile to waves [0, h))
#pragma unroll
  for (int half = KS / 2; half >= 1; half >>= 1) {
    if (wave >= (unsigned int)half && wave < 2u * half) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - half][(jt * 16 + r) * 64 + lane] = acc[jt][r];
    }
    __syncthreads();
    if (wave < (unsigned int)half) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] += red[wave][(jt * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave != 0) return;
  const unsigned int i = 32u * T + (lane & 31u), h = lane >> 5;
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned int j = j0 + 32u * jt + (unsigned int)((r & 3) + 8 * (r >> 2)) + 4u * h;
      if (j >= (unsigned int)p.n) continue;
      const float v = acc[jt][r];
      const size_t e = (size_t)j * (size_t)p.ldc + i;
      if (p.c_type == LIBXSMM_DATATYPE_F32) { GM float* cp = (GM float*)p.c + e; *cp = p.beta0 ? v : v + *cp; }
      else if (F16) { GM _Float16* cp = (GM _Float16*)p.c + e; *cp = (_Float16)(p.beta0 ? v : v + (float)*cp); }
      else {
        GM unsigned short* cp = (GM unsigned short*)p.c + e;
        float y = p.beta0 ? v : v + __uint_as_float((unsigned int)*cp << 16);
        unsigned int u = __float_as_uint(y);                                  // RNE with denormals-are-zero and NaN quieting [ref: src/libxsmm_math.c:684-704]
        if ((u & 0x7f800000u) == 0u) u &= 0x80000000u;
        if ((u & 0x7f800000u) == 0x7f800000u) { if (u & 0x007fffffu) u |= 0x00400000u; } else u += 0x00007fffu + ((u >> 16) & 1u);
        *cp = (unsigned short)(u >> 16);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool bitmask_reg_ok(const GemmArgs& a) {
  constexpr bool off = false;
  const bool t16 = (a.a_type == LIBXSMM_DATATYPE_BF16 || a.a_type == LIBXSMM_DATATYPE_F16) && a.b_type == a.a_type;
  if (off || !t16 || a.m <= 0 || a.n <= 0 || a.k <= 0 || (a.m % 32) || a.m > 32768 || (a.k % 16) || a.k / 16 < 16) return false;
  if (a.c_type != LIBXSMM_DATATYPE_F32 && a.c_type != a.a_type) return false;
  if ((((size_t)a.a) & 1) || (((size_t)a.b) & 1)) return false;
  // every 64-column block of C expands A again: beyond a few blocks the dense image, built once, is the cheaper form (the caller's other path)
  return a.n <= 256 && (long long)a.m * a.k < (1ll << 31);
}
// bytes of workspace the register-expanding path needs for this problem; 0: not taken
size_t gemm_bitmask_reg_workspace(const GemmArgs& a) {
  if (!bitmask_reg_ok(a)) return 0;
  const size_t rows = (size_t)a.k / 2, tiles = (size_t)a.m / 32, n_pad = a.n <= 32 ? 32 : ((size_t)a.n + 63) / 64 * 64;
  return ((rows * 4 + 255) & ~(size_t)255) + ((rows * tiles * 2 + 255) & ~(size_t)255) + ((rows * tiles * 8 + 255) & ~(size_t)255) + (size_t)a.k * n_pad * 2;
}
int launch_gemm_bitmask_reg(const GemmArgs& a, const void* bitmap, void* ws, size_t ws_bytes, void* stream, const char** name, int* taken) {
  *taken = 0;
  const size_t need = gemm_bitmask_reg_workspace(a);
  if (need == 0 || ws == nullptr || ws_bytes < need || (((size_t)bitmap) & 7)) return 0;
  hipStream_t st = (hipStream_t)stream;
  BitmaskArgs p{};
  p.vals = (const unsigned short*)a.a; p.bitmap = (const unsigned char*)bitmap; p.b = a.b; p.c = a.c;
  p.m = a.m; p.n = a.n; p.k = a.k; p.ldb = a.ldb; p.ldc = a.ldc; p.c_type = a.c_type; p.beta0 = (a.flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 1 : 0;
  p.rows = a.k / 2; p.row_bytes = a.m / 4; p.tiles = a.m / 32; p.n_pad = a.n <= 32 ? 32 : (a.n + 63) / 64 * 64; p.steps = a.k / 16;        // (whole column blocks: no operand read past the re-laid B)
  constexpr int KS = 16;
  p.steps_per_wave = (p.steps + KS - 1) / KS;
  char* w = (char*)ws;
  p.tot = (unsigned int*)w; w += ((size_t)p.rows * 4 + 255) & ~(size_t)255;
  p.tpre = (unsigned short*)w; w += ((size_t)p.rows * p.tiles * 2 + 255) & ~(size_t)255;
  p.mt = (unsigned long long*)w; w += ((size_t)p.rows * p.tiles * 8 + 255) & ~(size_t)255;
  p.bp = (unsigned short*)w;
  const unsigned int row_blocks = (unsigned int)p.rows / 8u, b_blocks = ((unsigned int)p.steps * (unsigned int)p.n_pad + 511u) / 512u;
  hipLaunchKernelGGL(bitmask_prepass_kernel, dim3(row_blocks + b_blocks), dim3(512), 0, st, p, row_blocks);
  const bool f16 = a.a_type == LIBXSMM_DATATYPE_F16;
  const bool two = p.n_pad >= 64;                                   // 64 columns per workgroup where C has them
  const dim3 grid((unsigned int)p.tiles, (unsigned int)(p.n_pad / (two ? 64 : 32)));
#ifdef LIBXSMM_HIP_EXPERIMENTS
  constexpr int abl = 0;
  if (two && !f16 && abl) {
    switch (abl) {
      case 1: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 1>), grid, dim3(64 * KS), 0, st, p); break;
      case 2: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 2>), grid, dim3(64 * KS), 0, st, p); break;
      case 3: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 3>), grid, dim3(64 * KS), 0, st, p); break;
      case 4: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 4>), grid, dim3(64 * KS), 0, st, p); break;
      case 8: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 8>), grid, dim3(64 * KS), 0, st, p); break;
      case 16: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 16>), grid, dim3(64 * KS), 0, st, p); break;
      case 32: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 32>), grid, dim3(64 * KS), 0, st, p); break;
      case 128: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 128>), grid, dim3(64 * KS), 0, st, p); break;
      case 256: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 256>), grid, dim3(64 * KS), 0, st, p); break;
      case 129: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 1>), grid, dim3(64 * KS), 0, st, p); break;
      case 144: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 16>), grid, dim3(64 * KS), 0, st, p); break;
      case 64: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 64>), grid, dim3(64 * KS), 0, st, p); break;
      case 4096: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 4096>), grid, dim3(64 * KS), 0, st, p); break;
      case 2048: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 2048>), grid, dim3(64 * KS), 0, st, p); break;
      case 576: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 576>), grid, dim3(64 * KS), 0, st, p); break;
      case 1088: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 1088>), grid, dim3(64 * KS), 0, st, p); break;
      case 48: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 48>), grid, dim3(64 * KS), 0, st, p); break;
      case 7: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 7>), grid, dim3(64 * KS), 0, st, p); break;
      case 23: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 23>), grid, dim3(64 * KS), 0, st, p); break;
      default: hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS, 31>), grid, dim3(64 * KS), 0, st, p); break;
    }
    if (name) *name = "gemm_bitmask_reg_kernel(ablation)";
    *taken = 1;
    return (int)hipGetLastError();
  }
#endif
  if (two) { if (f16) hipLaunchKernelGGL((gemm_bitmask_reg_kernel<true, 2, KS>), grid, dim3(64 * KS), 0, st, p); else hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 2, KS>), grid, dim3(64 * KS), 0, st, p); }
  else { if (f16) hipLaunchKernelGGL((gemm_bitmask_reg_kernel<true, 1, KS>), grid, dim3(64 * KS), 0, st, p); else hipLaunchKernelGGL((gemm_bitmask_reg_kernel<false, 1, KS>), grid, dim3(64 * KS), 0, st, p); }
  if (name) *name = "gemm_bitmask_reg_kernel";
  *taken = 1;
  return (int)hipGetLastError();
}

}  // namespace xamd
