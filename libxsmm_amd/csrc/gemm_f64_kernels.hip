// gemm_f64_kernels.hip -- double-precision GEMM / BRGEMM on v_mfma_f64_16x16x4_f64 (gfx950).
//
// Semantics [ref: src/generator_gemm_reference_impl.c:1322-1358 (the f64 loop), :180-197 (batch-reduce addressing)]:
//   C[m x n] = beta * C + sum_{r < br} op(A_r) * op(B_r),  beta in {0, 1}, all four transpose combinations, column-major.
// The reference accepts no fused operator on f64 (gemm_supported), so there is no epilogue besides beta.
//
// The instruction: D[row][col] += sum_{s < 4} X[row][s] * Y[s][col] on a 16 x 16 tile; lane l hands over X[row = l & 15][s = l >> 4] and
// Y[s = l >> 4][col = l & 15] (one double each) and holds D[row = (l >> 4) + 4 r][col = l & 15] in register pair r = 0..3 -- NOT the f32 map.
// As everywhere in this library the product is formed TRANSPOSED: X is the B side (row = j), Y the A side (col = i), so a lane's results
// lie along i, the contiguous dimension of C.  Both "row", "col" and the slot s are just LABELS: any bijection label <-> matrix index works as
// long as the two operands agree on k.  The kernels below choose the bijections so that every global access is a 16-byte access of a
// 256-byte contiguous run:
//   * an operand whose OUTER index is contiguous in memory (A, or B under TRANS_B) is read straight into registers, a lane taking the two
//     neighbouring rows (2g, 2g + 1) of one k: label g <-> rows 2g + t, the two halves of the register pair feed two different MFMA tiles
//     (the tile of even and the tile of odd rows).  C then leaves as 16-byte pieces as well: a lane holds C(2g, j) and C(2g + 1, j);
//   * an operand whose K index is contiguous (B, or A under TRANS_A) is fetched lane-linear (16 lanes x 16 bytes = one 256-byte column of a
//     32-deep chunk), parked in a wave-private 8 KiB LDS image whose 16-byte slots are XOR-swizzled by the column, and read back as
//     ds_read_b128: a lane gets the k pair (2c, 2c + 1) of its column, the halves feed two consecutive MFMA steps.  No barrier anywhere:
//     an image is written and read by the same wave.
// A step's four slots therefore carry k = 8u + 2s + h (u = step pair, h = half), not four consecutive k: sums differ from the reference's
// serial loop by rounding order only (test bound 1e-12 in normf, the reference driver's own f64 bound is 1.2e-5).
//
// Kernels
//   gemm_f64_stream_kernel<TA, TB>   whole 32 x 32 x (32 c) tiles, 16-byte aligned strided operands: one wave per C tile, any batch form
//                                    that is strided (1-D, 2-D, shared operands), br = none / stride; chunks software-pipelined.
//   gemm_f64_ragged_kernel<TA, TB>   everything else (13 x 5 x 7, 23^3, pointer lists, offset lists, odd leading dimensions): one wave per
//                                    32 x 32 tile on 2 x 2 MFMA tiles, operands read element-wise in the natural label order, masked.
//   gemm_f64_p16_kernel              16 x 16 x 16 problems, one per wave, both operands through wave-private LDS images.
//   gemm_f64_blocked_kernel          2-D batches of 32^3-tiles = blocked GEMMs: a workgroup owns 128 x 128 of C, operands shared through LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "internal.hpp"
#include "gemm_device.hpp"

namespace xamd {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f64x4 mfma_f64(double x, double y, const f64x4& acc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------
// streaming kernel
// ------------------------------------------------------------------------------------------------
// One 32-deep chunk of an operand tile is 32 outer x 32 k doubles = 8 KiB = eight 16-byte requests per lane.
//   direct (outer index contiguous, element (o, k) at base[o + k * ld]): request e = 2u + h is k = 8u + 2s + h, rows 2g, 2g + 1 -> registers
//   staged (k contiguous, element (o, k) at base[k + o * ld]): request x is column o = 4x + s, LDS-DMA (no registers): the destination is lane
//     linear (row o = 256 bytes, lane g writes slot g), so the swizzle sits on the SOURCE: slot g receives the k pair g ^ (o & 15).
// Fragments of a staged chunk: frag[t][u] = (k = 8u + 2s, 8u + 2s + 1) of column 16t + g = slot (4u + s) ^ g of row 16t + g: the 16 lanes that
// ds_read_b128 serves together ({0-3, 12-15, 20-27} ...) hit 16 different slots -- conflict free, like the 8-lane groups of the write side.
template <int AUX>
__device__ __forceinline__ void f64_direct_load(f64x2& v, __amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff) {
  v = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
__device__ __forceinline__ void f64_read_frags(f64x2 (&frag)[2][4], const f64x2* img, unsigned int g, unsigned int s) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) frag[t][u] = img[(16u * t + g) * 16u + (((4u * u + s) ^ g) & 15u)];
}

// AUX: cache policy of the operand requests (0 default, 2 = nt: a launch whose operands cannot be cache resident, see launch_gemm_f64)
template <bool TA, bool TB, int AUX>
__global__ __launch_bounds__(256) void gemm_f64_stream_kernel(GemmArgs p) {
  constexpr bool SA = TA, SB = !TB;                       // which operand has k contiguous and goes through LDS
  constexpr int NST = (SA ? 1 : 0) + (SB ? 1 : 0);
  __shared__ __attribute__((aligned(16))) f64x2 lds_all[4][NST ? NST * 512 : 1];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = logical_block(p) * 4u + wave;
  const unsigned int per_gemm = (unsigned int)(p.tiles_m * p.tiles_n);
  if (wid >= per_gemm * p.nbatch) return;
  unsigned int bidx = wid, i0 = 0, j0 = 0;
  if (per_gemm != 1) {
    bidx = wid / per_gemm;
    const unsigned int t = wid - bidx * per_gemm, tn = t / (unsigned int)p.tiles_m;
    i0 = (t - tn * (unsigned int)p.tiles_m) * 32u; j0 = tn * 32u;
  }
  const unsigned int lane = threadIdx.x & 63u, g = lane & 15u, s = lane >> 4;
  f64x2* img_a = lds_all[wave];
  f64x2* img_b = lds_all[wave] + (SA ? 512 : 0);
  const BatchPtrs q = batch_ptrs(p, bidx);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb, ldc = (unsigned int)p.ldc;
  // lane offsets (bytes) of the eight requests of a chunk: loop invariant, ONE register for a direct operand, four for a staged one (the
  // swizzle depends on the column modulo 16); which request and which chunk it is are scalar offsets
  const unsigned int vdirA = (2u * s * lda + 2u * g) * 8u, vdirB = (2u * s * ldb + 2u * g) * 8u;
  unsigned int vstA[4], vstB[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    vstA[e] = (s * lda + 2u * (g ^ ((4u * e + s) & 15u))) * 8u;
    vstB[e] = (s * ldb + 2u * (g ^ ((4u * e + s) & 15u))) * 8u;
  }
  const unsigned long long orgA = SA ? 8ull * i0 * lda : 8ull * i0, orgB = SB ? 8ull * j0 * ldb : 8ull * j0;
  const unsigned int kstepA = SA ? 256u : 256u * lda, kstepB = SB ? 256u : 256u * ldb;     // bytes per 32-deep chunk
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  // C addressing: i = i0 + (SA ? 16 tp + g : 2g + tp), j = j0 + (SB ? 16 tq + s + 4r : 2 (s + 4r) + tq)
  gptr ctile = q.c + 8ull * ((unsigned long long)j0 * ldc + i0);
  const bool c16 = ((((unsigned long long)(size_t)ctile) | (8ull * ldc)) & 15ull) == 0ull;      // wave-uniform
  auto c_off = [&](int tp, int tq, int r) -> unsigned long long {
    const unsigned int i = SA ? 16u * tp + g : 2u * g + tp;
    const unsigned int j = SB ? 16u * tq + s + 4u * r : 2u * (s + 4u * r) + tq;
    return 8ull * ((unsigned long long)j * ldc + i);
  };
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  const unsigned long long total = p.br_count * kchunks;
  gcptr ar = nullptr, br = nullptr;
  if (p.br_count != 0) br_base(p, q, 0, ar, br);
  __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + orgA), rb = wave_rsrc(br + orgB);
  f64x2 da[8], db[8];                           // a direct operand's chunk (unused for a staged operand)
  auto request_a = [&](int e, unsigned int kc) {
    if constexpr (SA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)((char*)img_a + 1024 * e), 16, (int)vstA[e & 3], (int)(kc * kstepA + 32u * e * lda), 0, AUX);
    else f64_direct_load<AUX>(da[e], ra, vdirA, kc * kstepA + (8u * (e >> 1) + (e & 1)) * 8u * lda);
  };
  auto request_b = [&](int e, unsigned int kc) {
    if constexpr (SB) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)((char*)img_b + 1024 * e), 16, (int)vstB[e & 3], (int)(kc * kstepB + 32u * e * ldb), 0, AUX);
    else f64_direct_load<AUX>(db[e], rb, vdirB, kc * kstepB + (8u * (e >> 1) + (e & 1)) * 8u * ldb);
  };
  if (total != 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) request_b(e, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) request_a(e, 0);
  }
  f64x4 acc[2][2];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) acc[tp][tq] = f64x4{0.0, 0.0, 0.0, 0.0};
  if (!beta0) {
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!SA && c16) {
          const f64x2 v = *(GM const f64x2*)(ctile + c_off(0, tq, r));
          acc[0][tq][r] = v.x; acc[1][tq][r] = v.y;
        } else {
          acc[0][tq][r] = *(GM const double*)(ctile + c_off(0, tq, r));
          acc[1][tq][r] = *(GM const double*)(ctile + c_off(1, tq, r));
        }
      }
  }
  // Chunk sequence over (batch-reduce element, 32-deep chunk).  Per chunk: everything requested has landed (vmcnt 0); the staged operand's
  // fragments are read out of the image; then the NEXT chunk is requested -- the image is free again, and a direct operand's registers are
  // re-requested pair by pair right behind the MFMAs that consumed them -- so the next chunk travels under this chunk's 32 MFMAs (2048 cycles).
  unsigned long long r = 0; unsigned int kc = 0;
  for (unsigned long long t = 0; t < total; ++t) {
    f64x2 fa[2][4], fb[2][4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (SA) f64_read_frags(fa, img_a, g, s);
    if constexpr (SB) f64_read_frags(fb, img_b, g, s);
    if constexpr (NST != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool more = t + 1 < total;                                        // wave-uniform
    if (++kc == kchunks) {
      kc = 0;
      if (++r < p.br_count) { br_base(p, q, r, ar, br); ra = wave_rsrc(ar + orgA); rb = wave_rsrc(br + orgB); }
    }
    if (more) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { if constexpr (SA) request_a(e, kc); if constexpr (SB) request_b(e, kc); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
          for (int tq = 0; tq < 2; ++tq) {
            const double x = SB ? fb[tq][u][h] : db[2 * u + h][tq];       // the B side: row label <-> j
            const double y = SA ? fa[tp][u][h] : da[2 * u + h][tp];       // the A side: column label <-> i
            acc[tp][tq] = mfma_f64(x, y, acc[tp][tq]);
          }
      if (more) {
        if constexpr (!SA) { request_a(2 * u, kc); request_a(2 * u + 1, kc); }
        if constexpr (!SB) { request_b(2 * u, kc); request_b(2 * u + 1, kc); }
      }
    }
  }
#pragma unroll
  for (int tq = 0; tq < 2; ++tq)
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) {
      if (!SA && c16) st_stream((GM f64x2*)(ctile + c_off(0, tq, r2)), f64x2{acc[0][tq][r2], acc[1][tq][r2]});
      else {
        st_stream((GM double*)(ctile + c_off(0, tq, r2)), acc[0][tq][r2]);
        st_stream((GM double*)(ctile + c_off(1, tq, r2)), acc[1][tq][r2]);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// streaming kernel for 64 x 64 tiles: one WAVE per tile, nothing fetched twice
// ------------------------------------------------------------------------------------------------
// With 32 x 32 tiles a 64^3 problem is four waves that each fetch half of A and half of B: every operand byte travels L2 -> CU twice.  Here one
// wave owns the whole 64 x 64 tile -- 4 x 4 MFMA tiles, 128 accumulator registers, the wave of the blocked kernel with private operands -- and
// walks K in stages of 16: A (two 32-row blocks) as eight 16-byte row-pair requests into registers, re-requested for the next stage right
// behind the MFMAs that consumed them; B (64 columns x 16 k) by LDS-DMA into a wave-private 8 KiB image [64 columns][8 slots], slot XOR-swizzled
// by the column, read back as eight ds_read_b128; the next stage's DMA goes out as soon as the fragments are in registers.  64 MFMAs per stage.
// NN, strided operands, 16-byte aligned; two waves per SIMD.
template <int AUX>
__global__ __launch_bounds__(256, 2) void gemm_f64_stream64_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) f64x2 lds_all[4][512];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int nwaves = gridDim.x * 4u;
  const unsigned int per_gemm = (unsigned int)(p.tiles_m * p.tiles_n), ntiles = per_gemm * p.nbatch;
  unsigned int tile = logical_block(p) * 4u + wave;
  if (tile >= ntiles) return;
  const unsigned int lane = threadIdx.x & 63u, g = lane & 15u, s = lane >> 4;
  f64x2* img = lds_all[wave];
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb, ldc = (unsigned int)p.ldc;
  const unsigned int vA = (2u * s * lda + 2u * g) * 8u;                                  // request (ib, e = 2u + h): k = 8u + 2s + h, rows 32 ib + 2g, + 1
  const unsigned int ob = lane >> 3, sl = lane & 7u;
  const unsigned int vB = (ob * ldb + 2u * (sl ^ (ob & 7u))) * 8u;                       // request x: column 8x + ob, slot sl <- k pair sl ^ (column & 7)
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const unsigned int kstages = (unsigned int)p.k >> 4;
  if (p.br_count * kstages == 0) return;                                                 // (the host does not launch empty chains)
  // A wave walks the tiles tile, tile + nwaves, ... as ONE sequence of stages: the first stage of the next tile is requested under the last
  // stage's MFMAs of the current one, so only the very first request of a wave is exposed (0.69 / 0.71 -> 0.71 / 0.72 at batch 4096 / 32768).
  // Requests TWO stages ahead (a second B image, a second set of A registers, in-order vmcnt counts) measured no better: 0.72 / 0.69.
  BatchPtrs q; unsigned int i0, j0;
  auto locate = [&](unsigned int t) {
    unsigned int bidx = t; i0 = 0; j0 = 0;
    if (per_gemm != 1) {
      bidx = t / per_gemm;
      const unsigned int tt = t - bidx * per_gemm, tn = tt / (unsigned int)p.tiles_m;
      i0 = (tt - tn * (unsigned int)p.tiles_m) * 64u; j0 = tn * 64u;
    }
    q = batch_ptrs(p, bidx);
  };
  locate(tile);
  gcptr ar, br;
  br_base(p, q, 0, ar, br);
  __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + 8ull * i0), rb = wave_rsrc(br + 8ull * j0 * ldb);
  f64x2 da[2][4];
  auto request_a = [&](int ib, int e, unsigned int kc) {
    da[ib][e] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)vA, (int)((kc * 16u + 8u * (e >> 1) + (e & 1)) * 8u * lda + 256u * ib), AUX));
  };
  auto request_b = [&](unsigned int kc) {
#pragma unroll
    for (int x = 0; x < 8; ++x) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)((char*)img + 1024 * x), 16, (int)vB, (int)(kc * 128u + 64u * x * ldb), 0, AUX);
  };
  request_b(0);
#pragma unroll
  for (int e = 0; e < 4; ++e) { request_a(0, e, 0); request_a(1, e, 0); }
  f64x4 acc[2][2][4];                          // [ib][par][jt]: rows 32 ib + 2g + par, columns 16 jt + s + 4r
  gptr ctile = q.c + 8ull * ((unsigned long long)j0 * ldc + i0 + 2u * g + (unsigned long long)s * ldc);
  bool c16 = ((((unsigned long long)(size_t)q.c) | (8ull * ldc)) & 15ull) == 0ull;       // wave-uniform (i0, 2g are even)
  auto init_acc = [&]() {
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        acc[ib][0][jt] = f64x4{0.0, 0.0, 0.0, 0.0}; acc[ib][1][jt] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (!beta0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gptr at = ctile + 8ull * (32ull * ib + (16ull * jt + 4ull * r) * ldc);
            if (c16) { const f64x2 v = *(GM const f64x2*)at; acc[ib][0][jt][r] = v.x; acc[ib][1][jt][r] = v.y; }
            else { acc[ib][0][jt][r] = *(GM const double*)at; acc[ib][1][jt][r] = *(GM const double*)(at + 8); }
          }
        }
      }
  };
  init_acc();
  unsigned long long r = 0; unsigned int kc = 0;
  for (;;) {
    f64x2 fb[4][2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int u = 0; u < 2; ++u) fb[jt][u] = img[(16u * jt + g) * 8u + ((4u * u + s) ^ (g & 7u))];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the stage after this one: next k stage, next batch-reduce element, or the first stage of the wave's next tile
    bool last_of_tile = false, more = true;
    gptr ctile_now = ctile; const bool c16_now = c16;
    if (++kc == kstages) {
      kc = 0;
      if (++r == p.br_count) {
        r = 0; last_of_tile = true;
        tile += nwaves;
        more = tile < ntiles;
        if (more) {
          locate(tile);
          ctile = q.c + 8ull * ((unsigned long long)j0 * ldc + i0 + 2u * g + (unsigned long long)s * ldc);
          c16 = ((((unsigned long long)(size_t)q.c) | (8ull * ldc)) & 15ull) == 0ull;
        }
      }
      if (more) { br_base(p, q, r, ar, br); ra = wave_rsrc(ar + 8ull * i0); rb = wave_rsrc(br + 8ull * j0 * ldb); }
    }
    if (more) request_b(kc);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
              acc[ib][par][jt] = mfma_f64(fb[jt][u][h], da[ib][2 * u + h][par], acc[ib][par][jt]);
        if (more) { request_a(0, 2 * u + h, kc); request_a(1, 2 * u + h, kc); }
      }
    if (last_of_tile) {
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r2 = 0; r2 < 4; ++r2) {
            gptr at = ctile_now + 8ull * (32ull * ib + (16ull * jt + 4ull * r2) * ldc);
            if (c16_now) st_stream((GM f64x2*)at, f64x2{acc[ib][0][jt][r2], acc[ib][1][jt][r2]});
            else { st_stream((GM double*)at, acc[ib][0][jt][r2]); st_stream((GM double*)(at + 8), acc[ib][1][jt][r2]); }
          }
      if (!more) break;
      init_acc();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// general kernel: any shape, any leading dimensions, any batch / batch-reduce form
// ------------------------------------------------------------------------------------------------
// One wave per 32 x 32 tile of C (2 x 2 MFMA tiles; tiles that lie outside the matrix are skipped wave-uniformly).  Natural labels: the
// A side's lane (g, s) is A(i0 + 16 tp + g, k0 + s), the B side's B(k0 + s, j0 + 16 tq + g).  A lane reads its elements one by one; rows
// i >= m and columns j >= n are never loaded (their results are never stored), k >= K contributes exact zeros on BOTH sides (a zero times
// whatever lies behind the operand could be a NaN).  Sixteen k per trip: all 16 loads of a trip are in flight before its MFMAs.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f64_ragged_kernel(GemmArgs p) {
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = logical_block(p) * 4u + wave;
  const unsigned int per_gemm = (unsigned int)(p.tiles_m * p.tiles_n);
  if (wid >= per_gemm * p.nbatch) return;
  unsigned int bidx = wid, i0 = 0, j0 = 0;
  if (per_gemm != 1) {
    bidx = wid / per_gemm;
    const unsigned int t = wid - bidx * per_gemm, tn = t / (unsigned int)p.tiles_m;
    i0 = (t - tn * (unsigned int)p.tiles_m) * 32u; j0 = tn * 32u;
  }
  const unsigned int lane = threadIdx.x & 63u, g = lane & 15u, s = lane >> 4;
  const BatchPtrs q = batch_ptrs(p, bidx);
  const long long lda = p.lda, ldb = p.ldb, ldc = p.ldc;
  const int m = p.m, n = p.n, K = p.k;
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const bool two_i = (int)i0 + 16 < m, two_j = (int)j0 + 16 < n;          // wave-uniform: is the second tile row / column inside the matrix
  f64x4 acc[2][2];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) acc[tp][tq] = f64x4{0.0, 0.0, 0.0, 0.0};
  GM double* c = (GM double*)q.c;
  if (!beta0) {
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = (int)i0 + 16 * tp + (int)g, j = (int)j0 + 16 * tq + (int)s + 4 * r;
          if (i < m && j < n) acc[tp][tq][r] = c[(long long)j * ldc + i];
        }
  }
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ab, bb;
    br_base(p, q, r, ab, bb);
    GM const double* a = (GM const double*)ab;
    GM const double* b = (GM const double*)bb;
    for (int k0 = 0; k0 < K; k0 += 16) {
      double av[4][2], bv[4][2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + 4 * e + (int)s;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int i = (int)i0 + 16 * t + (int)g, j = (int)j0 + 16 * t + (int)g;
          av[e][t] = (k < K && i < m) ? (TA ? a[(long long)i * lda + k] : a[(long long)k * lda + i]) : 0.0;
          bv[e][t] = (k < K && j < n) ? (TB ? b[(long long)k * ldb + j] : b[(long long)j * ldb + k]) : 0.0;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (k0 + 4 * e < K) {                     // wave-uniform
          acc[0][0] = mfma_f64(bv[e][0], av[e][0], acc[0][0]);
          if (two_i) acc[1][0] = mfma_f64(bv[e][0], av[e][1], acc[1][0]);
          if (two_j) acc[0][1] = mfma_f64(bv[e][1], av[e][0], acc[0][1]);
          if (two_i && two_j) acc[1][1] = mfma_f64(bv[e][1], av[e][1], acc[1][1]);
        }
      }
    }
  }
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = (int)i0 + 16 * tp + (int)g, j = (int)j0 + 16 * tq + (int)s + 4 * r;
        if (i < m && j < n) c[(long long)j * ldc + i] = acc[tp][tq][r];
      }
}

// ------------------------------------------------------------------------------------------------
// 16 x 16 x 16 problems, one per wave
// ------------------------------------------------------------------------------------------------
// A 16^3 problem is 2 KiB per operand and four MFMAs.  B (k contiguous) comes by LDS-DMA -- two 16-byte requests per lane into a wave-private
// 2 KiB image [16 columns][8 slots], slot XOR-swizzled by the column on the source side -- and is read back as two ds_read_b128 (the k pairs
// 8u + 2s, + 1 of column g); A (rows contiguous) is read straight into the k order the B fragments dictate: step (u, h) is k = 8u + 2s + h,
// 16 lanes = one 128-byte column.  C leaves as 8-byte stores, 128-byte runs (a 16-row column is all there is).  NN, strided operands.
template <int AUX>
__global__ __launch_bounds__(256) void gemm_f64_p16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) f64x2 lds_all[4][128];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int bidx = logical_block(p) * 4u + wave;
  if (bidx >= p.nbatch) return;
  const unsigned int lane = threadIdx.x & 63u, g = lane & 15u, s = lane >> 4;
  f64x2* img = lds_all[wave];
  const BatchPtrs q = batch_ptrs(p, bidx);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb, ldc = (unsigned int)p.ldc;
  const unsigned int o0 = lane >> 3, sl = lane & 7u;
  const unsigned int vB0 = (o0 * ldb + 2u * (sl ^ (o0 & 7u))) * 8u;                      // columns 0..7; columns 8..15: + 8 ldb (same swizzle: (o + 8) & 7 == o & 7)
  const unsigned int vA = (2u * s * lda + g) * 8u;
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const unsigned int kchunks = (unsigned int)p.k >> 4;
  const unsigned long long total = p.br_count * kchunks;
  f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
  GM double* c = (GM double*)q.c + ((unsigned long long)s * ldc + g);
  if (!beta0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = c[(unsigned long long)(4u * r) * ldc];
  }
  unsigned long long r = 0; unsigned int kc = 0;
  for (unsigned long long t = 0; t < total; ++t) {
    gcptr ar, br;
    br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar), rb = wave_rsrc(br);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)((char*)img), 16, (int)vB0, (int)(kc * 128u), 0, AUX);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)((char*)img + 1024), 16, (int)vB0, (int)(kc * 128u + 64u * ldb), 0, AUX);
    double av[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)             // e = 2u + h: k = 8u + 2s + h
      av[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ra, (int)vA, (int)((kc * 16u + 8u * (e >> 1) + (e & 1)) * 8u * lda), AUX));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f64x2 fb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) fb[u] = img[g * 8u + ((4u * u + s) ^ (g & 7u))];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc = mfma_f64(fb[u][h], av[2 * u + h], acc);
    if (++kc == kchunks) { kc = 0; ++r; }
  }
#pragma unroll
  for (int r2 = 0; r2 < 4; ++r2) st_stream(c + (unsigned long long)(4u * r2) * ldc, acc[r2]);
}

// ------------------------------------------------------------------------------------------------
// blocked kernel: 2-D batches of 32^3 / 64^3 tiles (libxsmm_hip_gemm_batch_strided_2d: C(i, j) = sum_r A(i, r) B(r, j)) -- the operand-reuse regime
// ------------------------------------------------------------------------------------------------
// The structure of gemm_f32_blocked_kernel on the f64 instruction: a WORKGROUP owns a 128 x 128 macro tile of C (4 x 4 problems of 32^3 or
// 2 x 2 of 64^3), wave (wi, wj) its 64 x 64 quarter = 4 x 4 MFMA tiles (128 accumulator registers).  K advances in stages of 16: per stage
// the workgroup brings 4 A blocks (32 rows x 16 k) and 4 B blocks (16 k x 32 columns), 32 KiB, in ONCE by LDS-DMA -- wave w fetches A block w
// and B block w -- into one half of a 64 KiB double buffer; every wave then reads the fragments of its two A and two B blocks (16
// ds_read_b128, all conflict free), issues its share of the NEXT stage's requests and runs 64 MFMAs (4096 matrix-pipe cycles).  One barrier
// per stage.  Two workgroups fit a CU (LDS and registers), so one computes while the other waits at its barrier.
//   A image [16 k][32 rows]: linear (a request is four k rows); fragment (u, h) = the 16-byte row pair (2g, 2g + 1) of k = 8u + 2s + h
//   B image [32 columns][8 slots]: slot XOR-swizzled by the column (source side); fragment u = k pair 4u + s of column 16t + g
// NN, strided 2-D batch, STRIDE or no batch-reduce, K % 16 == 0, 16-byte aligned operands and C.
template <int MB>
__global__ __launch_bounds__(256, 2) void gemm_f64_blocked_kernel(GemmArgs p) {
  constexpr int PPW = 4 / MB;                       // problems per macro-tile edge
  __shared__ __attribute__((aligned(16))) f64x2 lds_all[2][8][256];
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, g = lane & 15u, s = lane >> 4;
  // macro tile of this workgroup: contiguous bands of macro columns per XCD (hardware workgroup x runs on XCD x % 8)
  const unsigned int ni = p.batch_inner, MI = ni / PPW;
  unsigned int wg = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) wg = (wg & 7u) * (gridDim.x >> 3) + (wg >> 3);
  const unsigned int mj = wg / MI, mi = wg - mj * MI;
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb, ldc = (unsigned int)p.ldc;
  // --- request duty of this wave: A block w and B block w of the macro tile, every stage
  const unsigned int pa = mi * PPW + w / MB, pb = mj * PPW + w / MB;
  const __amdgpu_buffer_rsrc_t ra = wave_rsrc((gcptr)p.a + (long long)pa * p.bs_a + 256ull * (w % MB));
  const __amdgpu_buffer_rsrc_t rb = wave_rsrc((gcptr)p.b + (long long)pb * p.bs_b + 256ull * ldb * (w % MB));
  const unsigned int vA = (s * lda + 2u * g) * 8u;                                       // request x: k = 4x + s, row pair g
  const unsigned int ob = lane >> 3, sl = lane & 7u;
  const unsigned int vB = (ob * ldb + 2u * (sl ^ (ob & 7u))) * 8u;                       // request x: column 8x + ob, slot sl <- k pair sl ^ (column & 7)
  const long long brs_a = p.br_mode == 3 ? p.br_stride_a : 0, brs_b = p.br_mode == 3 ? p.br_stride_b : 0;
  const unsigned int kstages = (unsigned int)p.k >> 4;
  const unsigned long long total = p.br_count * kstages;
  auto issue = [&](unsigned long long r, unsigned int kc, int buf) {
    const unsigned int sa = (unsigned int)((long long)r * brs_a) + kc * 128u * lda, sb = (unsigned int)((long long)r * brs_b) + kc * 128u;
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)((char*)&lds_all[buf][w][0] + 1024 * x), 16, (int)vA, (int)(sa + 32u * x * lda), 0, 0);
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)((char*)&lds_all[buf][4 + w][0] + 1024 * x), 16, (int)vB, (int)(sb + 64u * x * ldb), 0, 0);
  };
  // --- compute duty: the 64 x 64 quarter (wi, wj): rows 32 (2 wi + ib) + 2g + par, columns 32 (2 wj + jb) + 16 t + s + 4 r
  const unsigned int wi = w & 1u, wj = w >> 1;
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  f64x4 acc[2][2][2][2];                           // [ib][par][jb][t]
  gptr cblk[2][2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const unsigned int bi = mi * PPW + (2 * wi + ib) / MB, bj = mj * PPW + (2 * wj + jb) / MB;
      cblk[ib][jb] = (gptr)p.c + (long long)bi * p.bs_c + (long long)bj * p.bs_c2 + 8ull * (32ull * ((2 * wi + ib) % MB) + 32ull * ((2 * wj + jb) % MB) * ldc + 2u * g + (unsigned long long)s * ldc);
    }
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[ib][0][jb][t] = f64x4{0.0, 0.0, 0.0, 0.0}; acc[ib][1][jb][t] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (!beta0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f64x2 v = *(GM const f64x2*)(cblk[ib][jb] + 8ull * (16u * t + 4u * r) * ldc);
            acc[ib][0][jb][t][r] = v.x; acc[ib][1][jb][t][r] = v.y;
          }
        }
      }
  unsigned long long r = 0; unsigned int kc = 0;
  if (total != 0) issue(0, 0, 0);
  for (unsigned long long t = 0; t < total; ++t) {
    const int buf = (int)(t & 1ull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA is ordered by the issuing wave's vmcnt only
    __syncthreads();                                   // ... and by a barrier for the other waves: stage t's eight blocks are in LDS
    f64x2 fa[2][4], fb[2][2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int e = 0; e < 4; ++e) fa[ib][e] = lds_all[buf][2 * wi + ib][(8u * (e >> 1) + 2u * s + (e & 1)) * 16u + g];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int u = 0; u < 2; ++u) fb[jb][tt][u] = lds_all[buf][4 + 2 * wj + jb][(16u * tt + g) * 8u + ((4u * u + s) ^ (g & 7u))];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (++kc == kstages) { kc = 0; ++r; }
    if (t + 1 < total) issue(r, kc, buf ^ 1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
                acc[ib][par][jb][tt] = mfma_f64(fb[jb][tt][u][h], fa[ib][2 * u + h][par], acc[ib][par][jb][tt]);
  }
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2)
          *(GM f64x2*)(cblk[ib][jb] + 8ull * (16u * tt + 4u * r2) * ldc) = f64x2{acc[ib][0][jb][tt][r2], acc[ib][1][jb][tt][r2]};
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool f64_stream_ok(const GemmArgs& a) {
  if ((a.m % 32) || (a.n % 32) || (a.k % 32) || a.k <= 0) return false;
  if ((a.list_a && !a.lists_aligned16) || a.br_mode == 1 || a.br_mode == 2) return false;      // pointer / offset lists live on the device: alignment unknown here (unless the library built them)
  unsigned long long bits = (unsigned long long)((long long)a.lda * 8) | (unsigned long long)((long long)a.ldb * 8);
  if (!a.list_a) bits |= (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b;
  if (a.br_mode == 3) bits |= (unsigned long long)a.br_stride_a | (unsigned long long)a.br_stride_b;
  if (bits & 15ull) return false;
  return a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22);            // 32-bit byte offsets inside a chunk
}

static bool f64_strided_aligned(const GemmArgs& a) {
  if (a.list_a || a.br_mode == 1 || a.br_mode == 2) return false;
  unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)((long long)a.lda * 8) | (unsigned long long)((long long)a.ldb * 8);
  if (a.br_mode == 3) bits |= (unsigned long long)a.br_stride_a | (unsigned long long)a.br_stride_b;
  return (bits & 15ull) == 0ull && a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22);
}
// 16^3 problems: NN, one problem per wave (8-byte accesses of A and C: alignment of B only)
static bool f64_p16_ok(const GemmArgs& a) {
  if (a.m != 16 || a.n != 16 || (a.k % 16) || a.k <= 0 || (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B))) return false;
  if (a.list_a || a.br_mode == 1 || a.br_mode == 2) return false;
  unsigned long long bits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)((long long)a.ldb * 8);
  if (a.br_mode == 3) bits |= (unsigned long long)a.br_stride_b;
  return (bits & 15ull) == 0ull && a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22);
}
// 2-D batches of 32^3 / 64^3 tiles: the blocked kernel (whole macro tiles of 128 x 128, 32-bit request offsets)
static bool f64_blocked_ok(const GemmArgs& a) {
  if (!a.batch_inner || a.m != a.n || (a.m != 32 && a.m != 64) || (a.k % 16) || a.k <= 0) return false;
  if (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B)) return false;
  if (!f64_strided_aligned(a)) return false;
  const unsigned int ppw = 128u / (unsigned int)a.m, ni = a.batch_inner, nj = a.nbatch / a.batch_inner;
  if (ni % ppw || nj % ppw) return false;
  if ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)a.bs_c2 | (unsigned long long)((long long)a.ldc * 8)) & 15ull) != 0ull) return false;
  const unsigned long long span_a = (a.br_mode == 3 ? (unsigned long long)std::max<long long>(a.br_stride_a, 0) * a.br_count : 0ull) + (unsigned long long)a.lda * a.k * 8ull;
  const unsigned long long span_b = (a.br_mode == 3 ? (unsigned long long)std::max<long long>(a.br_stride_b, 0) * a.br_count : 0ull) + (unsigned long long)a.ldb * a.n * 8ull;
  if (a.br_mode == 3 && (a.br_stride_a < 0 || a.br_stride_b < 0)) return false;
  return span_a < (1ull << 31) && span_b < (1ull << 31);
}

const char* gemm_f64_kernel_name(const libxsmm_gemm_descriptor& d) {
  const bool whole = (d.m % 32) == 0 && (d.n % 32) == 0 && (d.k % 32) == 0 && d.k > 0 && (d.lda % 2) == 0 && (d.ldb % 2) == 0 &&
    !(d.flags & (LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET));
  return whole ? "gemm_f64_stream_kernel" : "gemm_f64_ragged_kernel";
}

int launch_gemm_f64(const GemmArgs& a_in, void* stream, const char** kernel_name) {
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a = a_in;
  const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B;
  a.tiles_m = (a.m + 31) / 32; a.tiles_n = (a.n + 31) / 32; a.map2d_shift = 0;
  const long long tiles = (long long)a.tiles_m * a.tiles_n * (long long)a.nbatch;
  if (tiles >= (1ll << 31)) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned int)((tiles + 3) / 4));
  constexpr int pol_env = -1;
  // cache policy as for the f32 kernels (DESIGN decision 8): non-temporal requests only when the operands cannot be cache resident -- one launch
  // moves more than the 256 MiB Infinity Cache holds -- or the caller declared a streaming pass (libxsmm_hip_set_streaming_hint(2)); never with hint 1
  const unsigned long long moved = (unsigned long long)a.nbatch * ((unsigned long long)a.br_count * (unsigned long long)a.k * (unsigned long long)(a.m + a.n) + (unsigned long long)a.m * a.n) * 8ull;
  bool nt = a.stream_hint == 2 || (a.stream_hint == 0 && (moved > (256ull << 20) || rt_recent_operands_exceed_cache(a.a, moved, a.c)));
  if (pol_env == 0) nt = false; else if (pol_env == 1) nt = true;
  constexpr bool blocked_off = false;
  if (!blocked_off && f64_blocked_ok(a)) {
    const unsigned int ppw = 128u / (unsigned int)a.m;
    const dim3 bgrid((a.batch_inner / ppw) * ((a.nbatch / a.batch_inner) / ppw));
    if (a.m == 32) { if (kernel_name) *kernel_name = "gemm_f64_blocked_kernel<1>"; hipLaunchKernelGGL((gemm_f64_blocked_kernel<1>), bgrid, dim3(256), 0, st, a); }
    else { if (kernel_name) *kernel_name = "gemm_f64_blocked_kernel<2>"; hipLaunchKernelGGL((gemm_f64_blocked_kernel<2>), bgrid, dim3(256), 0, st, a); }
    return (int)hipGetLastError();
  }
  if (f64_p16_ok(a)) {
    if (kernel_name) *kernel_name = "gemm_f64_p16_kernel";
    const dim3 pgrid((a.nbatch + 3u) / 4u);
    if (nt) hipLaunchKernelGGL((gemm_f64_p16_kernel<2>), pgrid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_f64_p16_kernel<0>), pgrid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
  }
  constexpr bool s64_off = false;
  if (!s64_off && !ta && !tb && (a.m % 64) == 0 && (a.n % 64) == 0 && f64_stream_ok(a)) {       // 64 x 64 tiles: one wave per tile, no operand byte fetched twice
    a.tiles_m = a.m / 64; a.tiles_n = a.n / 64;
    const long long t64 = (long long)a.tiles_m * a.tiles_n * (long long)a.nbatch;
    if (kernel_name) *kernel_name = "gemm_f64_stream64_kernel";
    // waves walk several tiles once the launch exceeds what is resident at two waves per SIMD (256 CUs x 8): LIBXSMM_HIP_F64_S64_WAVES overrides the cap
    constexpr long long cap = 2048ll;
    const long long per_wave = (t64 + cap - 1) / cap, waves = (t64 + per_wave - 1) / per_wave;
    if (a.br_count == 0) { if (kernel_name) *kernel_name = "gemm_f64_stream_kernel"; goto f64_small_tiles; }      // beta-only call: the 32 x 32 kernel has that path
    if (nt) hipLaunchKernelGGL((gemm_f64_stream64_kernel<2>), dim3((unsigned int)((waves + 3) / 4)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_f64_stream64_kernel<0>), dim3((unsigned int)((waves + 3) / 4)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
  }
  f64_small_tiles:
  a.tiles_m = (a.m + 31) / 32; a.tiles_n = (a.n + 31) / 32;
  if (f64_stream_ok(a)) {
    if (kernel_name) *kernel_name = "gemm_f64_stream_kernel";
#define LAUNCH_F64S_(TA_, TB_) do { if (nt) hipLaunchKernelGGL((gemm_f64_stream_kernel<TA_, TB_, 2>), grid, dim3(256), 0, st, a); \
                                    else hipLaunchKernelGGL((gemm_f64_stream_kernel<TA_, TB_, 0>), grid, dim3(256), 0, st, a); } while (0)
    if (!ta && !tb) LAUNCH_F64S_(false, false); else if (ta && !tb) LAUNCH_F64S_(true, false); else if (!ta && tb) LAUNCH_F64S_(false, true); else LAUNCH_F64S_(true, true);
#undef LAUNCH_F64S_
    return (int)hipGetLastError();
  }
  if (kernel_name) *kernel_name = "gemm_f64_ragged_kernel";
  if (!ta && !tb) hipLaunchKernelGGL((gemm_f64_ragged_kernel<false, false>), grid, dim3(256), 0, st, a);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_f64_ragged_kernel<true, false>), grid, dim3(256), 0, st, a);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_f64_ragged_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_f64_ragged_kernel<true, true>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

}  // namespace xamd
