// gemm_wgp.hpp -- 16-bit (bf16 / f16) GEMMs and 8-bit weights x bf16 of SEVERAL TILES, ONE PROBLEM PER WORKGROUP with the whole problem staged in LDS (round 5); the
// geometry, the tile ownership (wgp_deal / wgp_waves) and the register bounds are shared with the 8-bit x 8-bit kernel (gemm_wgp8_kernels.hip).
// Where the family stands at the end of round 5 (tools/shape_scan.py, profiles/r05_shape_scan_640MB.jsonl): m = n = k = 40 .. 120 at 0.56-0.72 of the HBM roofline for
// every 16-bit / 8-bit type -- 80-87 % of what the whole-tile streaming kernels reach (0.83).
//
// What was wrong with the wave-per-tile kernel on shapes like 40^3 and 72^3 (gemm_mfma_bf16_kernel<2,2>: 0.49 / 0.35 of the HBM roofline, 0.57 / 0.43 in its bounded
// form): a wave walks its K chunks one after the other -- request a 32-deep panel, wait for it, multiply, request the next -- so a 72^3 problem is three memory
// round trips per wave with nothing in flight in between, the four waves of a problem fetch overlapping panels, and they cover 128 x 128 with 64 x 64 tiles.
// The counters said the same: traffic 1.02-1.19 x algorithmic, MFMA work 4 x the problem's (profiles/r04_pmc_traffic.json, r04_mfma_busy.json) -- latency, not bytes.
//
// Here the operand BLOCKS of a problem -- A as [k/2][lda] dwords (VNNI-2 pairs), B as [n][ldb] halves: both contiguous in memory -- are brought into LDS as what
// they are: rows of A and columns of B cut into 16-byte pieces, one piece per lane and request (global -> LDS DMA, no registers, every request a full 16 bytes of a
// row that is read exactly once), ALL of them issued before the first wait.  One round trip per problem and batch-reduce block.  The four waves then deal the
// problem's ceil(m/32) x ceil(n/32) tiles of 32 x 32 among themselves (72^3: nine tiles = 96 x 96 covered instead of 128 x 128) and multiply out of LDS:
//   A fragment of row i, k pairs kp..kp+3:   four ds_read_b32 at (kp + e) * RP + i        (lanes along i: conflict free)
//   B fragment of column j, 8 consecutive k: one ds_read_b128 at j * CP + 16 * piece      (CP = 16 bytes x pieces per column; odd piece counts are conflict free)
// Overlap comes from the workgroups a CU holds at once (72^3: 20 KiB of LDS each, five to seven resident; 40^3: 6.4 KiB, eight): while one multiplies the others'
// requests are in flight.  Results: tile_init / tile_store of gemm_tile.hpp -- any beta, fused column bias / ReLU (+ bitmask) / sigmoid -- in the matrix core's
// summation order (the same chunking as the wave-per-tile kernel: k in steps of 16, batch-reduce blocks in order).
//
// Taken by launch_gemm for 1-D batches (strided or pointer lists are not needed: strided only) when every piece request lies inside its operand block:
// m % 4 == 0, k % 8 == 0, lda % 4 == 0, ldb % 8 == 0, 16-byte aligned blocks, 2 <= tiles <= 12 or 4 x 4 tiles, LDS image <= 64 KiB.  Everything else keeps the wave-per-tile
// kernel.  What was added after the first form (each step an A/B on the GPU, DESIGN.md decision 34): register bounds = the most waves per SIMD without scratch; STRIPS -- a
// wave owns a tile row or column and reads the shared fragment once per k step, the workgroup has as many waves as strips (2 x 2 tiles: two waves with a row each; 3 x 3:
// three; 4 x 4: four); the fused column bias and beta * C through LDS images like the operands; base pointers in SGPRs.
// [ref: the loop being computed is src/generator_gemm_reference_impl.c:2127-2170 (bf16 -> f32), :2367-2419 (bf16 -> bf16), :2025-2124 (f16)]
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "internal.hpp"
#include "gemm_device.hpp"
#include "gemm_tile.hpp"
#include "gemm_8bit.hpp"

#pragma clang fp contract(off)


namespace xamd {

struct Wgp16Geo {
  unsigned int rp;          // dwords per k-pair row of the A image = 4 x pieces per row
  unsigned int ppr;         // 16-byte pieces per k-pair row of A
  unsigned int ppc;         // 16-byte pieces per column of B
  unsigned int a_pieces, b_pieces;      // total pieces of one block
  unsigned int a_img;       // bytes of the A image (whole 1 KiB request slots)
  unsigned int bias_off, bias_dw;      // fused column bias: LDS offset of its image and its dwords (m elements of C's type; 0 = no bias)
  unsigned int c_off, c_ppc, c_pieces; // beta = 1: LDS offset of the image of C ([n][m] elements, compact), 16-byte pieces per column, pieces in all (0 = C by element loads)
};

// AK = -1: 16-bit A (VNNI-2 dwords).  AK = 0..4: 8-bit WEIGHTS x bf16 activations (KIND of gemm_w8_bf16_kernel: 0 / 1 BF8 / HF8 in VNNI-2 byte pairs, 2 / 3 flat, 4 int8 with
// one f32 scale per row) -- the A block is a BYTE image ([k/2][m][2] or [k][m], lda == m) that comes in as a linear copy (whole 16-byte pieces of the packed block) and is
// turned into the bf16 pairs the reference multiplies with on the way out of LDS (w8_pair_to_bf16: exact for the 8-bit floats, one rounding for the scaled int8).
// Register bounds = waves per SIMD the compiler must leave room for (__launch_bounds__' second argument).  LDS never limits these kernels (6-20 KiB per workgroup of
// 160 KiB); resident workgroups are what hides a problem's single round trip, so every form is bounded to the most waves that compile WITHOUT scratch:
// one tile per wave 8 (<= 64 registers), two 6 (<= 80), three 5 (<= 96).  Measured: profiles/r05_wgp_bound5.jsonl (three tiles), r05_wgp_waves.jsonl (one / two).
// Round 5, last session: with the problem's base pointers in SGPRs and the store addresses formed behind the multiply loop, three tiles per wave fit 80 registers without
// scratch (76-79), two fit 64 (57-60), four fit 96: the bounds below.  A variant that does not fit its bound spills -- and scratch costs a third of the kernel -- so
// tests/test_kernel_resources_cpu.py pins "no scratch" on the built library.
#ifndef WGP_W1
#define WGP_W1 8
#endif
#ifndef WGP_W2
#define WGP_W2 8
#endif
#ifndef WGP_W3
#define WGP_W3 5
#endif
#ifndef WGP_W3S
#define WGP_W3S 6
#endif
#ifndef WGP_W4
#define WGP_W4 5
#endif
#define WGP_WAVES(T) ((T) == 4 ? WGP_W4 : (T) == 3 ? WGP_W3 : (T) == 2 ? WGP_W2 : WGP_W1)      // round-robin deal (three tiles: 83-88 registers)
#define WGP_WAVES_D(T, D, AK) ((T) == 3 && (D) != 0 ? (((D) == 2 && (AK) >= 2) ? 5 : WGP_W3S) : WGP_WAVES(T))      // strips of three: six waves (column strips of flat 8-bit weights: 81-84 registers, five)
// Which tiles a wave owns.  DEAL 0: round robin (tile w + 4 t).  DEAL 1: wave w owns tile ROW w, its tiles t are the tile columns -- one A fragment per k step feeds all of
// them.  DEAL 2: wave w owns tile COLUMN w (one B fragment).  The strips are chosen by the launcher when they do not lengthen the critical path (3 or 4 strips of
// ceil(tiles / 4) tiles: 72^3 and 96^3 are 3 x 3 -- three waves with a row each instead of 3 + 3 + 2 + 1 tiles with nothing in common).
template <int DEAL>
__device__ __forceinline__ bool wgp_tile_of(unsigned int w, unsigned int nw, unsigned int t, unsigned int tiles_m, unsigned int tiles_n, unsigned int& ti, unsigned int& tj) {
  if constexpr (DEAL == 1) { ti = w; tj = t; return w < tiles_m && t < tiles_n; }
  else if constexpr (DEAL == 2) { ti = t; tj = w; return t < tiles_m && w < tiles_n; }
  else { const unsigned int id = w + nw * t; tj = id / tiles_m; ti = id - tj * tiles_m; return id < tiles_m * tiles_n; }
}

template <bool F16, int TPW, int AK = -1, int DEAL = 0>
__global__ __launch_bounds__(256, WGP_WAVES_D(TPW, DEAL, AK)) void gemm_wgp16_kernel(GemmArgs p, Wgp16Geo g) {
  extern __shared__ __attribute__((aligned(16))) char lds_wgp[];
  const unsigned int TS = (DEAL == 0 && TPW > 1) ? 4u : blockDim.x >> 6;                        // the waves of the workgroup share the problem: four, or one per strip / tile when those are fewer
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int bidx = blockIdx.x;
  BatchPtrs q = batch_ptrs(p, bidx);
  // (the problem's five base pointers are wave-uniform but come out of 64-bit vector multiplies: ten VGPRs for the life of the kernel unless they are moved to SGPRs)
  q.a = (gcptr)(size_t)uniform_u64((unsigned long long)(size_t)q.a); q.b = (gcptr)(size_t)uniform_u64((unsigned long long)(size_t)q.b);
  q.c = (gptr)(size_t)uniform_u64((unsigned long long)(size_t)q.c); q.d = (gcptr)(size_t)uniform_u64((unsigned long long)(size_t)q.d);
  q.mask = (GM unsigned char*)(size_t)uniform_u64((unsigned long long)(size_t)q.mask);
  char* const img_a = lds_wgp;
  char* const img_b = img_a + g.a_img;
  const unsigned int tiles_m = (unsigned int)p.tiles_m, tiles_n = (unsigned int)p.tiles_n;
  f32x16 acc[TPW];
  TileCtx tc[TPW];
  bool mine[TPW];
  static_for<TPW>([&](auto tt) {
    constexpr int t = tt.value;
    unsigned int ti, tj;
    mine[t] = wgp_tile_of<DEAL>(w, TS, (unsigned int)t, tiles_m, tiles_n, ti, tj);
    tc[t].i = (int)(32u * ti + li); tc[t].j0 = (int)(32u * tj); tc[t].h = (int)h; tc[t].ivalid = tc[t].i < p.m;
  });
  // the accumulators' start values (zeros, the column bias, beta * C) are fetched AFTER the first block's requests have been issued: a wave that waited for its bias
  // first would put a second memory round trip in front of the one the whole kernel is built around
  // The column bias is the SAME few bytes for every workgroup of the launch: fetched per tile by every wave it queues up on one L2 channel, and the wave waits for it in
  // front of the barrier (measured: 25 of 120 us on 72^3, profiles/r05_fused_parts.jsonl).  One wave brings it in with the first block's requests (a dword per lane into
  // an LDS image) and the tiles pick it up behind the barrier.
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  auto init_start = [&]() {                                        // beta * C: requested with the first block, no bias yet
    static_for<TPW>([&](auto tt) {
      constexpr int t = tt.value;
      if (mine[t]) tile_init<false, false, true, true>(acc[t], p, q, tc[t]);
    });
  };
  // (IEEE halves, round 6: the finished start value -- beta * C, the bias, or their f32 sum -- is rounded to f16, the rule of the reference's F16 loop: tile_init<.., F16S>)
  auto round_start = [&]() {
    if constexpr (F16) {
      if (!beta0 || g.bias_dw) {
        static_for<TPW>([&](auto tt) { constexpr int t = tt.value;
          if (mine[t]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = (float)(_Float16)acc[t][r];
          } });
      }
    }
  };
  auto init_bias = [&]() {                                         // behind the first barrier: C from its LDS image, bias (+ beta * C) in the order of tile_init
    if (g.c_pieces) {
      const unsigned int m = (unsigned int)p.m;
      static_for<TPW>([&](auto tt) {
        constexpr int t = tt.value;
        if (mine[t]) {
          const char* img = lds_wgp + g.c_off;
          if (p.c_type == LIBXSMM_DATATYPE_F32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const unsigned int j = (unsigned int)(tc[t].j0 + jl_of(r, tc[t].h)); acc[t][r] = (tc[t].ivalid && j < (unsigned int)p.n) ? ((const float*)img)[j * m + (unsigned int)tc[t].i] : 0.0f; }
          } else if (p.c_type == LIBXSMM_DATATYPE_F16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const unsigned int j = (unsigned int)(tc[t].j0 + jl_of(r, tc[t].h)); acc[t][r] = (tc[t].ivalid && j < (unsigned int)p.n) ? (float)((const _Float16*)img)[j * m + (unsigned int)tc[t].i] : 0.0f; }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const unsigned int j = (unsigned int)(tc[t].j0 + jl_of(r, tc[t].h)); acc[t][r] = (tc[t].ivalid && j < (unsigned int)p.n) ? bf16_to_f32(((const unsigned short*)img)[j * m + (unsigned int)tc[t].i]) : 0.0f; }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        }
      });
    }
    if (!g.bias_dw) return;
    static_for<TPW>([&](auto tt) {
      constexpr int t = tt.value;
      if (mine[t]) {
        const char* img = lds_wgp + g.bias_off;
        float bias = 0.0f;
        if (tc[t].ivalid) bias = p.c_type == LIBXSMM_DATATYPE_F32 ? ((const float*)img)[tc[t].i] : p.c_type == LIBXSMM_DATATYPE_F16 ? (float)((const _Float16*)img)[tc[t].i] : bf16_to_f32(((const unsigned short*)img)[tc[t].i]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = beta0 ? bias : bias + acc[t][r];
      }
    });
  };
  float scf[TPW];
  static_for<TPW>([&](auto tt) { constexpr int t = tt.value;
    scf[t] = (AK == 4 && tc[t].ivalid) ? ((GM const float*)(p.a_scf + (long long)bidx * p.bs_scf))[tc[t].i] : 1.0f; });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int kchunks = ((unsigned int)p.k + 31u) >> 5, kgroups = (unsigned int)p.k >> 3;      // 8-deep k groups (k % 8 == 0)
  auto issue = [&](unsigned long long r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    // ---- all requests of the block, dealt round-robin to the four waves: request x fills the 1 KiB slot x of its image, lane = piece 64 x + lane
    for (unsigned int x = w; x * 64u < g.a_pieces; x += TS) {
      const unsigned int P = 64u * x + lane;
      if (P < g.a_pieces) {
        if constexpr (AK >= 0) __builtin_amdgcn_global_load_lds((GM const void*)(ar + 16ull * P), (lds_vptr)(img_a + 1024u * x), 16, 0, 0);      // the packed byte image, piece by piece
        else {
          const unsigned int kp = P / g.ppr, pc = P - kp * g.ppr;
          __builtin_amdgcn_global_load_lds((GM const void*)(ar + ((unsigned long long)kp * lda + 4u * pc) * 4ull), (lds_vptr)(img_a + 1024u * x), 16, 0, 0);
        }
      }
    }
    for (unsigned int x = w; x * 64u < g.b_pieces; x += TS) {
      const unsigned int P = 64u * x + lane;
      if (P < g.b_pieces) {
        const unsigned int col = P / g.ppc, pc = P - col * g.ppc;
        __builtin_amdgcn_global_load_lds((GM const void*)(br + ((unsigned long long)col * ldb + 8u * pc) * 2ull), (lds_vptr)(img_b + 1024u * x), 16, 0, 0);
      }
    }
  };
  if (p.br_count) issue(0);
  if (g.bias_dw && w == TS - 1u) {                                // (the last wave has the fewest block requests)
    for (unsigned int x = 0; x * 64u < g.bias_dw; ++x) {
      const unsigned int P = 64u * x + lane;
      if (P < g.bias_dw) __builtin_amdgcn_global_load_lds((GM const void*)(q.d + 4ull * P), (lds_vptr)(lds_wgp + g.bias_off + 256u * x), 4, 0, 0);
    }
  }
  // beta = 1: C comes in like the operands -- whole 16-byte pieces of its columns by LDS-DMA (72^3 bf16: ten requests per workgroup instead of 432 two-byte loads, which
  // kept the address unit busy for three times the kernel's beta = 0 duration: 0.28 of the roofline, profiles/r05_wgp_beta1.jsonl)
  if (g.c_pieces) {
    const unsigned int ces = p.c_type == LIBXSMM_DATATYPE_F32 ? 4u : 2u;
    for (unsigned int x = w; x * 64u < g.c_pieces; x += TS) {
      const unsigned int P = 64u * x + lane;
      if (P < g.c_pieces) {
        const unsigned int col = P / g.c_ppc, pc = P - col * g.c_ppc;
        __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)q.c + (unsigned long long)col * (unsigned int)p.ldc * ces + 16u * pc), (lds_vptr)(lds_wgp + g.c_off + 1024u * x), 16, 0, 0);
      }
    }
  } else init_start();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();
  init_bias();                                                   // (an empty chain: C = beta * C (+ bias))
  round_start();
  // (chains with TWO pairs of images -- block r + 1 in flight while block r is multiplied, one barrier per block -- measured and not adopted: the second pair halves the
  //  workgroups a CU holds; 72^3 x 4 blocks 0.60 -> 0.52, x 16 blocks 0.60 -> 0.45, 40^3 x 4 0.64 -> 0.60, profiles/r05_wgp_chains.jsonl.  Resident workgroups beat overlap
  //  inside a workgroup, every time it was tried this round.)
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    if (r != 0) {
      wg_barrier();                                              // everybody has read the previous block's images
      issue(r);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wg_barrier();
    }
    // ---- multiply out of LDS: my tiles, K in chunks of 32 (two MFMA steps of 16)
    if constexpr (DEAL != 0) {
      // a strip of tiles: K outside, the strip's tiles inside -- the shared fragment is read (and, for 8-bit weights, converted) once per k step
      constexpr int TA = DEAL == 1 ? 1 : TPW, TB = DEAL == 2 ? 1 : TPW;
      if (mine[0]) {                                               // (wave-uniform: a wave without a strip has nothing to do)
        const unsigned int i0 = (unsigned int)tc[0].i, j0 = (unsigned int)tc[0].j0 + li;      // tile t of the strip: + 32 t on the side that is not shared
        for (unsigned int kc = 0; kc < kchunks; ++kc) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const unsigned int kg = 4u * kc + 2u * (unsigned int)s + h;
            const bool ok = kg < kgroups;
            const unsigned int kgc = ok ? kg : 0u;
            u32x4 af[TA], bfr[TB];
#pragma unroll
            for (int t = 0; t < TA; ++t) {
              const unsigned int i = i0 + 32u * (unsigned int)t;
              if constexpr (AK < 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) af[t][e] = ((const unsigned int*)img_a)[(4u * kgc + (unsigned int)e) * g.rp + i];
              } else if constexpr (AK < 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) af[t][e] = w8_pair_to_bf16<AK>(*((const unsigned short*)img_a + (4u * kgc + (unsigned int)e) * g.rp + i), 1.0f);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const unsigned char* b0 = (const unsigned char*)img_a + (8u * kgc + 2u * (unsigned int)e) * g.rp + i;
                  af[t][e] = w8_pair_to_bf16<AK>((unsigned int)b0[0] | ((unsigned int)b0[g.rp] << 8), scf[t]);
                }
              }
              if (!ok) af[t] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
              bfr[t] = *(const u32x4*)(img_b + (size_t)(j0 + 32u * (unsigned int)t) * (g.ppc * 16u) + 16u * kgc);
              if (!ok) bfr[t] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t)
              if (mine[t]) acc[t] = mfma_16bit<F16>(bfr[DEAL == 2 ? 0 : t], af[DEAL == 1 ? 0 : t], acc[t]);
          }
        }
      }
    } else
    static_for<TPW>([&](auto tt) {
      constexpr int t = tt.value;
      if (mine[t]) {
        const unsigned int ti = (unsigned int)tc[t].i >> 5, tj = (unsigned int)tc[t].j0 >> 5;
        const unsigned int* const arow = (const unsigned int*)img_a + 32u * ti + li;              // + kp * rp
        const char* const bcol = img_b + (size_t)(32u * tj + li) * (g.ppc * 16u);                  // + 16 * piece
        for (unsigned int kc = 0; kc < kchunks; ++kc) {
          u32x4 af[2], bfr[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const unsigned int kg = 4u * kc + 2u * (unsigned int)s + h;           // this lane's 8-deep k group of the step
            const bool ok = kg < kgroups;                                         // (k % 8 == 0: a group is whole or absent)
            const unsigned int kgc = ok ? kg : 0u;                                // absent groups read group 0 (inside the image) and are zeroed
            if constexpr (AK < 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) af[s][e] = arow[(4u * kgc + (unsigned int)e) * g.rp];
            } else if constexpr (AK < 2) {        // byte pairs [k/2][m][2]: two bytes of my row per k pair
#pragma unroll
              for (int e = 0; e < 4; ++e) af[s][e] = w8_pair_to_bf16<AK>(*((const unsigned short*)img_a + (4u * kgc + (unsigned int)e) * g.rp + 32u * ti + li), 1.0f);
            } else {                              // flat [k][m]: the even and the odd k of a pair are m bytes apart
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned char* b0 = (const unsigned char*)img_a + (8u * kgc + 2u * (unsigned int)e) * g.rp + 32u * ti + li;
                af[s][e] = w8_pair_to_bf16<AK>((unsigned int)b0[0] | ((unsigned int)b0[g.rp] << 8), scf[t]);
              }
            }
            bfr[s] = *(const u32x4*)(bcol + 16u * kgc);
            if (!ok) { af[s] = u32x4{0u, 0u, 0u, 0u}; bfr[s] = u32x4{0u, 0u, 0u, 0u}; }
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) acc[t] = mfma_16bit<F16>(bfr[s], af[s], acc[t]);
        }
      }
    });
  }
  // (round 5, measured and not adopted -- profiles/r05_wgp16_c_image_not_adopted.jsonl: the results through an LDS image of C and out as whole 16-byte pieces.  The timing
  //  ablation had put the element stores at 41 of 147 us on 40^3, but the image costs LDS -- 96^3: 54 instead of 36 KiB per workgroup -- and a second barrier: 40^3 0.54 ->
  //  0.51, 72^3 0.53 -> 0.47, 96^3 0.65 -> 0.40.  Likewise a wave per problem (no barrier at all, a quarter of the workgroups: 40^3 0.56 -> 0.48, 48^3 0.67 -> 0.36,
  //  r05_wave_per_problem_not_adopted.jsonl), persistent workgroups with two images in flight (0.56 -> 0.46, r05_wgp16_forms.jsonl) and two problems per workgroup, two
  //  waves each (half the workgroups: 0.536 -> 0.542, 48^3 0.65 -> 0.64: nothing, r05_wgp16_two_problems_per_wg_not_adopted.jsonl): what these shapes need is many
  //  short workgroups in different phases, which is exactly what the hardware's own workgroup scheduler provides.  40^3 stays at 0.54 - 0.57 in EVERY form, the
  //  wave-per-tile kernel included.)
  // (the row index is made opaque here: addresses of the stores must not be formed -- and kept in registers -- in front of the multiply loop, where beta * C's loads
  //  use the same expressions; two 64-bit values spilled that way were all that stood between three tiles per wave and six waves per SIMD)
#pragma unroll
  for (int t = 0; t < TPW; ++t) { int opaque_i = tc[t].i; asm volatile("" : "+v"(opaque_i)); tc[t].i = opaque_i; }
  static_for<TPW>([&](auto tt) {
    constexpr int t = tt.value;
    if (mine[t]) tile_store<false, false, false>(acc[t], p, q, tc[t]);
  });
}

// rows / columns beyond m / n of a tile read LDS beyond their operand's rows (another k pair's row, the other image, or nothing): they feed results nobody stores,
// and an LDS read beyond the allocation returns zero by definition -- no fault is possible on that side.
static inline bool wgp16_shape_ok(const GemmArgs& a, Wgp16Geo& g, unsigned int& lds_bytes, int& tpw, int ak = -1) {
  constexpr bool off = false;
  if (off) return false;
  if (a.batch_inner || (a.list_a && !a.lists_aligned16) || a.br_mode == 1 || a.br_mode == 2 || a.vnni_c) return false;      // 1-D batches: strided, or pointer lists the library built itself (the coalescing queue: every pointer known to be 16-byte aligned); plain / STRIDE batch-reduce
  if (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B)) return false;
  if (ak < 0 && !(a.flags & LIBXSMM_GEMM_FLAG_VNNI_A)) return false;
  if ((a.m & 3) || (a.k & 7) || (a.lda & 3) || (a.ldb & 7) || a.k <= 0) return false;
  if (ak >= 0 && (a.lda != a.m || (((long long)a.m * a.k) & 15))) return false;       // 8-bit weights: the packed byte image of the block, whole 16-byte pieces of it
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0);
  if (bits & 15ull) return false;
  const int tiles = ((a.m + 31) / 32) * ((a.n + 31) / 32);
  if (tiles < 2 || (tiles > 12 && !(tiles == 16 && a.m > 96 && a.n > 96))) return false;      // (sixteen: 4 x 4 tiles only -- a tile row of four per wave)
  g.ppr = (unsigned int)a.m / 4u; g.rp = (unsigned int)a.m;
  g.ppc = (unsigned int)a.k / 8u;
  g.a_pieces = ak >= 0 ? (unsigned int)(((long long)a.m * a.k) / 16) : ((unsigned int)a.k / 2u) * g.ppr; g.b_pieces = (unsigned int)a.n * g.ppc;
  g.a_img = ((g.a_pieces + 63u) / 64u) * 1024u;
  lds_bytes = g.a_img + ((g.b_pieces + 63u) / 64u) * 1024u;
  g.bias_off = 0; g.bias_dw = 0;
  if (a.colbias) {                                               // m elements of C's type as whole dwords (m % 4 == 0) behind the operand images
    if (((unsigned long long)(size_t)a.d | (unsigned long long)a.bs_d) & 3ull) return false;
    const unsigned int es = a.c_type == LIBXSMM_DATATYPE_F32 ? 4u : 2u;
    g.bias_off = lds_bytes; g.bias_dw = (unsigned int)a.m * es / 4u;
    lds_bytes += ((g.bias_dw + 63u) / 64u) * 256u;
  }
  g.c_off = 0; g.c_ppc = 0; g.c_pieces = 0;
  if (!(a.flags & LIBXSMM_GEMM_FLAG_BETA_0) && (a.c_type == LIBXSMM_DATATYPE_F32 || a.c_type == LIBXSMM_DATATYPE_BF16 || a.c_type == LIBXSMM_DATATYPE_F16)) {       // beta = 1: C as whole 16-byte pieces of its columns (else: element loads)
    const unsigned int ces = a.c_type == LIBXSMM_DATATYPE_F32 ? 4u : 2u;
    const unsigned long long cbits = (unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)a.ldc * ces | (unsigned long long)a.m * ces;
    const unsigned int c_img = ((((unsigned int)a.m * ces / 16u) * (unsigned int)a.n + 63u) / 64u) * 1024u;
    if (!(cbits & 15ull) && lds_bytes + c_img <= 64u * 1024u) {
      g.c_off = lds_bytes; g.c_ppc = (unsigned int)a.m * ces / 16u; g.c_pieces = g.c_ppc * (unsigned int)a.n;
      lds_bytes += c_img;
    }
  }
  if (lds_bytes > 64u * 1024u) return false;
  tpw = (tiles + 3) / 4;
  return true;
}

// waves of a workgroup: one per strip, or one per tile when the round-robin deal has fewer than four
static inline unsigned int wgp_waves(int tiles_m, int tiles_n, int deal) {
  constexpr bool four = false;
  if (four) return 4u;
  const int n = deal == 1 ? tiles_m : deal == 2 ? tiles_n : tiles_m * tiles_n;
  return (unsigned int)(n < 4 ? n : 4);
}

// strips of tiles (DEAL 1: a tile row per wave, 2: a tile column) when they are as short as the round-robin deal's longest wave
static inline int wgp_deal(int tiles_m, int tiles_n, int& tpw) {
  constexpr int forced = -1;
  // 2 x 2 tiles: TWO waves with a tile row each instead of four with a tile each -- half the waves to launch, the A fragment read once for two MFMAs, and a CU holds twelve
  // problems instead of eight (40^3: bf16 0.60 -> 0.68, i8 0.57 -> 0.65, 8-bit weights 0.50 -> 0.61, profiles/r05_wgp_pair.jsonl; LIBXSMM_HIP_WGP_PAIR=0: four waves)
  // (ONE wave with the whole 2 x 2 block, no barrier at all, sixteen problems per CU: measured and not adopted -- 40^3 0.62 against 0.65, 48^3 0.65 against 0.70, same file)
  constexpr bool pair = true;
  if (pair && tiles_m == 2 && tiles_n == 2 && forced != 0) { tpw = 2; return 1; }
  if (forced == 0 || tpw < 2) return 0;
  if ((tiles_m == 3 || tiles_m == 4) && tiles_n == tpw && forced != 2) return 1;
  if ((tiles_n == 3 || tiles_n == 4) && tiles_m == tpw) return 2;
  if ((tiles_m == 3 || tiles_m == 4) && tiles_n == tpw) return 1;
  return 0;
}

}  // namespace xamd
