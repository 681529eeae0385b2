// gemm_wgp16_kernels.hip -- ragged 16-bit (bf16 / f16) GEMMs, ONE PROBLEM PER WORKGROUP with the whole problem staged in LDS (round 5).
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
// m % 4 == 0, k % 8 == 0, lda % 4 == 0, ldb % 8 == 0, 16-byte aligned blocks, 2 <= tiles <= 12, LDS image <= 64 KiB.  Everything else keeps the wave-per-tile kernel.
// [ref: the loop being computed is src/generator_gemm_reference_impl.c:2127-2170 (bf16 -> f32), :2367-2419 (bf16 -> bf16), :2025-2124 (f16)]
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
};

// AK = -1: 16-bit A (VNNI-2 dwords).  AK = 0..4: 8-bit WEIGHTS x bf16 activations (KIND of gemm_w8_bf16_kernel: 0 / 1 BF8 / HF8 in VNNI-2 byte pairs, 2 / 3 flat, 4 int8 with
// one f32 scale per row) -- the A block is a BYTE image ([k/2][m][2] or [k][m], lda == m) that comes in as a linear copy (whole 16-byte pieces of the packed block) and is
// turned into the bf16 pairs the reference multiplies with on the way out of LDS (w8_pair_to_bf16: exact for the 8-bit floats, one rounding for the scaled int8).
// Register bounds = waves per SIMD the compiler must leave room for (__launch_bounds__' second argument).  LDS never limits these kernels (6-20 KiB per workgroup of
// 160 KiB); resident workgroups are what hides a problem's single round trip, so every form is bounded to the most waves that compile WITHOUT scratch:
// one tile per wave 8 (<= 64 registers), two 6 (<= 80), three 5 (<= 96).  Measured: profiles/r05_wgp_bound5.jsonl (three tiles), r05_wgp_waves.jsonl (one / two).
#ifndef WGP_W1
#define WGP_W1 8
#endif
#ifndef WGP_W2
#define WGP_W2 6
#endif
#ifndef WGP_W3
#define WGP_W3 5
#endif
#define WGP_WAVES(T) ((T) == 3 ? WGP_W3 : (T) == 2 ? WGP_W2 : WGP_W1)
template <bool F16, int TPW, int AK = -1>
__global__ __launch_bounds__(256, WGP_WAVES(TPW)) void gemm_wgp16_kernel(GemmArgs p, Wgp16Geo g) {      // (three tiles per wave: 120 registers = four waves per SIMD without the bound)
  extern __shared__ __attribute__((aligned(16))) char lds_wgp[];
  constexpr unsigned int TS = 4u;                                 // the four waves of the workgroup share the problem
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int bidx = blockIdx.x;
  const BatchPtrs q = batch_ptrs(p, bidx);
  char* const img_a = lds_wgp;
  char* const img_b = img_a + g.a_img;
  const unsigned int ntiles = (unsigned int)(p.tiles_m * p.tiles_n);
  f32x16 acc[TPW];
  TileCtx tc[TPW];
  static_for<TPW>([&](auto tt) {
    constexpr int t = tt.value;
    const unsigned int id = w + TS * (unsigned int)t;
    const unsigned int tj = id / (unsigned int)p.tiles_m, ti = id - tj * (unsigned int)p.tiles_m;
    tc[t].i = (int)(32u * ti + li); tc[t].j0 = (int)(32u * tj); tc[t].h = (int)h; tc[t].ivalid = tc[t].i < p.m;
    if (id < ntiles) {
      if (F16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
      } else tile_init<false, false>(acc[t], p, q, tc[t]);
    }
  });
  float scf[TPW];
  static_for<TPW>([&](auto tt) { constexpr int t = tt.value;
    scf[t] = (AK == 4 && tc[t].ivalid) ? ((GM const float*)(p.a_scf + (long long)bidx * p.bs_scf))[tc[t].i] : 1.0f; });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int kchunks = ((unsigned int)p.k + 31u) >> 5, kgroups = (unsigned int)p.k >> 3;      // 8-deep k groups (k % 8 == 0)
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    if (r != 0) wg_barrier();                                   // everybody has read the previous block's images
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    // ---- multiply out of LDS: my tiles, K in chunks of 32 (two MFMA steps of 16)
    static_for<TPW>([&](auto tt) {
      constexpr int t = tt.value;
      const unsigned int id = w + TS * (unsigned int)t;
      if (id < ntiles) {
        const unsigned int tj = id / (unsigned int)p.tiles_m, ti = id - tj * (unsigned int)p.tiles_m;
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
  static_for<TPW>([&](auto tt) {
    constexpr int t = tt.value;
    if (w + TS * (unsigned int)t < ntiles) tile_store<false, false, false>(acc[t], p, q, tc[t]);
  });
}

// rows / columns beyond m / n of a tile read LDS beyond their operand's rows (another k pair's row, the other image, or nothing): they feed results nobody stores,
// and an LDS read beyond the allocation returns zero by definition -- no fault is possible on that side.
static bool wgp16_shape_ok(const GemmArgs& a, Wgp16Geo& g, unsigned int& lds_bytes, int& tpw, int ak = -1) {
  static const bool off = []() { const char* e = getenv("LIBXSMM_HIP_WGP16"); return e && e[0] == '0'; }();
  if (off) return false;
  if (a.batch_inner || a.list_a || a.br_mode == 1 || a.br_mode == 2 || a.vnni_c) return false;          // 1-D strided batches, plain / STRIDE batch-reduce
  if (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B)) return false;
  if (ak < 0 && !(a.flags & LIBXSMM_GEMM_FLAG_VNNI_A)) return false;
  if ((a.m & 3) || (a.k & 7) || (a.lda & 3) || (a.ldb & 7) || a.k <= 0) return false;
  if (ak >= 0 && (a.lda != a.m || (((long long)a.m * a.k) & 15))) return false;       // 8-bit weights: the packed byte image of the block, whole 16-byte pieces of it
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0);
  if (bits & 15ull) return false;
  const int tiles = ((a.m + 31) / 32) * ((a.n + 31) / 32);
  if (tiles < 2 || tiles > 12) return false;
  g.ppr = (unsigned int)a.m / 4u; g.rp = (unsigned int)a.m;
  g.ppc = (unsigned int)a.k / 8u;
  g.a_pieces = ak >= 0 ? (unsigned int)(((long long)a.m * a.k) / 16) : ((unsigned int)a.k / 2u) * g.ppr; g.b_pieces = (unsigned int)a.n * g.ppc;
  g.a_img = ((g.a_pieces + 63u) / 64u) * 1024u;
  lds_bytes = g.a_img + ((g.b_pieces + 63u) / 64u) * 1024u;
  if (lds_bytes > 64u * 1024u) return false;
  tpw = (tiles + 3) / 4;
  return true;
}

int launch_gemm_wgp16(const GemmArgs& a_in, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
  Wgp16Geo g; unsigned int lds_bytes = 0; int tpw = 0;
  const bool f16 = a_in.a_type == LIBXSMM_DATATYPE_F16;
  if (f16 && (!(a_in.flags & LIBXSMM_GEMM_FLAG_BETA_0) || a_in.colbias || a_in.act)) return 0;       // halves: beta * C is added AFTER the sum [ref: gemm ref :2025-2124] -- the wave-per-tile kernel's own epilogue
  if (!wgp16_shape_ok(a_in, g, lds_bytes, tpw)) return 0;
  GemmArgs a = a_in;
  a.tiles_m = (a.m + 31) / 32; a.tiles_n = (a.n + 31) / 32; a.map2d_shift = 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(256);
  *taken = 1;
  if (kernel_name) *kernel_name = f16 ? "gemm_f16_wgp_kernel" : "gemm_bf16_wgp_kernel";
  const dim3 grid(a.nbatch);
#define WGP_(F_, T_) hipLaunchKernelGGL((gemm_wgp16_kernel<F_, T_>), grid, block, lds_bytes, st, a, g)
  if (f16) { if (tpw == 1) WGP_(true, 1); else if (tpw == 2) WGP_(true, 2); else WGP_(true, 3); }
  else { if (tpw == 1) WGP_(false, 1); else if (tpw == 2) WGP_(false, 2); else WGP_(false, 3); }
#undef WGP_
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// 8-bit x 8-bit GEMMs (KIND 0: u8 / i8 -> i32 or scaled f32 on v_mfma_i32_32x32x32_i8; 1 / 2: BF8 / HF8 -> f32 on v_mfma_f32_32x32x16_*), the same form: one problem per
// workgroup, both PACKED operand blocks (A in VNNI-4: [k/4][m] dwords, lda == m; B flat: [n][k] bytes, ldb == k) brought in as linear copies, the products and signedness
// corrections of gemm_mfma_8bit_kernel (m8_products) fed from LDS: A as four ds_read_b32 (the k quads of my row), B as eight-byte reads of my column (k % 8 == 0 keeps
// them aligned).  A k quad beyond k is zeroed on both sides after the unsigned -> signed shift, exactly as in the wave-per-tile kernel.  C: i32 / f32 (the 8-bit float
// result types stay with the wave-per-tile kernel).
// ------------------------------------------------------------------------------------------------------------------------------------------------------------
template <int KIND, bool UA, bool UB, int TPW>
__global__ __launch_bounds__(256, WGP_WAVES(TPW)) void gemm_wgp8_kernel(GemmArgs p, Wgp16Geo g) {      // (three tiles per wave with an unsigned operand: 132 registers without the bound = three waves per SIMD)
  constexpr bool INT = KIND == 0;
  extern __shared__ __attribute__((aligned(16))) char lds_wgp[];
  constexpr unsigned int TS = 4u;
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int bidx = blockIdx.x;
  const BatchPtrs q = batch_ptrs(p, bidx);
  char* const img_a = lds_wgp;
  char* const img_b = img_a + g.a_img;
  const unsigned int ntiles = (unsigned int)(p.tiles_m * p.tiles_n);
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  i32x16 iacc[TPW][INT ? 1 : 1][1];
  f32x16 facc[TPW][1][1];
  int sum_a[TPW][1], sum_b[TPW][1];
  TileCtx tc[TPW];
  static_for<TPW>([&](auto tt) {
    constexpr int t = tt.value;
    const unsigned int id = w + TS * (unsigned int)t;
    const unsigned int tj = id / (unsigned int)p.tiles_m, ti = id - tj * (unsigned int)p.tiles_m;
    tc[t].i = (int)(32u * ti + li); tc[t].j0 = (int)(32u * tj); tc[t].h = (int)h; tc[t].ivalid = tc[t].i < p.m;
    iacc[t][0][0] = (i32x16)0; sum_a[t][0] = 0; sum_b[t][0] = 0;
    if (!INT && id < ntiles) tile_init<false, true>(facc[t][0][0], p, q, tc[t]);
  });
  const unsigned int m = (unsigned int)p.m, k = (unsigned int)p.k;
  const unsigned int kquads = k >> 2, kchunks = (k + 31u) >> 5;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    if (r != 0) wg_barrier();
    for (unsigned int x = w; x * 64u < g.a_pieces; x += TS) {
      const unsigned int P = 64u * x + lane;
      if (P < g.a_pieces) __builtin_amdgcn_global_load_lds((GM const void*)(ar + 16ull * P), (lds_vptr)(img_a + 1024u * x), 16, 0, 0);
    }
    for (unsigned int x = w; x * 64u < g.b_pieces; x += TS) {
      const unsigned int P = 64u * x + lane;
      if (P < g.b_pieces) __builtin_amdgcn_global_load_lds((GM const void*)(br + 16ull * P), (lds_vptr)(img_b + 1024u * x), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    static_for<TPW>([&](auto tt) {
      constexpr int t = tt.value;
      const unsigned int id = w + TS * (unsigned int)t;
      if (id < ntiles) {
        const unsigned int tj = id / (unsigned int)p.tiles_m, ti = id - tj * (unsigned int)p.tiles_m;
        const unsigned int* const arow = (const unsigned int*)img_a + 32u * ti + li;                 // + kq * m
        const unsigned int j = 32u * tj + li;
        const bool iok = tc[t].ivalid, jok = j < (unsigned int)p.n;
        const unsigned int* const bcol = (const unsigned int*)(img_b + (size_t)(jok ? j : 0u) * k);      // dword q of my column = bytes 4 q .. (k % 8 == 0: 8-byte aligned)
        for (unsigned int kc = 0; kc < kchunks; ++kc) {
          unsigned int aw[1][4], bw[1][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned int kq = INT ? 8u * kc + 4u * h + (unsigned int)e : 8u * kc + 4u * ((unsigned int)e >> 1) + 2u * h + ((unsigned int)e & 1u);
            const bool kok = kq < kquads;
            const unsigned int kqc = kok ? kq : 0u;
            unsigned int av = arow[kqc * m], bv = bcol[kqc];
            if (INT && UA) av ^= 0x80808080u;
            if (INT && UB) bv ^= 0x80808080u;
            aw[0][e] = (kok && iok) ? av : 0u;
            bw[0][e] = (kok && jok) ? bv : 0u;
          }
          m8_products<1, 1, KIND, UA, UB>(aw, bw, iacc[t], facc[t], sum_a[t], sum_b[t]);
        }
      }
    });
  }
  if constexpr (INT) {
    const bool c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
    const int kconst = (UA && UB) ? 16384 * (int)(p.br_count * (unsigned long long)p.k) : 0;
    static_for<TPW>([&](auto tt) {
      constexpr int t = tt.value;
      // (every wave executes the exchanges, also for a tile it does not own: ds_bpermute needs all lanes)
      int sb = sum_b[t][0], sa = sum_a[t][0];
      if constexpr (UA) sb += __builtin_amdgcn_ds_bpermute(4 * (int)(lane ^ 32u), sb);           // both k halves of a column: lane j + lane j + 32
      if constexpr (UB) sa += __builtin_amdgcn_ds_bpermute(4 * (int)(lane ^ 32u), sa);
      const bool mine = w + TS * (unsigned int)t < ntiles;
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) {
        const int jl = jl_of(r2, (int)h), j = tc[t].j0 + jl;
        int v = iacc[t][0][0][r2] + kconst;
        if constexpr (UA) v += 128 * __builtin_amdgcn_ds_bpermute(4 * jl, sb);                   // the sum of column jl lives in lane jl
        if constexpr (UB) v += 128 * sa;
        if (!(mine && tc[t].ivalid && j < p.n)) continue;
        GM char* cp = (GM char*)q.c + 4ll * ((long long)j * p.ldc + tc[t].i);
        if (c_f32) { float f = mul_rn((float)v, p.scf); if (!beta0) f = add_rn(f, *(GM const float*)cp); *(GM float*)cp = f; }
        else { if (!beta0) v += *(GM const int*)cp; *(GM int*)cp = v; }
      }
    });
  } else {
    static_for<TPW>([&](auto tt) { constexpr int t = tt.value; if (w + TS * (unsigned int)t < ntiles) tile_store<false, true, false>(facc[t][0][0], p, q, tc[t]); });
  }
}

// kind: 0 integers (ua / ub: the operand is unsigned), 1 BF8, 2 HF8 -- launch_gemm's P_M8 case; packed blocks only (lda == m, ldb == k)
int launch_gemm_wgp8(const GemmArgs& a_in, int kind, bool ua, bool ub, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
  static const bool off = []() { const char* e = getenv("LIBXSMM_HIP_WGP16"); return e && e[0] == '0'; }();
  const GemmArgs& a = a_in;
  if (off || kind < 0 || kind > 2) return 0;
  if (a.batch_inner || a.list_a || a.br_mode == 1 || a.br_mode == 2 || a.vnni_c || a.colbias || a.act) return 0;
  if ((a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B)) || !(a.flags & LIBXSMM_GEMM_FLAG_VNNI_A)) return 0;
  if (a.c_type != LIBXSMM_DATATYPE_F32 && a.c_type != LIBXSMM_DATATYPE_I32) return 0;
  if ((a.m & 3) || (a.k & 7) || a.lda != a.m || a.ldb != a.k || a.k <= 0) return 0;
  const long long abytes = (long long)a.m * a.k, bbytes = (long long)a.n * a.k;
  if ((abytes & 15) || (bbytes & 15)) return 0;
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0);
  if (bits & 15ull) return 0;
  if ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c) & 3ull) != 0ull) return 0;
  const int tiles = ((a.m + 31) / 32) * ((a.n + 31) / 32);
  if (tiles < 2 || tiles > 12) return 0;
  Wgp16Geo g; g.rp = (unsigned int)a.m; g.ppr = 0; g.ppc = 0;
  g.a_pieces = (unsigned int)(abytes / 16); g.b_pieces = (unsigned int)(bbytes / 16);
  g.a_img = ((g.a_pieces + 63u) / 64u) * 1024u;
  const unsigned int lds_bytes = g.a_img + ((g.b_pieces + 63u) / 64u) * 1024u;
  if (lds_bytes > 64u * 1024u) return 0;
  const int tpw = (tiles + 3) / 4;
  GemmArgs b = a_in;
  b.tiles_m = (a.m + 31) / 32; b.tiles_n = (a.n + 31) / 32; b.map2d_shift = 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.nbatch), block(256);
  *taken = 1;
  if (kernel_name) *kernel_name = "gemm_8bit_wgp_kernel";
#define WGP8_(K_, UA_, UB_, T_) hipLaunchKernelGGL((gemm_wgp8_kernel<K_, UA_, UB_, T_>), grid, block, lds_bytes, st, b, g)
#define WGP8T_(K_, UA_, UB_) do { if (tpw == 1) WGP8_(K_, UA_, UB_, 1); else if (tpw == 2) WGP8_(K_, UA_, UB_, 2); else WGP8_(K_, UA_, UB_, 3); } while (0)
  if (kind == 0) { if (ua && ub) WGP8T_(0, true, true); else if (ua) WGP8T_(0, true, false); else if (ub) WGP8T_(0, false, true); else WGP8T_(0, false, false); }
  else if (kind == 1) WGP8T_(1, false, false);
  else WGP8T_(2, false, false);
#undef WGP8T_
#undef WGP8_
  return (int)hipPeekAtLastError();
}

// 8-bit weights x bf16 activations on ragged / several-tile shapes (kind as in launch_gemm's P_W8 case); plain strided batches, one block per problem or STRIDE chains
int launch_gemm_wgp16_w8(const GemmArgs& a_in, int kind, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
  Wgp16Geo g; unsigned int lds_bytes = 0; int tpw = 0;
  if (kind < 0 || kind > 4 || a_in.b_type != LIBXSMM_DATATYPE_BF16) return 0;
  if (!wgp16_shape_ok(a_in, g, lds_bytes, tpw, kind)) return 0;
  if (kind == 4 && (!a_in.a_scf || (a_in.bs_scf & 3))) return 0;
  GemmArgs a = a_in;
  a.tiles_m = (a.m + 31) / 32; a.tiles_n = (a.n + 31) / 32; a.map2d_shift = 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.nbatch), block(256);
  *taken = 1;
  if (kernel_name) *kernel_name = "gemm_w8_wgp_kernel";
#define WGPW_(K_, T_) hipLaunchKernelGGL((gemm_wgp16_kernel<false, T_, K_>), grid, block, lds_bytes, st, a, g)
#define WGPWT_(K_) do { if (tpw == 1) WGPW_(K_, 1); else if (tpw == 2) WGPW_(K_, 2); else WGPW_(K_, 3); } while (0)
  switch (kind) { case 0: WGPWT_(0); break; case 1: WGPWT_(1); break; case 2: WGPWT_(2); break; case 3: WGPWT_(3); break; default: WGPWT_(4); break; }
#undef WGPWT_
#undef WGPW_
  return (int)hipGetLastError();
}

}  // namespace xamd
