// gemm_wgp_f32_kernels.hip -- f32 GEMMs of several 16 x 16 tiles that are not whole 32 / 64 tiles (40^3, 72^3, 56 x 88 ...), ONE PROBLEM PER WORKGROUP with the operand
// blocks brought into LDS as 16-byte pieces (round 5; the form of gemm_wgp.hpp for f32).
//
// gemm_f32_ragged_kernel (gemm_kernels.hip) stages such a problem through the register file a dword per lane and round (72^3: 24 rounds per operand) and needs the whole
// problem -- operands, C in, C out -- in one LDS plan; what the plan does not hold (72^3 with beta = 1, deep K with wide n, ...) fell to the wave-per-tile kernel at 0.19
// of the HBM roofline.  Here, when rows of A and columns of B are whole 16-byte pieces (m % 4 == 0, k % 4 == 0, leading dimensions % 4 == 0, 16-byte aligned blocks), the
// blocks travel global -> LDS without registers, a piece per lane and request, every request issued before the first wait, K in chunks when the images exceed the budget.
// MEASURED against the register-staged kernel on the shapes BOTH take (profiles/r05_wgp_f32.jsonl): 72^3 0.36-0.41 against 0.51, 40^3 0.56 against 0.66, 56^3 0.58
// against 0.67 -- f32 problems of this size are as much matrix-pipe as memory bound (72^3: 25 tiles x 18 k steps x 32 cycles = 1.5 us per CU against 2.0 us of HBM
// time), and the lane-constant rounds with natural k order cost fewer MFMA slots -- so launch_gemm asks that kernel first and comes here with the rest
// (72^3, beta = 1: 0.19 -> 0.43-0.48).
//   A image [k][m] (compact rows), B image [n][kc] (compact columns); K in chunks of kc when both do not fit the LDS budget (a barrier and a round trip per chunk).
//   Product on v_mfma_f32_16x16x4_f32, transposed (a lane holds C(i, four consecutive j)): ceil(m / 16) x ceil(n / 16) tiles.  The waves form a WR x WC grid and each
//   owns a BLOCK of RM x RN tiles: per 16-deep k group a wave reads RM x 4 dwords of A (ds_read_b32 down a column: lanes along i) and RN 16-byte B fragments
//   (ds_read_b128: lane (j, kk) holds k = 16 c + 4 kk .. + 3 of its column) and issues RM x RN x 4 MFMAs.  MFMA s of a group multiplies the k = 16 c + 4 kk + s of the
//   four lane groups kk: every product of the sum, each once, in an order that differs from the k-ordered chain (f32 parity is a norm, tests/test_gemm_gpu.py _tol).
//   k groups beyond k are zeroed on both sides; rows >= m / columns >= n read whatever the images hold and feed results nobody stores.
// Plain epilogue (beta 0 / 1), NN, 1-D strided batches, one block per problem or STRIDE chains.  Everything else keeps gemm_f32_ragged_kernel.
// [ref: the loop being computed is src/generator_gemm_reference_impl.c:857-948 (f32)]
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include "internal.hpp"
#include "gemm_device.hpp"
#include "gemm_tile.hpp"

#pragma clang fp contract(off)

namespace xamd {

typedef float f32x4w __attribute__((ext_vector_type(4)));

struct WgpF32Geo {
  unsigned int ppr;              // 16-byte pieces per k row of A = m / 4
  unsigned int kc, kchunks;      // depth of a chunk (% 4 == 0), chunks per block
  unsigned int a_img;            // bytes of the A image (whole 1 KiB request slots)
  unsigned int wc;               // wave grid: wave w owns tile rows (w / wc) * RM .., tile columns (w % wc) * RN ..
  unsigned int tm, tn;           // 16 x 16 tiles along m, n
};

constexpr int wgp_f32_waves(int tiles) { return tiles <= 3 ? 8 : tiles <= 6 ? 7 : tiles <= 9 ? 6 : 4; }      // registers: 4 per tile + fragments

template <int RM, int RN>
__global__ __launch_bounds__(256, wgp_f32_waves(RM * RN)) void gemm_wgp_f32_kernel(GemmArgs p, WgpF32Geo g) {
  extern __shared__ __attribute__((aligned(16))) char lds_f32[];
  const unsigned int TS = blockDim.x >> 6;
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, x = lane & 15u, kk = lane >> 4;
  const unsigned int bidx = blockIdx.x;
  const BatchPtrs q = batch_ptrs(p, bidx);
  const float* const img_a = (const float*)lds_f32;
  const float* const img_b = (const float*)(lds_f32 + g.a_img);
  const unsigned int m = (unsigned int)p.m, n = (unsigned int)p.n, K = (unsigned int)p.k, lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int wr = w / g.wc, wcl = w - wr * g.wc;
  const unsigned int ti0 = wr * (unsigned int)RM, tj0 = wcl * (unsigned int)RN;           // my block of tiles
  const unsigned int i0 = 16u * ti0 + x, j0 = 16u * tj0;                                  // my row in tile row a: i0 + 16 a; tile column b starts at j0 + 16 b
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  f32x4w acc[RM][RN];
  auto issue = [&](unsigned long long step) {
    const unsigned long long r = step / g.kchunks;
    const unsigned int c = (unsigned int)(step - r * g.kchunks), k0 = c * g.kc, kcur = K - k0 < g.kc ? K - k0 : g.kc;
    gcptr ar, br; br_base(p, q, r, ar, br);
    const unsigned int a_pieces = kcur * g.ppr, ppc = kcur >> 2, b_pieces = n * ppc;
    for (unsigned int s = w; s * 64u < a_pieces; s += TS) {
      const unsigned int P = 64u * s + lane;
      if (P < a_pieces) {
        const unsigned int kr = P / g.ppr, pc = P - kr * g.ppr;
        __builtin_amdgcn_global_load_lds((GM const void*)(ar + ((unsigned long long)(k0 + kr) * lda + 4u * pc) * 4ull), (lds_vptr)(lds_f32 + 1024u * s), 16, 0, 0);
      }
    }
    for (unsigned int s = w; s * 64u < b_pieces; s += TS) {
      const unsigned int P = 64u * s + lane;
      if (P < b_pieces) {
        const unsigned int col = P / ppc, pc = P - col * ppc;
        __builtin_amdgcn_global_load_lds((GM const void*)(br + ((unsigned long long)col * ldb + k0 + 4u * pc) * 4ull), (lds_vptr)(lds_f32 + g.a_img + 1024u * s), 16, 0, 0);
      }
    }
  };
  const unsigned long long steps = p.br_count * (unsigned long long)g.kchunks;
  if (steps) issue(0);
  // start values behind the first requests: zeros or C (beta = 1)
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int b = 0; b < RN; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned int i = i0 + 16u * (unsigned int)a, j = j0 + 16u * (unsigned int)b + 4u * kk + (unsigned int)e;
        acc[a][b][e] = (!beta0 && i < m && j < n) ? ((GM const float*)q.c)[(unsigned long long)j * (unsigned int)p.ldc + i] : 0.0f;
      }
  for (unsigned long long step = 0; step < steps; ++step) {
    if (step != 0) { wg_barrier(); issue(step); }                 // (the barrier: everybody has read the previous images)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    const unsigned int c = (unsigned int)(step % g.kchunks), k0 = c * g.kc, kcur = K - k0 < g.kc ? K - k0 : g.kc;
    if (ti0 < g.tm && tj0 < g.tn) {                               // (wave-uniform: a wave of the grid without tiles)
      const unsigned int groups = (kcur + 15u) >> 4;
      for (unsigned int cg = 0; cg < groups; ++cg) {
        const unsigned int kg = 16u * cg + 4u * kk;               // my four k of the group (k % 4 == 0: whole or absent)
        const bool ok = kg < kcur;
        const unsigned int kgc = ok ? kg : 0u;
        float af[RM][4]; f32x4w bf[RN];
#pragma unroll
        for (int a = 0; a < RM; ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) { const float v = img_a[(kgc + (unsigned int)s) * m + i0 + 16u * (unsigned int)a]; af[a][s] = ok ? v : 0.0f; }
#pragma unroll
        for (int b = 0; b < RN; ++b) {
          const f32x4w v = *(const f32x4w*)(img_b + (size_t)(j0 + 16u * (unsigned int)b + x) * kcur + kgc);
          bf[b] = ok ? v : f32x4w{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int a = 0; a < RM; ++a)
#pragma unroll
            for (int b = 0; b < RN; ++b)
              if (ti0 + (unsigned int)a < g.tm && tj0 + (unsigned int)b < g.tn)          // (wave-uniform: the grid's last blocks are not full)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][s], af[a][s], acc[a][b], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int b = 0; b < RN; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned int i = i0 + 16u * (unsigned int)a, j = j0 + 16u * (unsigned int)b + 4u * kk + (unsigned int)e;
        if (i < m && j < n) ((GM float*)q.c)[(unsigned long long)j * (unsigned int)p.ldc + i] = acc[a][b][e];
      }
}

// the wave grid (wr x wc <= 4 waves) whose largest block of tiles is smallest; ties: the squarer block (fewer fragment reads per MFMA), then fewer waves
static void wgp_f32_grid(int tm, int tn, int& wr, int& wc, int& rm, int& rn) {
  static const int grids[8][2] = {{2, 2}, {1, 4}, {4, 1}, {1, 3}, {3, 1}, {1, 2}, {2, 1}, {1, 1}};
  int best = 1 << 30, best_sum = 1 << 30;
  wr = wc = rm = rn = 1;
  for (const auto& gr : grids) {
    if (gr[0] > tm || gr[1] > tn) continue;
    const int a = (tm + gr[0] - 1) / gr[0], b = (tn + gr[1] - 1) / gr[1];
    if (a > 4 || b > 4) continue;
    if (a * b < best || (a * b == best && a + b < best_sum)) { best = a * b; best_sum = a + b; wr = gr[0]; wc = gr[1]; rm = a; rn = b; }
  }
}

int launch_gemm_wgp_f32(const GemmArgs& a_in, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
  constexpr bool off = false;
  constexpr unsigned int budget = 48u * 1024u;      // LDS per workgroup (A/B switch)
  const GemmArgs& a = a_in;
  if (off || a.a_type != LIBXSMM_DATATYPE_F32 || a.b_type != LIBXSMM_DATATYPE_F32 || a.c_type != LIBXSMM_DATATYPE_F32) return 0;
  if (a.batch_inner || (a.list_a && !a.lists_aligned16) || a.br_mode == 1 || a.br_mode == 2 || a.vnni_c || a.colbias || a.act) return 0;
  if (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_A | LIBXSMM_GEMM_FLAG_VNNI_B)) return 0;
  if ((a.m & 3) || (a.k & 3) || (a.lda & 3) || (a.ldb & 3) || a.k <= 0 || a.m <= 0 || a.n <= 0 || a.m > 128 || a.n > 128) return 0;
  if (a.m <= 32 && a.n <= 32) return 0;                           // one wave's worth: the register-staged kernel streams those at 0.7
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0);
  if (bits & 15ull) return 0;
  if ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c) & 3ull) != 0ull) return 0;
  if ((long long)a.lda * a.k >= (1ll << 29) || (long long)a.ldb * a.n >= (1ll << 29)) return 0;
  WgpF32Geo g;
  g.tm = (unsigned int)(a.m + 15) / 16u; g.tn = (unsigned int)(a.n + 15) / 16u;
  int wr, wc, rm, rn;
  wgp_f32_grid((int)g.tm, (int)g.tn, wr, wc, rm, rn);
  if ((unsigned int)(wr * rm) < g.tm || (unsigned int)(wc * rn) < g.tn) return 0;
  g.wc = (unsigned int)wc; g.ppr = (unsigned int)a.m / 4u;
  // chunk depth: all of k when both images fit the budget, else the fewest even chunks (whole k quads) that do
  const unsigned int K = (unsigned int)a.k;
  auto img_bytes = [&](unsigned int kc) { return (((kc * g.ppr) + 63u) / 64u) * 1024u + ((((unsigned int)a.n * (kc / 4u)) + 63u) / 64u) * 1024u; };
  g.kc = K; g.kchunks = 1;
  if (img_bytes(K) > budget) {
    unsigned int nch = 2;
    for (; nch <= 16u; ++nch) { g.kc = (((K + nch - 1u) / nch) + 3u) & ~3u; if (img_bytes(g.kc) <= budget) break; }
    if (nch > 16u) return 0;
    g.kchunks = (K + g.kc - 1u) / g.kc;
  }
  g.a_img = (((g.kc * g.ppr) + 63u) / 64u) * 1024u;
  const unsigned int lds_bytes = img_bytes(g.kc);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.nbatch), block(64u * (unsigned int)(wr * wc));
  *taken = 1;
  if (kernel_name) *kernel_name = "gemm_f32_wgp_kernel";
#define WF_(RM_, RN_) hipLaunchKernelGGL((gemm_wgp_f32_kernel<RM_, RN_>), grid, block, lds_bytes, st, a, g)
#define WFR_(RM_) do { if (rn == 1) WF_(RM_, 1); else if (rn == 2) WF_(RM_, 2); else if (rn == 3) WF_(RM_, 3); else WF_(RM_, 4); } while (0)
  if (rm == 1) WFR_(1); else if (rm == 2) WFR_(2); else if (rm == 3) WFR_(3); else WFR_(4);
#undef WFR_
#undef WF_
  return (int)hipGetLastError();
}

}  // namespace xamd
