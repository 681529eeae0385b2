// gemm_wgp8_kernels.hip -- 8-bit x 8-bit GEMMs, one problem per workgroup out of LDS (the form of gemm_wgp.hpp; products and corrections of gemm_8bit.hpp)
#include <algorithm>
#include "gemm_wgp.hpp"

namespace xamd {

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// 8-bit x 8-bit GEMMs (KIND 0: u8 / i8 -> i32 or scaled f32 on v_mfma_i32_32x32x32_i8; 1 / 2: BF8 / HF8 -> f32 on v_mfma_f32_32x32x16_*), the same form: one problem per
// workgroup, both PACKED operand blocks (A in VNNI-4: [k/4][m] dwords, lda == m; B flat: [n][k] bytes, ldb == k) brought in as linear copies, the products and signedness
// corrections of gemm_mfma_8bit_kernel (m8_products) fed from LDS: A as four ds_read_b32 (the k quads of my row), B as eight-byte reads of my column (k % 8 == 0 keeps
// them aligned).  A k quad beyond k is zeroed on both sides after the unsigned -> signed shift, exactly as in the wave-per-tile kernel.  C: i32 / f32, or the 8-bit float
// type of the operands (the reference's two-step rounding, bytes through an LDS image of C and out as 16-byte pieces where the columns allow it).  8-bit floats: any epilogue.
// ------------------------------------------------------------------------------------------------------------------------------------------------------------
#ifndef WGP8_W3S
#define WGP8_W3S 5
#endif
#define WGP8_WAVES(KIND, UA, UB, TPW, DEAL) ((TPW) == 4 ? 4 : (TPW) == 3 ? ((DEAL) != 0 ? WGP8_W3S : WGP_W3) : (TPW) == 2 ? 6 : 8)      // (the correction sums and k-quad masks cost the 8-bit kernels a step against the 16-bit one)
// DEAL as in gemm_wgp16_kernel; a strip is ONE call of m8_products with MT x NT = 1 x TPW (a tile row: the A quads of my row feed TPW column tiles) or TPW x 1.
template <int KIND, bool UA, bool UB, int TPW, int DEAL = 0>
__global__ __launch_bounds__(256, WGP8_WAVES(KIND, UA, UB, TPW, DEAL))
void gemm_wgp8_kernel(GemmArgs p, Wgp16Geo g) {
  constexpr bool INT = KIND == 0;
  constexpr int G = DEAL == 0 ? TPW : 1, MT = DEAL == 2 ? TPW : 1, NT = DEAL == 1 ? TPW : 1;      // G groups of MT x NT tiles; tile t = (group, mt, nt)
  constexpr auto gi = [](int t) { return DEAL == 0 ? t : 0; };
  constexpr auto mi = [](int t) { return DEAL == 2 ? t : 0; };
  constexpr auto ni = [](int t) { return DEAL == 1 ? t : 0; };
  extern __shared__ __attribute__((aligned(16))) char lds_wgp[];
  const unsigned int TS = (DEAL == 0 && TPW > 1) ? 4u : blockDim.x >> 6;
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
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  i32x16 iacc[G][INT ? MT : 1][INT ? NT : 1];
  f32x16 facc[G][INT ? 1 : MT][INT ? 1 : NT];
  int sum_a[G][MT], sum_b[G][NT];
  TileCtx tc[TPW];
  bool mine[TPW];
  static_for<TPW>([&](auto tt) {
    constexpr int t = tt.value;
    unsigned int ti, tj;
    mine[t] = wgp_tile_of<DEAL>(w, TS, (unsigned int)t, tiles_m, tiles_n, ti, tj);
    tc[t].i = (int)(32u * ti + li); tc[t].j0 = (int)(32u * tj); tc[t].h = (int)h; tc[t].ivalid = tc[t].i < p.m;
    if constexpr (INT) iacc[gi(t)][mi(t)][ni(t)] = (i32x16)0;
    sum_a[gi(t)][mi(t)] = 0; sum_b[gi(t)][ni(t)] = 0;
  });
  const bool c8 = !INT && p.c_type != LIBXSMM_DATATYPE_F32;       // C in the operands' 8-bit float type [ref: gemm ref :2511-2619] (see gemm_fp8_stream_kernel<.., C8>)
  auto init_tiles = [&]() {          // after the first block's requests (gemm_wgp.hpp)
    static_for<TPW>([&](auto tt) { constexpr int t = tt.value;
      if (!INT && mine[t]) {
        f32x16& acc = facc[gi(t)][INT ? 0 : mi(t)][INT ? 0 : ni(t)];
        if (!c8) tile_init<false, true>(acc, p, q, tc[t]);
        else tile_init_c8<false>(acc, p, q, tc[t], KIND == 2);
      } });
  };
  const unsigned int m = (unsigned int)p.m, k = (unsigned int)p.k;
  const unsigned int kquads = k >> 2, kchunks = (k + 31u) >> 5;
  auto issue = [&](unsigned long long r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    for (unsigned int x = w; x * 64u < g.a_pieces; x += TS) {
      const unsigned int P = 64u * x + lane;
      if (P < g.a_pieces) __builtin_amdgcn_global_load_lds((GM const void*)(ar + 16ull * P), (lds_vptr)(img_a + 1024u * x), 16, 0, 0);
    }
    for (unsigned int x = w; x * 64u < g.b_pieces; x += TS) {
      const unsigned int P = 64u * x + lane;
      if (P < g.b_pieces) __builtin_amdgcn_global_load_lds((GM const void*)(br + 16ull * P), (lds_vptr)(img_b + 1024u * x), 16, 0, 0);
    }
  };
  if (p.br_count) issue(0);
  init_tiles();
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    if (r != 0) { wg_barrier(); issue(r); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    static_for<G>([&](auto gg) {
      constexpr int gr = gg.value;                                 // the group's first tile is tile gr (DEAL 0) / tile 0 (a strip)
      if (mine[gr]) {
        const unsigned int i0 = (unsigned int)tc[gr].i, j0 = (unsigned int)tc[gr].j0 + li;          // + 32 mt / + 32 nt inside a strip
        for (unsigned int kc = 0; kc < kchunks; ++kc) {
          unsigned int aw[MT][4], bw[NT][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned int kq = INT ? 8u * kc + 4u * h + (unsigned int)e : 8u * kc + 4u * ((unsigned int)e >> 1) + 2u * h + ((unsigned int)e & 1u);
            const bool kok = kq < kquads;
            const unsigned int kqc = kok ? kq : 0u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const unsigned int i = i0 + 32u * (unsigned int)mt;
              unsigned int av = ((const unsigned int*)img_a)[kqc * m + i];             // k quad kq of my row
              if (INT && UA) av ^= 0x80808080u;
              aw[mt][e] = (kok && i < m) ? av : 0u;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const unsigned int j = j0 + 32u * (unsigned int)nt;
              const bool jok = j < (unsigned int)p.n;
              unsigned int bv = ((const unsigned int*)(img_b + (size_t)(jok ? j : 0u) * k))[kqc];      // dword q of my column = bytes 4 q .. (k % 8 == 0: 8-byte aligned)
              if (INT && UB) bv ^= 0x80808080u;
              bw[nt][e] = (kok && jok) ? bv : 0u;
            }
          }
          m8_products<MT, NT, KIND, UA, UB>(aw, bw, iacc[gr], facc[gr], sum_a[gr], sum_b[gr]);
        }
      }
    });
  }
  if constexpr (INT) {
    const bool c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
    const int kconst = (UA && UB) ? (int)(16384u * (unsigned int)(p.br_count * (unsigned long long)p.k)) : 0;      // (mod 2^32 like the i32 sums themselves: unsigned arithmetic, no signed overflow)
    static_for<TPW>([&](auto tt) {
      constexpr int t = tt.value;
      // (every wave executes the exchanges, also for a tile it does not own: ds_bpermute needs all lanes)
      int sb = sum_b[gi(t)][ni(t)], sa = sum_a[gi(t)][mi(t)];
      if constexpr (UA) sb += __builtin_amdgcn_ds_bpermute(4 * (int)(lane ^ 32u), sb);           // both k halves of a column: lane j + lane j + 32
      if constexpr (UB) sa += __builtin_amdgcn_ds_bpermute(4 * (int)(lane ^ 32u), sa);
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) {
        const int jl = jl_of(r2, (int)h), j = tc[t].j0 + jl;
        int v = iacc[gi(t)][mi(t)][ni(t)][r2] + kconst;
        if constexpr (UA) v += 128 * __builtin_amdgcn_ds_bpermute(4 * jl, sb);                   // the sum of column jl lives in lane jl
        if constexpr (UB) v += 128 * sa;
        if (!(mine[t] && tc[t].ivalid && j < p.n)) continue;
        GM char* cp = (GM char*)q.c + 4ll * ((long long)j * p.ldc + tc[t].i);
        if (c_f32) { float f = mul_rn((float)v, p.scf); if (!beta0) f = add_rn(f, *(GM const float*)cp); *(GM float*)cp = f; }
        else { if (!beta0) v += *(GM const int*)cp; *(GM int*)cp = v; }
      }
    });
  } else if (!c8) {
    static_for<TPW>([&](auto tt) { constexpr int t = tt.value; if (mine[t]) tile_store<false, true, false>(facc[gi(t)][mi(t)][ni(t)], p, q, tc[t]); });
  } else {
    // byte results: through an LDS image [n][m] of the whole problem (in place of the operand images) and out as 16-byte pieces of its columns when those are whole and
    // aligned in memory -- sixteen one-byte stores per tile and lane are the bound otherwise (gemm_fp8_stream_kernel: 0.27 of the roofline that way)
    // (columns that are not whole 16-byte pieces -- 72^3 -- leave byte by byte: dwords out of the image measured SLOWER than the byte stores, 0.33 against 0.38, profiles/r05_wgp_pair.jsonl)
    const bool wide = !(m & 15u) && ((((unsigned long long)(size_t)q.c) | (unsigned long long)p.ldc) & 15ull) == 0ull;      // workgroup-uniform
    if (wide) wg_barrier();                                       // everybody has read the operand images
    // (the fused activation's ballots need every lane of a wave, and `mine` is wave-uniform)
    static_for<TPW>([&](auto tt) { constexpr int t = tt.value;
      if (mine[t]) {
        f32x16& acc = facc[gi(t)][mi(t)][ni(t)];
        tile_activate<false>(acc, p, q, tc[t]);                   // fused ReLU (+ bitmask) / sigmoid (round 6)
#pragma unroll
        for (int r2 = 0; r2 < 16; r2 += 2) {
          const unsigned int two = f32x2_to_fp8_ref(acc[r2], acc[r2 + 1], KIND == 2);
          const unsigned int ja = (unsigned int)(tc[t].j0 + jl_of(r2, (int)h)), jb = (unsigned int)(tc[t].j0 + jl_of(r2 + 1, (int)h)), i = (unsigned int)tc[t].i;
          if (wide) {
            if (i < m) { if (ja < (unsigned int)p.n) ((unsigned char*)lds_wgp)[ja * m + i] = (unsigned char)two; if (jb < (unsigned int)p.n) ((unsigned char*)lds_wgp)[jb * m + i] = (unsigned char)(two >> 8); }
          } else if (i < m) {
            if (ja < (unsigned int)p.n) ((GM unsigned char*)q.c)[(long long)ja * p.ldc + i] = (unsigned char)two;
            if (jb < (unsigned int)p.n) ((GM unsigned char*)q.c)[(long long)jb * p.ldc + i] = (unsigned char)(two >> 8);
          }
        }
      } });
    if (wide) {
      wg_barrier();
      const unsigned int ppc = m >> 4, pieces = (unsigned int)p.n * ppc;
      for (unsigned int P = threadIdx.x; P < pieces; P += blockDim.x) {
        const unsigned int j = P / ppc, c16 = P - j * ppc;
        *(GM u32x4*)((GM unsigned char*)q.c + (long long)j * p.ldc + 16u * c16) = *(const u32x4*)(lds_wgp + (size_t)j * m + 16u * c16);
      }
    }
  }
}

// kind: 0 integers (ua / ub: the operand is unsigned), 1 BF8, 2 HF8 -- launch_gemm's P_M8 case; packed blocks only (lda == m, ldb == k)
int launch_gemm_wgp8(const GemmArgs& a_in, int kind, bool ua, bool ub, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
  constexpr bool off = false;
  const GemmArgs& a = a_in;
  if (off || kind < 0 || kind > 2) return 0;
  if (a.batch_inner || (a.list_a && !a.lists_aligned16) || a.br_mode == 1 || a.br_mode == 2 || a.vnni_c || (kind == 0 && (a.colbias || a.act))) return 0;      // (fused operators on the 8-bit floats: round 6)
  if ((a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B)) || !(a.flags & LIBXSMM_GEMM_FLAG_VNNI_A)) return 0;
  const bool c8 = kind != 0 && a.c_type == a.a_type;              // 8-bit floats with a result of their own type
  if (a.c_type != LIBXSMM_DATATYPE_F32 && a.c_type != LIBXSMM_DATATYPE_I32 && !c8) return 0;
  if (kind != 0 && a.c_type == LIBXSMM_DATATYPE_I32) return 0;
  if ((a.m & 3) || (a.k & 7) || a.lda != a.m || a.ldb != a.k || a.k <= 0) return 0;
  const long long abytes = (long long)a.m * a.k, bbytes = (long long)a.n * a.k;
  if ((abytes & 15) || (bbytes & 15)) return 0;
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0);
  if (bits & 15ull) return 0;
  if (!c8 && (((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c) & 3ull) != 0ull) return 0;
  const int tiles = ((a.m + 31) / 32) * ((a.n + 31) / 32);
  if (tiles < 2 || (tiles > 12 && !(tiles == 16 && a.m > 96 && a.n > 96))) return 0;
  Wgp16Geo g; g.rp = (unsigned int)a.m; g.ppr = 0; g.ppc = 0; g.bias_off = 0; g.bias_dw = 0; g.c_off = 0; g.c_ppc = 0; g.c_pieces = 0;
  g.a_pieces = (unsigned int)(abytes / 16); g.b_pieces = (unsigned int)(bbytes / 16);
  g.a_img = ((g.a_pieces + 63u) / 64u) * 1024u;
  unsigned int lds_bytes = g.a_img + ((g.b_pieces + 63u) / 64u) * 1024u;
  if (c8) lds_bytes = std::max(lds_bytes, (unsigned int)(((size_t)a.m * a.n + 15u) & ~(size_t)15u));      // the byte image of C takes the operands' place
  if (lds_bytes > 64u * 1024u) return 0;
  int tpw = (tiles + 3) / 4;
  GemmArgs b = a_in;
  b.tiles_m = (a.m + 31) / 32; b.tiles_n = (a.n + 31) / 32; b.map2d_shift = 0;
  hipStream_t st = (hipStream_t)stream;
  const int deal = wgp_deal(b.tiles_m, b.tiles_n, tpw);
  const dim3 grid(a.nbatch), block(64u * wgp_waves(b.tiles_m, b.tiles_n, deal));
  *taken = 1;
  if (kernel_name) *kernel_name = "gemm_8bit_wgp_kernel";
#define WGP8_(K_, UA_, UB_, T_, D_) hipLaunchKernelGGL((gemm_wgp8_kernel<K_, UA_, UB_, T_, D_>), grid, block, lds_bytes, st, b, g)
#define WGP8D_(K_, UA_, UB_, T_) do { if (deal == 1) WGP8_(K_, UA_, UB_, T_, 1); else if (deal == 2) WGP8_(K_, UA_, UB_, T_, 2); else WGP8_(K_, UA_, UB_, T_, 0); } while (0)
#define WGP8T_(K_, UA_, UB_) do { if (tpw == 4) WGP8_(K_, UA_, UB_, 4, 1); else if (tpw == 1) WGP8_(K_, UA_, UB_, 1, 0); else if (tpw == 2) WGP8D_(K_, UA_, UB_, 2); else WGP8D_(K_, UA_, UB_, 3); } while (0)
  if (kind == 0) { if (ua && ub) WGP8T_(0, true, true); else if (ua) WGP8T_(0, true, false); else if (ub) WGP8T_(0, false, true); else WGP8T_(0, false, false); }
  else if (kind == 1) WGP8T_(1, false, false);
  else WGP8T_(2, false, false);
#undef WGP8T_
#undef WGP8D_
#undef WGP8_
  return (int)hipPeekAtLastError();
}

}  // namespace xamd
