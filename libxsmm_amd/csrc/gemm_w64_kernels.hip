// gemm_w64_kernels.hip -- 16-bit (bf16 / f16) 64 x 64 x (64 j) problems (any batch-reduce chain of them), one problem per WAVE, every byte moved in whole 128-byte lines (round 6; BASELINE config #5).
//
// gemm_bf16_wg64_kernel (gemm_kernels.hip) gives a problem to four waves: each fetches K chunks of 32 (half of every 128-byte column of B per request), the four meet at
// a barrier per chunk, and every wave stores 64-byte halves of C's columns -- 0.73 of the HBM roofline with traffic 1.02 x algorithmic: waiting, not bytes.  The 32^3
// kernel reaches 0.82-0.84 on the same bytes per wave; what it has and the 64^3 kernel lacked is (a) requests and stores that cover whole cache lines (which is also what
// lets the non-temporal policy help instead of hurt) and (b) waves that never wait for each other.  Here:
//   * a wave owns the problem: A (VNNI-2, [32 k-pairs][64 rows] dwords = 8 KiB) and B ([64 columns][64 k] = 8 KiB) arrive by sixteen LDS-DMA requests of 1 KiB, each
//     covering whole lines (A: four k-pair rows of 256 bytes; B: eight columns of 128 bytes), all in flight before the first wait; the image is wave-private -- no barrier;
//   * the 16-byte slots are XOR-swizzled on the SOURCE side (the DMA destination is lane-linear) so that the fragment reads are conflict free:
//       A: rows 32 .. 63 swapped with rows 0 .. 31 for k-pairs with bit 2 set (the two halves of a wave read k-pairs 4 apart: they land on opposite bank halves);
//       B: piece p of column f in slot p ^ ((f >> 1) & 7) (a ds_read_b128 is served sixteen lanes at a time; their columns alias mod 2);
//   * sixteen 32 x 32 x 16 MFMAs, k ascending (the accumulation order of every other 16-bit kernel: bitwise the same sums);
//   * 16-bit C leaves through the same LDS bytes: the packed pairs are written as the [64 columns][128 bytes] image of C (slot swizzle: bit 6 ^ column bit 2), read back
//     as 16 bytes per lane and stored as eight 1 KiB requests of whole lines; f32 C the same way (sixteen requests; from registers it is sixty-four of two lines each).
// 16 KiB of LDS per wave: ten waves per CU, 160 KiB of operand bytes in flight per CU.
// Any batch form of one problem per batch element (strided, pointer lists the library built), beta 0 / 1, column bias, ReLU (+ bitmask), sigmoid (tile_init / act_fixed).
#include "gemm_tile.hpp"

namespace xamd {

template <int ACT, bool F16>
__device__ __forceinline__ void w64_store_c16(f32x16 (&acc)[2][2], const GemmArgs& p, const BatchPtrs& q, unsigned int* img, unsigned int lane, unsigned int li, unsigned int h) {
  const bool odd = (lane & 1u) != 0;
  const unsigned int sel = odd ? 0x03020706u : 0x05040100u;
  static_for<4>([&](auto tc_) {
    constexpr int mt = tc_.value & 1, nt = tc_.value >> 1;
    float y[16]; unsigned int wp[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = act_fixed<ACT>(acc[mt][nt][r]);
    if (F16) {
#pragma unroll
      for (int g = 0; g < 8; ++g) wp[g] = cvt_pk_f16(y[2 * g], y[2 * g + 1]);
    } else bf16_pk_exact_n<8>(y, wp);
    // dword (rows i & ~1, i | 1) of column 32 nt + jr + 4 h + odd: byte col * 128 + ((2 (i & ~1)) ^ 64 h)   [(col >> 2) & 1 == h for every jr below]
    const unsigned int base = (32u * nt + 4u * h + (odd ? 1u : 0u)) * 128u + (((32u * mt + (li & ~1u)) * 2u) ^ (64u * h));
    static_for<8>([&](auto gc) {
      constexpr int g = gc.value, r0 = 2 * g, jr = (r0 & 3) + 8 * (r0 >> 2);
      const unsigned int w = wp[g];
      const unsigned int n = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
      *(unsigned int*)((char*)img + base + 128u * jr) = (unsigned int)__builtin_amdgcn_perm(n, w, sel);
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // read back: request x covers columns 8 x .. 8 x + 7; lane = (column lane / 8, rows 8 (lane % 8) .. + 7)
  const unsigned int ldc2 = (unsigned int)p.ldc * 2u;
  const unsigned int lsrc = (lane >> 3) * 128u + (((lane & 7u) * 16u) ^ (64u * h));
  GM char* cdst = (GM char*)q.c + (unsigned long long)(lane >> 3) * ldc2 + (lane & 7u) * 16u;
  u32x4 v[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) v[x] = *(const u32x4*)((const char*)img + 1024u * x + lsrc);
#pragma unroll
  for (int x = 0; x < 8; ++x) st_stream((GM u32x4*)(cdst + (unsigned long long)(8u * x) * ldc2), v[x]);
}

// f32 C the same way: the image is [64 columns][256 bytes] (all 16 KiB), bit 7 of the byte offset ^ column bit 2; sixteen requests of 1 KiB (four whole columns each)
template <int ACT>
__device__ __forceinline__ void w64_store_c32(f32x16 (&acc)[2][2], const GemmArgs& p, const BatchPtrs& q, unsigned int* img, unsigned int lane, unsigned int li, unsigned int h) {
  static_for<4>([&](auto tc_) {
    constexpr int mt = tc_.value & 1, nt = tc_.value >> 1;
    const unsigned int base = (32u * nt + 4u * h) * 256u + (((32u * mt + li) * 4u) ^ (128u * h));          // [(col >> 2) & 1 == h for every jr below]
    static_for<16>([&](auto rc) {
      constexpr int r = rc.value, jr = (r & 3) + 8 * (r >> 2);
      *(float*)((char*)img + base + 256u * jr) = act_fixed<ACT>(acc[mt][nt][r]);
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned int ldc4 = (unsigned int)p.ldc * 4u;
  GM char* cdst = (GM char*)q.c + (unsigned long long)(lane >> 4) * ldc4 + (lane & 15u) * 16u;
#pragma unroll
  for (int half = 0; half < 2; ++half) {          // eight requests at a time: sixteen 16-byte values would be 64 more registers
    u32x4 v[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) v[x] = *(const u32x4*)((const char*)img + 1024u * (8 * half + x) + (lane >> 4) * 256u + (((lane & 15u) * 16u) ^ (128u * (unsigned int)(x & 1))));
#pragma unroll
    for (int x = 0; x < 8; ++x) st_stream((GM u32x4*)(cdst + (unsigned long long)(4u * (8 * half + x)) * ldc4), v[x]);
  }
}

template <bool F16, int AUX, int WPB>      // WPB: waves (independent problems) per workgroup
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_16bit_w64_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned int img_all[WPB][4096];       // per wave: A as dwords [32 k-pairs][64 rows]; then B as bytes [64 columns][128]
  const unsigned int w = WPB == 1 ? 0u : (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int bidx = blockIdx.x * WPB + w;
  if (WPB > 1 && bidx >= p.nbatch) return;
  unsigned int* img = img_all[w];
  const BatchPtrs q = batch_ptrs(p, bidx);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int kchunks = (unsigned int)p.k >> 6;
  const unsigned long long total = p.br_count * kchunks;            // 64-deep chunks of the whole batch-reduce chain
  // The column bias is asked for BEFORE the requests: the compiler waits for everything outstanding at the first use of an ordinary load's result while an LDS-DMA is in
  // flight, so a bias loaded behind the requests (tile_init, once per tile) costs each tile a memory round trip of its own behind the operands' (fused 64^3: 0.67 -> see DESIGN).
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  float bias[2] = {0.0f, 0.0f};
  if (p.colbias) {
    if (p.c_type == LIBXSMM_DATATYPE_F32) { bias[0] = ((GM const float*)q.d)[li]; bias[1] = ((GM const float*)q.d)[32u + li]; }
    else if (p.c_type == LIBXSMM_DATATYPE_F16) { bias[0] = (float)((GM const _Float16*)q.d)[li]; bias[1] = (float)((GM const _Float16*)q.d)[32u + li]; }
    else { bias[0] = bf16_to_f32(((GM const unsigned short*)q.d)[li]); bias[1] = bf16_to_f32(((GM const unsigned short*)q.d)[32u + li]); }
  }
  // A request x: k-pair 4 x + lane / 16, LDS rows 4 (lane % 16) .. + 3 <- source rows swapped by halves when the k-pair has bit 2 set (x odd)
  // B request x: column 8 x + lane / 8, LDS slot lane % 8 <- source piece slot ^ ((column >> 1) & 7) = slot ^ (4 (x & 1) + lane / 16)
  unsigned int offA[2], offB[2];
#pragma unroll
  for (int xo = 0; xo < 2; ++xo) {
    offA[xo] = (lane >> 4) * lda * 4u + (((lane & 15u) ^ (8u * xo)) * 16u);
    offB[xo] = (lane >> 3) * ldb * 2u + (((lane & 7u) ^ (4u * xo + (lane >> 4))) * 16u);
  }
  // chunk t of the chain = k-chunk t % kchunks of batch-reduce block t / kchunks: sixteen requests, all in flight together
  auto issue = [&](unsigned long long t) {
    const unsigned int r = (unsigned int)(t / kchunks), kc = (unsigned int)t - r * kchunks;
    gcptr ar, br; br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + (unsigned long long)kc * 128ull * lda), rb = wave_rsrc(br + 128ull * kc);
#pragma unroll
    for (int x = 0; x < 8; ++x) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)((char*)img + 1024 * x), 16, (int)offA[x & 1], (int)(4u * x * lda * 4u), 0, AUX);
#pragma unroll
    for (int x = 0; x < 8; ++x) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)((char*)img + 8192 + 1024 * x), 16, (int)offB[x & 1], (int)(8u * x * ldb * 2u), 0, AUX);
  };
  issue(0);
  f32x16 acc[2][2];
  TileCtx tc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { tc[mt][nt].i = (int)(32u * mt + li); tc[mt][nt].j0 = 32 * nt; tc[mt][nt].h = (int)h; tc[mt][nt].ivalid = true; }
  if (beta0) {          // the streaming case: nothing of C to read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float v = F16 ? (float)(_Float16)bias[mt] : bias[mt];          // (halves: the start value is rounded to f16, tile_init F16S)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = v;
    }
  } else {              // beta = 1: tile_init's hoisted form reads C (its bias, if any, was added to p.colbias' account above: NOBIAS)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        tile_init<true, false, true, true, false>(acc[mt][nt], p, q, tc[mt][nt]);
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = p.colbias ? bias[mt] + acc[mt][nt][r] : acc[mt][nt][r]; acc[mt][nt][r] = F16 ? (float)(_Float16)v : v; }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // a chain (several batch-reduce blocks, k > 64) takes its chunks one after the other through the one image: the wave pays a round trip per chunk, the other nine
  // waves of the CU keep the memory system busy meanwhile (the image is refilled as soon as the chunk's fragments have been read)
  for (unsigned long long t = 0; t < total; ++t) {
    if (t > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 af[2], bfr[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) af[mt][e] = img[(8u * s + 4u * h + e) * 64u + ((32u * mt + li) ^ (32u * h))];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const unsigned int f = 32u * nt + li;
        bfr[nt] = *(const u32x4*)((const char*)img + 8192 + f * 128u + (((2u * s + h) ^ ((f >> 1) & 7u)) * 16u));
      }
      static_for<4>([&](auto idx) { constexpr int mt = idx.value & 1, nt = idx.value >> 1; acc[mt][nt] = mfma_16bit<F16>(bfr[nt], af[mt], acc[mt][nt]); });
    }
    if (t + 1 < total) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue(t + 1); }
  }
  const bool c16 = p.c_type != LIBXSMM_DATATYPE_F32;
  const bool lines = (((unsigned int)(size_t)q.c | ((unsigned int)p.ldc * (c16 ? 2u : 4u))) & 15u) == 0u;       // wave-uniform
  if (!lines) {
    static_for<4>([&](auto idx) { constexpr int mt = idx.value & 1, nt = idx.value >> 1; tile_store<true, false>(acc[mt][nt], p, q, tc[mt][nt]); });
    return;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the fragment reads have left the image before C's image overwrites it
  if (p.act == 2) static_for<4>([&](auto idx) { constexpr int mt = idx.value & 1, nt = idx.value >> 1; tile_relu_mask<true>(acc[mt][nt], p, q, tc[mt][nt]); });
  if (!c16) {
    if (p.act == 0) w64_store_c32<0>(acc, p, q, img, lane, li, h);
    else if (p.act == 3) w64_store_c32<3>(acc, p, q, img, lane, li, h);
    else w64_store_c32<1>(acc, p, q, img, lane, li, h);
    return;
  }
  if (p.act == 0) w64_store_c16<0, F16>(acc, p, q, img, lane, li, h);
  else if (p.act == 3) w64_store_c16<3, F16>(acc, p, q, img, lane, li, h);
  else w64_store_c16<1, F16>(acc, p, q, img, lane, li, h);
}

// *taken = 0: the caller's other kernels serve
int launch_gemm_16bit_w64(const GemmArgs& a_in, bool nt, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
#if defined(XAMD_W64_NO_CHAINS)      // A/B build only (tools/build_variant.sh): chains and k > 64 stay with gemm_bf16_wg64_kernel, as before the kernel learnt them
  if (a_in.k != 64 || a_in.br_count != 1) return 0;
#endif
  const GemmArgs& a = a_in;
  const bool f16 = a.a_type == LIBXSMM_DATATYPE_F16;
  if (a.m != 64 || a.n != 64 || a.k < 64 || (a.k & 63) != 0 || a.br_count < 1 || a.br_count * (unsigned long long)(a.k >> 6) >= (1ull << 31) || a.batch_inner || a.vnni_c || (a.list_a && !a.lists_aligned16) || (a.br_mode != 0 && a.br_mode != 3)) return 0;
  if (a.c_type != LIBXSMM_DATATYPE_F32 && a.c_type != (f16 ? LIBXSMM_DATATYPE_F16 : LIBXSMM_DATATYPE_BF16)) return 0;
  const unsigned long long brs = a.br_mode == 3 ? (unsigned long long)(a.br_stride_a | a.br_stride_b) : 0ull;
  const unsigned long long bits = brs | (a.list_a ? 0ull : ((unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b)) |
    (unsigned long long)((long long)a.lda * 4) | (unsigned long long)((long long)a.ldb * 2);
  if ((bits & 15ull) != 0 || a.lda >= (1 << 22) || a.ldb >= (1 << 22)) return 0;
  *taken = 1;
  if (kernel_name) *kernel_name = f16 ? "gemm_f16_w64_kernel" : "gemm_bf16_w64_kernel";
  GemmArgs b = a; b.map2d_shift = 0;
  hipStream_t st = (hipStream_t)stream;
  const bool use_nt = nt;          // (2^17 problems: nt loads 0.75-0.78 against 0.73-0.74 cacheable, profiles/r06_w64.jsonl)
  // two waves (two independent problems) per workgroup: measured 0.784 / 0.787 (fused / plain, nt) against 0.769 / 0.753 with one and 0.747 / 0.765 with four
  // (2^17 problems, profiles/r06_w64.jsonl); nothing is shared, the pairing only halves the number of workgroups the dispatcher places
#define W64_(F_, A_) hipLaunchKernelGGL((gemm_16bit_w64_kernel<F_, A_, 2>), dim3((a.nbatch + 1u) / 2u), dim3(128), 0, st, b)
  if (f16) { if (use_nt) W64_(true, 2); else W64_(true, 0); }
  else { if (use_nt) W64_(false, 2); else W64_(false, 0); }
#undef W64_
  return (int)hipGetLastError();
}

}  // namespace xamd
