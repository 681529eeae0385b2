// gemm_small_kernels.hip -- 16 x 16 x 16 f32 / bf16 problems, round 4: every global access a 16-byte access of a contiguous 1 KiB run.
//
// Semantics as gemm_kernels.hip [ref: src/generator_gemm_reference_impl.c:1359-1426 (f32), :2127-2170 / :2367-2419 (bf16)]; NN, beta = 0,
// plain epilogue, strided operands (launch_gemm's p16_ok).  A 16^3 problem is 3 KiB (f32) or 1.5 KiB (bf16): the round-2 kernel
// (gemm_p16_kernel) fetched A as dwords straight into MFMA operand order -- four 64-byte runs per instruction -- and measured 0.63 of the
// HBM roofline at 65 536 problems.  Here A (rows contiguous, the operand whose register layout does not match its memory layout) travels
// global -> LDS by DMA, ONE 16-byte request per lane for a whole 1 KiB tile (f32: one problem, bf16: two), and is read back in operand order
// with conflict-free ds_read_b32; B and C were 16- / 8-byte accesses of contiguous tiles already.  The image is wave-private: no barrier.
//   f32 : v_mfma_f32_16x16x4_f32.  Lane (x = lane & 15, g = lane >> 4) supplies A(i = x, k = 4g + s) and B(k = 4g + s, j = x), s = 0..3 --
//         the summation order of gemm_p16_kernel, bit for bit.  Image [16 k][16 i] dwords; row k sits at position k ^ ((k >> 2) & 1), so the
//         two rows a half-wave reads together (k = s and k = 4 + s) fall into different halves of the 32 banks.
//   bf16: v_mfma_f32_16x16x16_bf16.  A in VNNI-2: image [8 k pairs][16 i] dwords per problem, pair kp at position kp ^ ((kp >> 1) & 1); lanes
//         0-31 of the request fetch problem 2w, lanes 32-63 problem 2w + 1.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "internal.hpp"
#include "gemm_device.hpp"
#include "bf16_cvt.hpp"

namespace xamd {

typedef short bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int small_cvt_pk_bf16(float lo, float hi) { return bf16_pk_exact(lo, hi); }       // the reference's conversion exactly (bf16_cvt.hpp)

template <int AUX>
__global__ __launch_bounds__(256) void gemm_f32_p16w_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds_all[4][256];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int bidx = logical_block(p) * 4u + wave;
  if (bidx >= p.nbatch) return;
  const unsigned int lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
  float* img = lds_all[wave];
  const BatchPtrs q = batch_ptrs(p, bidx);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int pos = lane >> 2;                                               // image row this lane's 16 bytes land in
  const unsigned int vA = ((pos ^ ((pos >> 2) & 1u)) * lda + (lane & 3u) * 4u) * 4u;   // ... which holds k = pos ^ ((pos >> 2) & 1)
  const unsigned int vB = (x * ldb + 4u * g) * 4u;
  const unsigned int kchunks = (unsigned int)p.k >> 4;
  f32x4 acc = (f32x4)0.0f;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br;
    br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar), rb = wave_rsrc(br);
    for (unsigned int kc = 0; kc < kchunks; ++kc) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)img, 16, (int)vA, (int)(kc * 64u * lda), 0, AUX);
      const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)vB, (int)(kc * 64u), AUX));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float af[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) af[s] = img[((4u * g + s) ^ (g & 1u)) * 16u + x];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bv[s], acc, 0, 0, 0);
    }
  }
  st_stream((GM f32x4*)((GM float*)q.c + (unsigned long long)x * (unsigned int)p.ldc + 4u * g), acc);
}

template <int AUX>
__global__ __launch_bounds__(256) void gemm_bf16_p16w_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned int lds_all[4][256];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int first = (logical_block(p) * 4u + wave) * 2u;
  if (first >= p.nbatch) return;
  const bool two = first + 1u < p.nbatch;                                           // wave-uniform
  const unsigned int lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
  unsigned int* img = lds_all[wave];
  BatchPtrs q[2];
  q[0] = batch_ptrs(p, first); q[1] = batch_ptrs(p, two ? first + 1u : first);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int l5 = lane & 31u, pos = l5 >> 2;
  const unsigned int vA = ((pos ^ ((pos >> 1) & 1u)) * lda + (l5 & 3u) * 4u) * 4u;     // dword (k pair, row): pair pos ^ ((pos >> 1) & 1), rows 4 (l5 & 3) ..
  const unsigned int vB = (x * ldb + 4u * g) * 2u;
  const unsigned int kchunks = (unsigned int)p.k >> 4;
  f32x4 acc[2] = {(f32x4)0.0f, (f32x4)0.0f};
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar[2], br[2];
    br_base(p, q[0], r, ar[0], br[0]); br_base(p, q[1], r, ar[1], br[1]);
    gcptr amine = (lane >> 5) ? ar[1] : ar[0];                                      // per-lane base: the upper half-wave fetches the second problem's A
    for (unsigned int kc = 0; kc < kchunks; ++kc) {
      __builtin_amdgcn_global_load_lds((GM const void*)(amine + vA + (unsigned long long)kc * 32ull * lda), (lds_vptr)img, 16, 0, AUX);
      u32x2_t bv[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) bv[pp] = *(GM const u32x2_t*)(br[pp] + vB + kc * 32u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      u32x2_t av[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int e = 0; e < 2; ++e) av[pp][e] = img[pp * 128u + ((2u * g + e) ^ (g & 1u)) * 16u + x];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
        acc[pp] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4_t, av[pp]), __builtin_bit_cast(bf16x4_t, bv[pp]), acc[pp], 0, 0, 0);
    }
  }
  const bool c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    if (pp == 0 || two) {
      if (c_f32) st_stream((GM f32x4*)((GM float*)q[pp].c + (unsigned long long)x * (unsigned int)p.ldc + 4u * g), acc[pp]);
      else { u32x2_t v; v[0] = small_cvt_pk_bf16(acc[pp][0], acc[pp][1]); v[1] = small_cvt_pk_bf16(acc[pp][2], acc[pp][3]);
             st_stream((GM u32x2_t*)((GM unsigned short*)q[pp].c + (unsigned long long)x * (unsigned int)p.ldc + 4u * g), v); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Waves that WALK several problems (round 4, second form): the kernels above are one-shot waves -- pointers, one round trip to memory, a handful of
// MFMAs, a store, exit -- so at 65 536 problems the chip launches 16 - 32 K waves that each spend their life waiting for ONE latency (0.76 / 0.68 of the
// HBM roofline, f32 / bf16).  Here a wave owns `per_wave` consecutive steps (f32: a problem, bf16: a pair of problems) and runs them as a two-deep
// software pipeline: the requests of step t + 1 (A by LDS-DMA into the other half of a two-slot image, B into the other register set) are issued
// BEFORE the wait for step t, so a wave always has the next step's 1.5 - 3 KiB in flight while it multiplies and stores.  s_waitcnt counts this
// wave's loads and stores in issue order: behind the loads of step t are the stores of step t - 1 and the loads of step t + 1.
// One k-chunk and one batch-reduce block per problem (k == 16, br_count == 1): everything else keeps the one-shot kernels.
// ------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// B travels by LDS-DMA as well: a load with a register destination inside the loop makes the compiler place its own (conservative: vmcnt(0) at the loop
// head) wait in front of the MFMA, which ends the overlap.  With both operands in LDS every wait in the loop is the explicit one.
// f32 : B image [16 columns][64 bytes]; the 16-byte piece g of column c sits in slot g ^ 2 (c >> 3): the four lanes of a ds_read_b128 lane group that
//       share c mod 4 (same 16 banks) then read four different slots.
// bf16: B image [16 columns][32 bytes] per problem; the 8-byte piece g of column c at byte (8 g) ^ 16 (c >> 3): lanes 0-31 of a ds_read_b64 hit 64 banks.
// SHB: B is ONE tile shared by the whole batch (batch stride 0: the weights of a layer): requested once per wave, kept in the image of slot 0, one request per step.
template <int AUX, bool SHB = false>
__global__ __launch_bounds__(256) void gemm_f32_p16s_kernel(GemmArgs p, unsigned int per_wave) {
  __shared__ __attribute__((aligned(16))) float lds_all[4][2][512];                // per wave and slot: A image (1 KiB) | B image (1 KiB)
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int t0 = (blockIdx.x * 4u + wave) * per_wave;
  if (t0 >= p.nbatch) return;
  const unsigned int np = (p.nbatch - t0 < per_wave) ? p.nbatch - t0 : per_wave;
  const unsigned int lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int pos = lane >> 2;
  const unsigned int vA = ((pos ^ ((pos >> 2) & 1u)) * lda + (lane & 3u) * 4u) * 4u;
  const unsigned int vB = (pos * ldb + 4u * ((lane & 3u) ^ ((pos >> 3) << 1))) * 4u;     // DMA lane = (column pos, slot lane & 3)
  const unsigned int rB = 256u + x * 16u + 4u * (g ^ ((x >> 3) << 1));                    // dword index of this lane's B operand inside a slot
  const unsigned long long vC = ((unsigned long long)x * (unsigned int)p.ldc + 4u * g) * 4ull;
  gptr cq[2];
  auto issue = [&](auto sc, unsigned int t) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    const long long e = (long long)(t0 + t);                                        // plain strided 1-D batch, one block per problem (the launcher checks)
    __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)p.a + e * p.bs_a + vA), (lds_vptr)lds_all[wave][S], 16, 0, AUX);
    if constexpr (!SHB) __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)p.b + e * p.bs_b + vB), (lds_vptr)(lds_all[wave][S] + 256), 16, 0, AUX);
    cq[S] = (gptr)p.c + e * p.bs_c;
  };
  constexpr int L = SHB ? 1 : 2;                                                     // requests per step
  auto step = [&](auto sc, unsigned int t) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    const bool more = t + 1u < np;
    if (more) issue(std::integral_constant<int, S ^ 1>{}, t + 1u);
    // younger than this step's two requests: the store of the step before and the next step's two requests
    if (t == 0u) { if (more) wait_vm<L>(); else wait_vm<0>(); }
    else { if (more) wait_vm<L + 1>(); else wait_vm<1>(); }
    const float* img = lds_all[wave][S];
    float af[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) af[s] = img[((4u * g + s) ^ (g & 1u)) * 16u + x];
    const f32x4 bv = *(const f32x4*)((SHB ? lds_all[wave][0] : img) + rB);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 acc = (f32x4)0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bv[s], acc, 0, 0, 0);
    st_stream((GM f32x4*)(cq[S] + vC), acc);
  };
  if constexpr (SHB) __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)p.b + vB), (lds_vptr)(lds_all[wave][0] + 256), 16, 0, 0);     // older than every step's request
  issue(std::integral_constant<int, 0>{}, 0u);
  for (unsigned int t = 0; t < np; t += 2u) {
    step(std::integral_constant<int, 0>{}, t);
    if (t + 1u < np) step(std::integral_constant<int, 1>{}, t + 1u);
  }
}

template <int AUX, bool SHB = false>
__global__ __launch_bounds__(256) void gemm_bf16_p16s_kernel(GemmArgs p, unsigned int per_wave) {
  __shared__ __attribute__((aligned(16))) unsigned int lds_all[4][2][512];         // per wave and slot: A images of the pair (2 x 512 B) | B images (2 x 512 B)
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int npairs = (p.nbatch + 1u) / 2u;
  const unsigned int t0 = (blockIdx.x * 4u + wave) * per_wave;                      // in pairs of problems
  if (t0 >= npairs) return;
  const unsigned int np = (npairs - t0 < per_wave) ? npairs - t0 : per_wave;
  const unsigned int lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int l5 = lane & 31u, pos = l5 >> 2, col = l5 >> 1;
  const unsigned int vA = ((pos ^ ((pos >> 1) & 1u)) * lda + (l5 & 3u) * 4u) * 4u;
  const unsigned int vB = col * ldb * 2u + 16u * ((l5 & 1u) ^ (col >> 3));             // DMA lane = (problem lane >> 5, column col, half l5 & 1)
  const unsigned int rB = 256u + x * 8u + ((2u * g) ^ (4u * (x >> 3)));                // dword index of this lane's 8 bytes of B inside a slot (problem 0)
  const bool c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
  const unsigned long long vC = ((unsigned long long)x * (unsigned int)p.ldc + 4u * g) * (c_f32 ? 4ull : 2ull);
  gptr cq[2][2]; bool two[2];
  auto issue = [&](auto sc, unsigned int t) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    const unsigned int first = 2u * (t0 + t);
    two[S] = first + 1u < p.nbatch;
    const long long e0 = (long long)first, e1 = (long long)(two[S] ? first + 1u : first);     // plain strided 1-D batch, one block per problem (the launcher checks)
    const long long emine = (lane >> 5) ? e1 : e0;                                  // the upper half-wave fetches the pair's second problem
    __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)p.a + emine * p.bs_a + vA), (lds_vptr)lds_all[wave][S], 16, 0, AUX);
    if constexpr (!SHB) __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)p.b + emine * p.bs_b + vB), (lds_vptr)(lds_all[wave][S] + 256), 16, 0, AUX);
    cq[S][0] = (gptr)p.c + e0 * p.bs_c; cq[S][1] = (gptr)p.c + e1 * p.bs_c;
  };
  constexpr int L = SHB ? 1 : 2;                                                     // requests per step
  auto step = [&](auto sc, unsigned int t) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    const bool more = t + 1u < np;
    if (more) issue(std::integral_constant<int, S ^ 1>{}, t + 1u);
    // younger than this step's two requests: the two stores of the step before (a pair that is not the last one is always whole) and the next step's two requests
    if (t == 0u) { if (more) wait_vm<L>(); else wait_vm<0>(); }
    else { if (more) wait_vm<L + 2>(); else wait_vm<2>(); }
    const unsigned int* img = lds_all[wave][S];
    u32x2_t av[2], bv[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) av[pp][e] = img[pp * 128u + ((2u * g + e) ^ (g & 1u)) * 16u + x];
      bv[pp] = *(const u32x2_t*)((SHB ? lds_all[wave][0] : img) + rB + pp * 128u);       // (SHB: both halves of the wave requested the same tile: the two images are equal)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4_t, av[pp]), __builtin_bit_cast(bf16x4_t, bv[pp]), (f32x4)0.0f, 0, 0, 0);
      if (pp == 0 || two[S]) {
        if (c_f32) st_stream((GM f32x4*)(cq[S][pp] + vC), acc);
        else { u32x2_t v; v[0] = small_cvt_pk_bf16(acc[0], acc[1]); v[1] = small_cvt_pk_bf16(acc[2], acc[3]); st_stream((GM u32x2_t*)(cq[S][pp] + vC), v); }
      }
    }
  };
  if constexpr (SHB) __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)p.b + vB), (lds_vptr)(lds_all[wave][0] + 256), 16, 0, 0);     // older than every step's request
  issue(std::integral_constant<int, 0>{}, 0u);
  for (unsigned int t = 0; t < np; t += 2u) {
    step(std::integral_constant<int, 0>{}, t);
    if (t + 1u < np) step(std::integral_constant<int, 1>{}, t + 1u);
  }
}

// 16-byte aligned A rows on top of launch_gemm's p16_ok (B, C and the strided forms are checked there); *taken = 0: the caller's older kernel serves
int launch_gemm_p16w(const GemmArgs& a, bool nt, void* stream, const char** kernel_name, int* taken) {
  constexpr bool off = false;
  *taken = 0;
  const bool bf16 = a.a_type == LIBXSMM_DATATYPE_BF16;
  unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | (unsigned long long)((long long)a.lda * 4);
  if (a.br_mode == 3) abits |= (unsigned long long)a.br_stride_a;
  if (off || (abits & 15ull) || a.lda >= (1 << 22) || a.ldb >= (1 << 22) || a.ldc >= (1 << 22)) return 0;
  hipStream_t st = (hipStream_t)stream;
  *taken = 1;
  // waves that walk `per_wave` steps (a problem / a pair of problems) as a two-deep pipeline: single-block 16^3 problems, launches that keep at least ~8 K waves
  // (LIBXSMM_HIP_P16_PW = 0 switches the form off, N forces N steps per wave)
  constexpr int pw_env = -1;
  const unsigned long long bbits16 = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)((long long)a.ldb * (bf16 ? 2 : 4));
  if (pw_env != 0 && a.k == 16 && a.br_count == 1 && !a.batch_inner && !a.list_a && (bbits16 & 15ull) == 0) {
    const unsigned int steps = bf16 ? (a.nbatch + 1u) / 2u : a.nbatch;
    // measured (profiles/r04b_p16s_times.jsonl; f32 / bf16 fractions of the HBM roofline, one-shot waves -> 2 / 4 / 8 / 16 steps per wave): 65 536 problems 0.745 / 0.675 ->
    // 0.779 / 0.721, 0.761 / 0.719, 0.737 / 0.701, 0.706 / 0.663; 524 288 problems 0.796 / 0.769 -> 0.794 / 0.787, 0.821 / 0.827, 0.743 / 0.787, 0.728 / 0.739
    // (workgroups of 8 / 16 waves instead of 4 -- fewer workgroups for the dispatcher to start -- measured slower on the small launches: 4096 problems f32 / bf16 3.43 / 2.76 us ->
    //  3.53 / 2.91 -> 4.17 / 3.59 us, and no better on the large ones: not kept)
    unsigned int pw = pw_env > 0 ? (unsigned int)pw_env : (steps >= 131072u ? 4u : steps >= 2048u ? 2u : 1u);        // small launches too: 4096 problems 3.73 / 3.88 -> 3.43 / 2.74 us
    if (pw > 1u) {
      const dim3 grid((unsigned int)(((steps + pw - 1u) / pw + 3u) / 4u));
      const bool shb = a.bs_b == 0;                          // one B tile for the whole batch
      if (bf16) {
        if (kernel_name) *kernel_name = "gemm_bf16_p16s_kernel";
        if (shb) { if (nt) hipLaunchKernelGGL((gemm_bf16_p16s_kernel<2, true>), grid, dim3(256), 0, st, a, pw); else hipLaunchKernelGGL((gemm_bf16_p16s_kernel<0, true>), grid, dim3(256), 0, st, a, pw); }
        else { if (nt) hipLaunchKernelGGL((gemm_bf16_p16s_kernel<2>), grid, dim3(256), 0, st, a, pw); else hipLaunchKernelGGL((gemm_bf16_p16s_kernel<0>), grid, dim3(256), 0, st, a, pw); }
      } else {
        if (kernel_name) *kernel_name = "gemm_f32_p16s_kernel";
        if (shb) { if (nt) hipLaunchKernelGGL((gemm_f32_p16s_kernel<2, true>), grid, dim3(256), 0, st, a, pw); else hipLaunchKernelGGL((gemm_f32_p16s_kernel<0, true>), grid, dim3(256), 0, st, a, pw); }
        else { if (nt) hipLaunchKernelGGL((gemm_f32_p16s_kernel<2>), grid, dim3(256), 0, st, a, pw); else hipLaunchKernelGGL((gemm_f32_p16s_kernel<0>), grid, dim3(256), 0, st, a, pw); }
      }
      return (int)hipGetLastError();
    }
  }
  if (bf16) {
    if (kernel_name) *kernel_name = "gemm_bf16_p16w_kernel";
    const dim3 grid((unsigned int)(((a.nbatch + 1u) / 2u + 3u) / 4u));
    if (nt) hipLaunchKernelGGL((gemm_bf16_p16w_kernel<2>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_bf16_p16w_kernel<0>), grid, dim3(256), 0, st, a);
  } else {
    if (kernel_name) *kernel_name = "gemm_f32_p16w_kernel";
    const dim3 grid((unsigned int)((a.nbatch + 3u) / 4u));
    if (nt) hipLaunchKernelGGL((gemm_f32_p16w_kernel<2>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_f32_p16w_kernel<0>), grid, dim3(256), 0, st, a);
  }
  return (int)hipGetLastError();
}

}  // namespace xamd
