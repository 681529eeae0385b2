// gemm_sharedb_kernels.hip -- 64 x 64 x 64 f32 problems whose B is ONE block shared by the whole batch (batch stride 0: the weights of a layer), round 4.
//
// Semantics as gemm_kernels.hip [ref: src/generator_gemm_reference_impl.c:1359-1426]; NN, beta = 0, plain epilogue, one block per problem, strided 1-D batch.
// gemm_f32_wg64_kernel gives every problem a workgroup that brings in A AND B (16 KiB each): with a shared B a third of what passes through L1 is the same
// 16 KiB again and again (0.62 of the HBM roofline on the bytes that count, A + C, against 0.79 with private operands).  Here a workgroup is PERSISTENT over
// `per_wg` consecutive problems: B is fetched once, its MFMA operands (32 registers per wave: both 32-deep k steps of B block wj) stay in registers, and per
// problem only A travels -- by LDS-DMA into a two-problem ring (16 KiB per problem: [k step][A block][1024 floats]), two problems ahead of the multiply.
// Wave (wi, wj) multiplies A block wi with B block wj: the k order of gemm_f32_wg64_kernel, bit for bit.
// s_waitcnt counts this wave's loads AND stores in issue order: behind the requests of problem q are the 16 stores of problems q - 2 and q - 1 and the four
// requests of problem q + 1.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "internal.hpp"
#include "gemm_device.hpp"

namespace xamd {

template <int N> __device__ __forceinline__ void shb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ int shb_jl_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// image [k][i] linear (A: free index contiguous) / [j][8 x 16 B] XOR-swizzled (B: k contiguous): the layouts of gemm_f32_wg64_kernel
template <bool KCONTIG>
__device__ __forceinline__ void shb_frag_read(float (&w)[16], const float* lds, int lane) {
  const int li = lane & 31, h = lane >> 5;
  if (!KCONTIG) {
#pragma unroll
    for (int s = 0; s < 16; ++s) w[s] = lds[(2 * s + h) * 32 + li];
  } else {
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 x = ((const f32x4*)lds)[li * 8 + ((4 * h + q) ^ ((li >> 1) & 7))];
      v[4 * q + 0] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
      w[s] = __uint_as_float(r[0]); w[s + 8] = __uint_as_float(r[1]);
    }
  }
}

template <int AUX>
__global__ __launch_bounds__(256) void gemm_f32_wg64_sharedb_kernel(GemmArgs p, unsigned int per_wg) {
  __shared__ __attribute__((aligned(16))) float ring[2][2][2][1024];              // [problem parity][k step][block][k or j][...]: 32 KiB; the second half holds B in the prologue
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int p0 = blockIdx.x * per_wg;
  if (p0 >= p.nbatch) return;
  const unsigned int np = (p.nbatch - p0 < per_wg) ? p.nbatch - p0 : per_wg;
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  // request duty of wave w: (k step w >> 1, block w & 1) of A -- and, once, of B
  const unsigned int ks = w >> 1, blk = w & 1u;
  unsigned int offA[4], offB[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const unsigned int L = lane + 64u * x, hi = L >> 3, lo = L & 7u;
    offA[x] = ((32u * ks + hi) * lda + 32u * blk + lo * 4u) * 4u;                                       // row (k) 32 ks + hi, columns (i) 32 blk + 4 lo ..
    offB[x] = ((32u * blk + hi) * ldb + 32u * ks + ((lo ^ ((hi >> 1) & 7u)) * 4u)) * 4u;                // column (j) 32 blk + hi, k 32 ks + 4 (lo ^ swizzle) ..
  }
  auto issueA = [&](unsigned int q) __attribute__((always_inline)) {
    gcptr src = (gcptr)p.a + (long long)(p0 + q) * p.bs_a;
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(src + offA[x]), (lds_vptr)((char*)&ring[q & 1u][ks][blk][0] + 1024 * x), 16, 0, AUX);
  };
  const unsigned int wi = w & 1u, wj = w >> 1;
  // prologue: A of problem 0 into half 0, B into half 1; the B operands of both k steps into registers; then A of problem 1 takes B's place
  issueA(0);
#pragma unroll
  for (int x = 0; x < 4; ++x)
    __builtin_amdgcn_global_load_lds((GM const void*)((gcptr)p.b + offB[x]), (lds_vptr)((char*)&ring[1][ks][blk][0] + 1024 * x), 16, 0, 0);
  shb_wait_vm<0>();
  wg_barrier();
  float bf[2][16];
  shb_frag_read<true>(bf[0], &ring[1][0][wj][0], (int)lane);
  shb_frag_read<true>(bf[1], &ring[1][1][wj][0], (int)lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wg_barrier();                                                      // every wave holds its B operands: half 1 is free
  if (np > 1u) issueA(1);
  const unsigned int ldc = (unsigned int)p.ldc;
  for (unsigned int q = 0; q < np; ++q) {
    // the requests of problem q have landed: what may still be in flight behind them
    const bool next = q + 1u < np;
    if (q == 0u) { if (next) shb_wait_vm<4>(); else shb_wait_vm<0>(); }
    else if (q == 1u) { if (next) shb_wait_vm<20>(); else shb_wait_vm<16>(); }
    else { if (next) shb_wait_vm<36>(); else shb_wait_vm<32>(); }
    wg_barrier();                                                    // ... for every wave's part of the image
    float af[2][16];
    shb_frag_read<false>(af[0], &ring[q & 1u][0][wi][0], (int)lane);
    shb_frag_read<false>(af[1], &ring[q & 1u][1][wi][0], (int)lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (q + 2u < np) { wg_barrier(); issueA(q + 2u); }               // every wave has read problem q's image: problem q + 2 takes its place
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[0][s2], af[0][s2], acc, 0, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[1][s2], af[1][s2], acc, 0, 0, 0);
    GM float* c = (GM float*)((gptr)p.c + (long long)(p0 + q) * p.bs_c) + (32u * wj) * ldc + 32u * wi + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) st_stream(c + (unsigned int)shb_jl_of(r, (int)h) * ldc, acc[r]);        // exactly 16 stores: the wait counts above rely on it
  }
}

// (A bf16 sibling -- the structure of gemm_bf16_wg64_kernel made persistent the same way, B's operands of both k steps in 16 registers, A as two requests per wave and
//  problem -- was built and measured: 0.665 against 0.671 of the one-problem-per-workgroup kernel on 65 536 problems with bf16 C, bitwise-tolerance equal results.
//  There the shared B is not what holds the kernel back; it was removed again.)

// *taken = 0: the caller's other kernels serve
int launch_gemm_f32_wg64_sharedb(const GemmArgs& a, bool nt, void* stream, const char** kernel_name, int* taken) {
  constexpr int env = -1;      // 0: off, N: problems per workgroup
  *taken = 0;
  if (env == 0) return 0;
  const bool plain = a.m == 64 && a.n == 64 && a.k == 64 && a.br_count == 1 && a.bs_b == 0 && !a.batch_inner && !a.list_a && (a.flags & LIBXSMM_GEMM_FLAG_BETA_0) &&
    !(a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B)) && !a.colbias && !a.act && !a.vnni_c &&
    a.a_type == LIBXSMM_DATATYPE_F32 && a.b_type == LIBXSMM_DATATYPE_F32 && a.c_type == LIBXSMM_DATATYPE_F32;
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)((long long)a.lda * 4) | (unsigned long long)((long long)a.ldb * 4);
  const unsigned long long cbits = (unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c;
  if (!plain || (bits & 15ull) || (cbits & 3ull) || a.nbatch < 2048u || a.lda >= (1 << 20) || a.ldb >= (1 << 20) || a.ldc >= (1 << 20)) return 0;
  // problems per workgroup, measured on 65 536 problems (fraction of the HBM roofline on A + C; one workgroup per problem: 0.623): 2 -> 0.693, 3 -> 0.695, 4 -> 0.706,
  // 5 -> 0.697, 6 -> 0.687, 8 -> 0.668, 16 -> 0.655, 32 -> 0.650
  unsigned int per = env > 0 ? (unsigned int)env : 4u;
  *taken = 1;
  if (kernel_name) *kernel_name = "gemm_f32_wg64_sharedb_kernel";
  const dim3 grid((a.nbatch + per - 1u) / per);
  hipStream_t st = (hipStream_t)stream;
  if (nt) hipLaunchKernelGGL((gemm_f32_wg64_sharedb_kernel<2>), grid, dim3(256), 0, st, a, per);
  else hipLaunchKernelGGL((gemm_f32_wg64_sharedb_kernel<0>), grid, dim3(256), 0, st, a, per);
  return (int)hipGetLastError();
}

}  // namespace xamd
