// gemm_device.hpp -- device helpers shared by the dense GEMM translation units (gemm_kernels.hip, gemm_f64_kernels.hip):
// address-space typedefs, compile-time loops, batch / batch-reduce addressing [ref: src/generator_gemm_reference_impl.c:180-197],
// streaming stores, the fence-free workgroup barrier and wave-uniform buffer resources.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "internal.hpp"

namespace xamd {

// Pointers that arrive inside the by-value argument block are "generic" to the compiler and would be
// accessed with flat_load/flat_store (slower, and they tie the LDS and VMEM wait counters together).
// Everything the kernels dereference is global memory: say so explicitly.
#define GM __attribute__((address_space(1)))
typedef GM const char* gcptr;
typedef GM char* gptr;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;

// compile-time loop: the body sees its index as a constant (no reliance on #pragma unroll, which the optimizer
// declines for bodies with convergent operations -- a dynamic index into an accumulator array means scratch memory)
template <typename F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct BatchPtrs { gcptr a; gcptr b; gptr c; gcptr d; GM unsigned char* mask; };
// A value every lane of the wave agrees on, made provably uniform (SGPR pair) for the compiler: anything
// loaded through a vector-memory load is "divergent" to it and would drag all address math into VGPRs.
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ gcptr list_entry(const void* list, unsigned long long i) {
  return (gcptr)(size_t)uniform_u64(((GM const unsigned long long*)list)[i]);
}
__device__ __forceinline__ BatchPtrs batch_ptrs(const GemmArgs& p, unsigned int bidx) {
  BatchPtrs q;
  if (p.batch_inner) {     // 2-D batch: (i, j) = (bidx % inner, bidx / inner); wave-uniform, one 32-bit division per wave
    unsigned int bi, bj;
    if (p.map2d_shift) {
      // Locality: hardware workgroup g runs on XCD g % 8 and the workgroups of an XCD start in increasing order.  The element grid is cut into
      // super-tiles of T x T elements (T = 2^shift); super-tile S goes to XCD S % 8 as T*T/4 consecutive workgroups of that XCD, so the waves
      // that are resident on one XCD at the same time work on a compact square: every A and B block they touch is shared by T of them
      // and the per-step working set (2 T blocks) fits the XCD's 4 MiB L2 many times over.  (bidx = 4 * workgroup + wave: one tile per element.)
      const unsigned int sh = p.map2d_shift, wg = bidx >> 2, x = wg & 7u, k = wg >> 3;
      const unsigned int wgs_shift = 2u * sh - 2u;                         // log2(workgroups per super-tile)
      const unsigned int S = x + 8u * (k >> wgs_shift), pl = ((k & ((1u << wgs_shift) - 1u)) << 2) | (bidx & 3u);
      const unsigned int nsi = p.batch_inner >> sh, sj = S / nsi, si = S - sj * nsi;
      bi = (si << sh) + (pl & ((1u << sh) - 1u)); bj = (sj << sh) + (pl >> sh);
    } else { bj = bidx / p.batch_inner; bi = bidx - bj * p.batch_inner; }
    q.a = (gcptr)p.a + (long long)bi * p.bs_a; q.b = (gcptr)p.b + (long long)bj * p.bs_b;
    q.c = (gptr)p.c + (long long)bi * p.bs_c + (long long)bj * p.bs_c2;
    q.d = p.d ? (gcptr)p.d + (long long)bi * p.bs_d : nullptr;
    q.mask = p.relu_mask ? (GM unsigned char*)p.relu_mask + (long long)bi * p.bs_mask + (long long)bj * p.bs_mask2 : nullptr;
    return q;
  }
  if (p.list_a) { q.a = list_entry(p.list_a, bidx); q.b = list_entry(p.list_b, bidx); q.c = (gptr)list_entry(p.list_c, bidx); }
  else { q.a = (gcptr)p.a + (long long)bidx * p.bs_a; q.b = (gcptr)p.b + (long long)bidx * p.bs_b; q.c = (gptr)p.c + (long long)bidx * p.bs_c; }
  q.d = p.d ? (gcptr)p.d + (long long)bidx * p.bs_d : nullptr;
  q.mask = p.relu_mask ? (GM unsigned char*)p.relu_mask + (long long)bidx * p.bs_mask : nullptr;
  return q;
}
// Workgroup -> logical block.  A 2-D batch re-uses operands between elements (A along j, B along i): hardware block b runs on
// XCD b % 8, so the grid is re-dealt to give every XCD (its own L2) one contiguous eighth of the element range -- a band of j
// whose B panels stay in that L2 while A streams through it once per XCD.  1-D batches have no reuse: identity.
__device__ __forceinline__ unsigned int logical_block(const GemmArgs& p) {
  const unsigned int b = blockIdx.x, nb = gridDim.x;
  if (p.batch_inner && !p.map2d_shift && (nb & 7u) == 0u) return (b & 7u) * (nb >> 3) + (b >> 3);
  return b;
}
// base of batch-reduce element r [ref: gemm ref :180-197]
__device__ __forceinline__ void br_base(const GemmArgs& p, const BatchPtrs& q, unsigned long long r, gcptr& a, gcptr& b) {
  if (p.br_mode == 1) { a = list_entry((const void*)(size_t)q.a, r); b = list_entry((const void*)(size_t)q.b, r); }
  else if (p.br_mode == 2) {
    a = q.a + (long long)uniform_u64((unsigned long long)((GM const long long*)p.offs_a)[r]);
    b = q.b + (long long)uniform_u64((unsigned long long)((GM const long long*)p.offs_b)[r]);
  }
  else if (p.br_mode == 3) { a = q.a + p.br_stride_a * (long long)r; b = q.b + p.br_stride_b * (long long)r; }
  else { a = q.a; b = q.b; }
}

// Streaming store: C tiles are written once and never read back by the kernel.  A/B on one box (same binary but for this switch, two
// rounds): headline 32^3 batch 4096 12.31 -> 11.60 us per step (+6 %), beta = 1 at batch 4096 +8 %, bf16 64^3 +4 %, mxfp4 -> f32 32^3 +3 %,
// large-batch f32 +1..2 %, i8 and fused bf16 unchanged; the 2x2-tile MX x MX kernel (f32 C = 80 % of its bytes) LOSES 3..7 % and opts out.
// Non-temporal stores in the BCSC and TPP kernels measured slower (0.57 -> 0.47, 0.78 -> 0.76) and were not kept.
template <bool NT = true, typename T> __device__ __forceinline__ void st_stream(GM T* p, T v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// Workgroup barrier WITHOUT the memory fence of __syncthreads().  The fence makes the compiler drain every outstanding vector-memory
// operation ("s_waitcnt vmcnt(0)" in front of s_barrier), which ends an LDS-DMA prefetch that is deliberately left in flight across the
// barrier.  What the pipelines here need is: this wave's LDS traffic is complete (lgkmcnt), its landed DMA is accounted for by the explicit
// vmcnt wait in front of the call, and nobody proceeds before everybody arrived.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// raw buffer resource over a wave-uniform base (gfx9 word 3: 32-bit raw data format); offsets are checked against 4 GiB only
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_rsrc(gcptr base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)uniform_u64((unsigned long long)(size_t)base), (short)0, -1, 0x00020000);
}

}  // namespace xamd
