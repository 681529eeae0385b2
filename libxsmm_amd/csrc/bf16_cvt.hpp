// bf16_cvt.hpp -- f32 -> bf16 for C stores, the reference's conversion EXACTLY [ref: src/libxsmm_math.c:684-704: RNE, f32 denormals become signed zeros first, NaNs quieted].
// v_cvt_pk_bf16_f32 gives the reference's half for every one of the 2^32 f32 patterns that is not a denormal; a denormal it rounds instead of flushing (16 711 678
// patterns differ).  With the wave's FP32 denormal mode switched to "flush" (MODE bits 4..5 = 0) the instruction flushes a denormal INPUT to a signed zero first and then
// agrees with the reference on ALL 2^32 patterns (tools/cvt_probe.hip, exhaustive, both modes; profiles/r06_cvt_probe.txt).  So every bf16 store of the library brackets
// its conversions with two s_setreg_imm32_b32 -- the way the compiler itself brackets the FMA chain of an f32 division on gfx9 -- inside ONE asm statement, so that no
// other f32 arithmetic can be scheduled into the window.  Kernels are compiled with FP32 denormals on (.amdhsa_float_denorm_mode_32 3), which is what is restored.
// Until round 5 the GEMM epilogues used the bare instruction (a documented deviation below 1.2e-38).  The first round-6 form (v_cmp_class per value + a wave-uniform
// branch around the selects) measured 2 % (32^3), 3 % (BCSC, 4096^3 blocked), 9 % (72^3) and 14 % (40^3) slower than the bare instruction on one box
// (profiles/r06_bf16_store_ab.jsonl); this form costs two scalar instructions per group of conversions.
#pragma once
#include <hip/hip_runtime.h>

namespace xamd {

typedef __bf16 bfc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bfc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int bf16_pk_hw(float lo, float hi) {          // bare v_cvt_pk_bf16_f32: lo -> bits 0..15, hi -> bits 16..31 (for values that cannot be denormal)
  const bfc_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bfc_bf16x2));
}
#if defined(XAMD_BF16_STORE_HW)      // A/B build only: the bare instruction, as until round 5
#define XAMD_FLUSH_ON_
#define XAMD_FLUSH_OFF_
#else
#define XAMD_FLUSH_ON_ "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0\n\t"
#define XAMD_FLUSH_OFF_ "\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 3"
#endif
// (the result register of a pair is the register of its first value: %g is read and written, %(N + g) is the pair's second value)
#define XAMD_CVT_(g_, h_) "v_cvt_pk_bf16_f32 %" #g_ ", %" #g_ ", %" #h_
// one pair, exact
__device__ __forceinline__ unsigned int bf16_pk_exact(float lo, float hi) {
  unsigned int io = __float_as_uint(lo);
  asm volatile(XAMD_FLUSH_ON_ XAMD_CVT_(0, 1) XAMD_FLUSH_OFF_ : "+v"(io) : "v"(hi));
  return io;
}
// N pairs (x[2 g], x[2 g + 1]) -> out[g], exact, one mode window for all of them (N = 1, 2, 4, 8)
template <int N> __device__ __forceinline__ void bf16_pk_exact_n(const float (&x)[2 * N], unsigned int (&out)[N]) {
  static_assert(N == 1 || N == 2 || N == 4 || N == 8, "groups of 1, 2, 4 or 8 pairs");
#pragma unroll
  for (int g = 0; g < N; ++g) out[g] = __float_as_uint(x[2 * g]);
  if constexpr (N == 1) asm volatile(XAMD_FLUSH_ON_ XAMD_CVT_(0, 1) XAMD_FLUSH_OFF_ : "+v"(out[0]) : "v"(x[1]));
  else if constexpr (N == 2)
    asm volatile(XAMD_FLUSH_ON_ XAMD_CVT_(0, 2) "\n\t" XAMD_CVT_(1, 3) XAMD_FLUSH_OFF_ : "+v"(out[0]), "+v"(out[1]) : "v"(x[1]), "v"(x[3]));
  else if constexpr (N == 4)
    asm volatile(XAMD_FLUSH_ON_ XAMD_CVT_(0, 4) "\n\t" XAMD_CVT_(1, 5) "\n\t" XAMD_CVT_(2, 6) "\n\t" XAMD_CVT_(3, 7) XAMD_FLUSH_OFF_
                 : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]) : "v"(x[1]), "v"(x[3]), "v"(x[5]), "v"(x[7]));
  else
    asm volatile(XAMD_FLUSH_ON_ XAMD_CVT_(0, 8) "\n\t" XAMD_CVT_(1, 9) "\n\t" XAMD_CVT_(2, 10) "\n\t" XAMD_CVT_(3, 11) "\n\t"
                 XAMD_CVT_(4, 12) "\n\t" XAMD_CVT_(5, 13) "\n\t" XAMD_CVT_(6, 14) "\n\t" XAMD_CVT_(7, 15) XAMD_FLUSH_OFF_
                 : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]), "+v"(out[4]), "+v"(out[5]), "+v"(out[6]), "+v"(out[7])
                 : "v"(x[1]), "v"(x[3]), "v"(x[5]), "v"(x[7]), "v"(x[9]), "v"(x[11]), "v"(x[13]), "v"(x[15]));
}

}  // namespace xamd
