// Compares v_cvt_pk_bf16_f32 with the reference's software RNE (DAZ + NaN quieting) over all 2^32 f32 bit patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ unsigned short sw(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7f800000u) == 0u) u &= 0x80000000u;
  if ((u & 0x7f800000u) == 0x7f800000u) { if (u & 0x007fffffu) u |= 0x00400000u; }
  else u += 0x00007fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// round 6: the same instruction with the wave's FP32 denormal mode switched to flush around it (MODE bits 4..5 = 0): does it flush a denormal INPUT to a signed zero,
// as the reference does in front of its rounding?  (s_setreg_imm32_b32 is what the compiler itself brackets its f32 division's FMA chain with on gfx9.)
__device__ __forceinline__ unsigned int cvt_flush(float lo, float hi) {
  unsigned int out;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 3" : "=v"(out) : "v"(lo), "v"(hi));
  return out;
}
template <bool FLUSH>
__global__ void k(unsigned long long* counts, unsigned int* example) {
  const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  unsigned long long bad_den = 0, bad_nan = 0, bad_other = 0;
  for (unsigned int r = 0; r < 256; ++r) {
    const unsigned int u = (unsigned int)(t * 256 + r);
    const float f = __uint_as_float(u);
    const f32x2 v = {f, f};
    const unsigned int both = FLUSH ? cvt_flush(f, f) : __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hwbf16x2));
    const unsigned int hw = both & 0xffffu;
    if ((both >> 16) != hw) { ++bad_other; example[3] = u; example[4] = both >> 16; example[5] = hw; }
    const unsigned int s = sw(f);
    if (hw != s) {
      if ((u & 0x7f800000u) == 0u) ++bad_den;
      else if ((u & 0x7f800000u) == 0x7f800000u) { ++bad_nan; if (example[0] == 0) { example[0] = u; example[1] = hw; example[2] = s; } }
      else { ++bad_other; example[3] = u; example[4] = hw; example[5] = s; }
    }
  }
  if (bad_den) atomicAdd(&counts[0], bad_den);
  if (bad_nan) atomicAdd(&counts[1], bad_nan);
  if (bad_other) atomicAdd(&counts[2], bad_other);
}
int main() {
  unsigned long long* c; unsigned int* e;
  hipMalloc(&c, 24); hipMalloc(&e, 24); hipMemset(c, 0, 24); hipMemset(e, 0, 24);
  for (int flush = 0; flush < 2; ++flush) {
    hipMemset(c, 0, 24); hipMemset(e, 0, 24);
    if (flush) hipLaunchKernelGGL(k<true>, dim3(65536), dim3(256), 0, 0, c, e); else hipLaunchKernelGGL(k<false>, dim3(65536), dim3(256), 0, 0, c, e);
    unsigned long long hc[3]; unsigned int he[6];
    hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost); hipMemcpy(he, e, 24, hipMemcpyDeviceToHost);
    printf("%s: mismatch: denormal-in %llu, nan %llu, other %llu\n", flush ? "FP32 denormal mode = flush around the instruction" : "default mode", hc[0], hc[1], hc[2]);
    printf("  nan example in %08x hw %04x sw %04x ; other example in %08x hw %04x sw %04x\n", he[0], he[1], he[2], he[3], he[4], he[5]);
  }
  return 0;
}
