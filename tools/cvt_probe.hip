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
__global__ void k(unsigned long long* counts, unsigned int* example) {
  const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  unsigned long long bad_den = 0, bad_nan = 0, bad_other = 0;
  for (unsigned int r = 0; r < 256; ++r) {
    const unsigned int u = (unsigned int)(t * 256 + r);
    const float f = __uint_as_float(u);
    const f32x2 v = {f, f};
    const unsigned int hw = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hwbf16x2)) & 0xffffu;
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
  hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, c, e);
  unsigned long long hc[3]; unsigned int he[6];
  hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost); hipMemcpy(he, e, 24, hipMemcpyDeviceToHost);
  printf("mismatch: denormal-in %llu, nan %llu, other %llu\n", hc[0], hc[1], hc[2]);
  printf("nan example in %08x hw %04x sw %04x ; other example in %08x hw %04x sw %04x\n", he[0], he[1], he[2], he[3], he[4], he[5]);
  return 0;
}
