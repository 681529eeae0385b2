// probe of v_cvt_scalef32_pk_bf16_fp4: nibble order, scale handling (tools only; not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* sc, unsigned int* out) {
  const unsigned int w = threadIdx.x | ((threadIdx.x ^ 0x5au) << 8);
  for (int s = 0; s < 4; ++s) {
    out[(s * 256 + threadIdx.x) * 2 + 0] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc[s], 0));
    out[(s * 256 + threadIdx.x) * 2 + 1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc[s], 1));
  }
}
static float bf(unsigned int h) { unsigned int u = h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
int main() {
  float hs[4] = {1.0f, 0.25f, 0.0f, 3.0f}; float* ds; unsigned int* dout; unsigned int ho[2048];
  hipMalloc(&ds, 16); hipMalloc(&dout, sizeof(ho)); hipMemcpy(ds, hs, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, ds, dout); hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  const float lut[16] = {0, .5f, 1, 1.5f, 2, 3, 4, 6, -0.f, -.5f, -1, -1.5f, -2, -3, -4, -6};
  int bad = 0;
  for (int s = 0; s < 4; ++s) for (int x = 0; x < 256; ++x) {
    const unsigned int r = ho[(s * 256 + x) * 2];
    const float lo = bf(r & 0xffff), hi = bf(r >> 16), elo = lut[x & 15] * hs[s], ehi = lut[x >> 4] * hs[s];
    if (lo != elo || hi != ehi) { if (bad < 12) printf("s=%g x=%02x got (%g,%g) expect (%g,%g)\n", hs[s], x, lo, hi, elo, ehi); ++bad; }
  }
  printf("byte0 mismatches vs low-nibble-first LUT*scale: %d of 1024\n", bad);
  bad = 0;
  for (int x = 0; x < 256; ++x) { const unsigned int r = ho[x * 2 + 1]; const int y = x ^ 0x5a; if (bf(r & 0xffff) != lut[y & 15] || bf(r >> 16) != lut[y >> 4]) ++bad; }
  printf("byte1 mismatches: %d of 256\n", bad);
  return 0;
}
