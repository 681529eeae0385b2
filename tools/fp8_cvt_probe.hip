// Compares the hardware conversions v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 (gfx950: OCP E4M3 / E5M2), fed with an IEEE half widened to f32, with the
// reference's half -> HF8 / BF8 roundings (lowp.hpp, bit-identical to src/libxsmm_math.c) over all 65 536 halves.  Prints the mismatches by class:
// the 8-bit-C epilogue of the fp8 GEMM kernels may use the instruction wherever the classes agree.
//   hipcc --offload-arch=gfx950 -O2 -I libxsmm_amd/csrc tools/fp8_cvt_probe.hip -o tools/fp8_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lowp.hpp"
__global__ void k(unsigned int* out) {      // out[h] = hw_hf8 | hw_bf8 << 8 | sw_hf8 << 16 | sw_bf8 << 24
  const unsigned int h = blockIdx.x * 256 + threadIdx.x;
  const float f = (float)__builtin_bit_cast(_Float16, (unsigned short)h);
  const unsigned int a = (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(f, f, 0, false) & 0xffu;
  const unsigned int b = (unsigned int)__builtin_amdgcn_cvt_pk_bf8_f32(f, f, 0, false) & 0xffu;
  out[h] = a | (b << 8) | ((unsigned int)lowp::f16_to_hf8_rne((unsigned short)h) << 16) | ((unsigned int)lowp::f16_to_bf8_rne((unsigned short)h) << 24);
}
int main() {
  unsigned int* d; hipMalloc(&d, 65536 * 4);
  k<<<256, 256>>>(d);
  static unsigned int r[65536];
  hipMemcpy(r, d, sizeof r, hipMemcpyDeviceToHost);
  const char* cls[] = {"nan", "inf", "finite above the largest", "subnormal or zero result", "normal"};
  for (int fmt = 0; fmt < 2; ++fmt) {
    unsigned int bad[5] = {0, 0, 0, 0, 0}, ex[5][3];
    for (unsigned int h = 0; h < 65536; ++h) {
      const unsigned int hw = (r[h] >> (8 * fmt)) & 0xffu, sw = (r[h] >> (16 + 8 * fmt)) & 0xffu;
      if (hw == sw) continue;
      const unsigned int e = (h >> 10) & 31u, m = h & 1023u;
      int c = 4;
      if (e == 31u) c = m ? 0 : 1;
      else if (fmt == 0 ? (sw & 0x7fu) == 0x7fu : (sw & 0x7fu) >= 0x7cu) c = 2;
      else if (fmt == 0 ? (sw & 0x78u) == 0u : (sw & 0x7cu) == 0u) c = 3;
      if (!bad[c]++) { ex[c][0] = h; ex[c][1] = hw; ex[c][2] = sw; }
    }
    printf("%s:", fmt == 0 ? "hf8 (E4M3)" : "bf8 (E5M2)");
    for (int c = 0; c < 5; ++c) { printf(" [%s] %u", cls[c], bad[c]); if (bad[c]) printf(" (half 0x%04x: hw 0x%02x, reference 0x%02x)", ex[c][0], ex[c][1], ex[c][2]); }
    printf("\n");
  }
  return 0;
}
