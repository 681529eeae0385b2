// lowp.hpp -- 16/8-bit float conversions shared by the host helpers (utils.cpp) and the TPP kernels (meltw_kernels.hip).
// Behaviour = the reference's [ref: src/libxsmm_math.c:600-900]: IEEE half with RNE (f32 denormals flushed first, NaNs quieted),
// BF8 (E5M2) = RNE of the half's upper byte, HF8 (E4M3, bias 7, no infinities: specials and overflow -> NaN 0x7f).
// One generic "round the significand at bit s" routine instead of a function per format; pinned bit-exactly in tests/test_utils_cpu.py.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
# define LOWP_HD __host__ __device__ inline
#else
# define LOWP_HD inline
#endif

namespace lowp {

LOWP_HD uint32_t f32_bits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
LOWP_HD float bits_f32(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
LOWP_HD uint32_t rne_shift(uint32_t x, unsigned s) {            // round-to-nearest-even right shift
  if (s == 0) return x;
  if (s > 31) return 0;
  const uint32_t q = x >> s, rem = x & ((1u << s) - 1u), half = 1u << (s - 1);
  return q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
}

LOWP_HD uint16_t f32_to_f16(float in) {
  const uint32_t u = f32_bits(in), a = u & 0x7fffffffu;
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  if (a < 0x00800000u) return sign;                                                   // zero and f32 denormals (DAZ)
  if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? (((a >> 13) & 0x3ffu) | 0x200u) : 0u));
  const int e = (int)(a >> 23) - 127;
  if (e > 15) return (uint16_t)(sign | 0x7c00u);
  if (e < -25) return sign;
  const uint32_t mant = (a & 0x007fffffu) | 0x00800000u;                              // 1.m as a 24-bit integer
  if (e >= -14) return (uint16_t)(sign + (uint16_t)(((uint32_t)(e + 15) << 10) + (rne_shift(mant, 13) - 0x400u)));   // a carry walks into the exponent (up to inf)
  return (uint16_t)(sign | rne_shift(mant, (unsigned)(13 + (-14 - e))));               // subnormal half (0x400 = smallest normal)
}

LOWP_HD float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0x1fu) return bits_f32(sign | 0x7f800000u | (m ? ((m | 0x200u) << 13) : 0u));
  if (e == 0) {
    if (m == 0) return bits_f32(sign);
    int shift = 0;
    while (!(m & 0x400u)) { m <<= 1; ++shift; }                                        // normalise the subnormal
    return bits_f32(sign | ((uint32_t)(127 - 15 + 1 - shift) << 23) | ((m & 0x3ffu) << 13));
  }
  return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

// half -> E5M2: round the half's bit pattern at bit 8; infinities stay, NaNs are quieted
LOWP_HD uint8_t f16_to_bf8_rne(uint16_t h) {
  if ((h & 0x7c00u) == 0x7c00u) return (uint8_t)(((h & 0x3ffu) ? (h | 0x200u) : h) >> 8);
  return (uint8_t)((uint16_t)(h + 0x7fu + ((h >> 8) & 1u)) >> 8);
}

// half -> E4M3 (bias 7): no infinities, everything too large (and every special) becomes NaN 0x7f
LOWP_HD uint8_t f16_to_hf8_rne(uint16_t h) {
  const uint8_t sign = (uint8_t)((h & 0x8000u) >> 8);
  const uint32_t e16 = (h >> 10) & 0x1fu, m16 = h & 0x3ffu;
  if (e16 == 0x1fu || e16 > 23u || (e16 == 23u && m16 > 0x340u)) return (uint8_t)(sign | 0x7fu);
  if (e16 < 5u) return sign;                                                           // below half of the smallest subnormal 2^-9
  const int e = (int)e16 - 15;
  const uint32_t mant = m16 | 0x400u;
  if (e >= -6) return (uint8_t)(sign + (uint8_t)(((uint32_t)(e + 7) << 3) + (rne_shift(mant, 7) - 8u)));
  return (uint8_t)(sign | rne_shift(mant, (unsigned)(7 + (-6 - e))));
}

LOWP_HD float bf8_to_f32(uint8_t x) { return f16_to_f32((uint16_t)((uint16_t)x << 8)); }
LOWP_HD float hf8_to_f32(uint8_t x) {
  const uint32_t sign = (uint32_t)(x & 0x80u) << 24, e = (x >> 3) & 0xfu;
  uint32_t m = x & 7u;
  if (e == 0xfu && m == 7u) return bits_f32(sign | 0x7fc00000u);
  if (e == 0) {
    if (m == 0) return bits_f32(sign);
    int shift = 0;
    while (!(m & 8u)) { m <<= 1; ++shift; }
    return bits_f32(sign | ((uint32_t)(127 - 7 + 1 - shift) << 23) | ((m & 7u) << 20));
  }
  return bits_f32(sign | ((e + 120u) << 23) | (m << 20));
}

}  // namespace lowp
