// utils.cpp -- helpers of include/libxsmm_utils.h: 16/8-bit float conversions, external RNG state, strings, matrix init.
// Host-only code.  Behaviour follows the reference's documented semantics [ref: src/libxsmm_math.c:600-900 (conversions),
// src/libxsmm_rng.c:172-210, src/libxsmm_lpflt_quant.c:303-370, include/libxsmm_math.h:17-55]; written from scratch as one
// generic "round the significand at bit s" routine instead of one hand-unrolled function per format.
#include "../../include/libxsmm_utils.h"
#include <cctype>
#include <cstdint>
#include <cstring>

namespace {

inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
// round-to-nearest-even right shift
inline uint32_t rne_shift(uint32_t x, unsigned s) {
  if (s == 0) return x;
  if (s > 31) return 0;
  const uint32_t q = x >> s, rem = x & ((1u << s) - 1u), half = 1u << (s - 1);
  return q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
}

uint16_t f32_to_f16(float in) {
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

float f16_to_f32(uint16_t h) {
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
uint8_t f16_to_bf8_rne(uint16_t h) {
  if ((h & 0x7c00u) == 0x7c00u) return (uint8_t)(((h & 0x3ffu) ? (h | 0x200u) : h) >> 8);
  return (uint8_t)((uint16_t)(h + 0x7fu + ((h >> 8) & 1u)) >> 8);
}

// half -> E4M3 (bias 7): no infinities, everything too large (and every special) becomes NaN 0x7f
uint8_t f16_to_hf8_rne(uint16_t h) {
  const uint8_t sign = (uint8_t)((h & 0x8000u) >> 8);
  const uint32_t e16 = (h >> 10) & 0x1fu, m16 = h & 0x3ffu;
  if (e16 == 0x1fu || e16 > 23u || (e16 == 23u && m16 > 0x340u)) return (uint8_t)(sign | 0x7fu);
  if (e16 < 5u) return sign;                                                           // below half of the smallest subnormal 2^-9
  const int e = (int)e16 - 15;
  const uint32_t mant = m16 | 0x400u;
  if (e >= -6) return (uint8_t)(sign + (uint8_t)(((uint32_t)(e + 7) << 3) + (rne_shift(mant, 7) - 8u)));
  return (uint8_t)(sign | rne_shift(mant, (unsigned)(7 + (-6 - e))));
}

float bf8_to_f32(uint8_t x) { return f16_to_f32((uint16_t)((uint16_t)x << 8)); }
float hf8_to_f32(uint8_t x) {
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

// one xoshiro128+ step of lane `lane` of a 4 x 16 word state [ref: src/libxsmm_lpflt_quant.c:303-330]
uint32_t xoshiro_lane(uint32_t* st, unsigned lane) {
  uint32_t s0 = st[lane], s1 = st[lane + 16], s2 = st[lane + 32], s3 = st[lane + 48];
  const uint32_t sum = s0 + s3, out = ((sum << 7) | (sum >> 25)) + s0, t = s1 << 9;
  s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = (s3 << 11) | (s3 >> 21);
  st[lane] = s0; st[lane + 16] = s1; st[lane + 32] = s2; st[lane + 48] = s3;
  return out;
}

size_t a_coprime(size_t n) {   // some number in [1, n) without a common factor with n, near n/2 + 1
  auto gcd = [](size_t a, size_t b) { while (b) { const size_t t = a % b; a = b; b = t; } return a; };
  if (n < 3) return 1;
  for (size_t c = n / 2 + 1; c < n; ++c) if (gcd(c, n) == 1) return c;
  return 1;
}

}  // namespace

LIBXSMM_API libxsmm_float16 libxsmm_convert_f32_to_f16(float in) { return f32_to_f16(in); }
LIBXSMM_API float libxsmm_convert_f16_to_f32(libxsmm_float16 in) { return f16_to_f32(in); }
LIBXSMM_API libxsmm_bfloat8 libxsmm_convert_f32_to_bf8_rne(float in) { return f16_to_bf8_rne(f32_to_f16(in)); }
LIBXSMM_API libxsmm_hfloat8 libxsmm_convert_f16_to_hf8_rne(libxsmm_float16 in) { return f16_to_hf8_rne(in); }
LIBXSMM_API libxsmm_hfloat8 libxsmm_convert_f32_to_hf8_rne(float in) { return f16_to_hf8_rne(f32_to_f16(in)); }
LIBXSMM_API float libxsmm_convert_bf8_to_f32(libxsmm_bfloat8 in) { return bf8_to_f32(in); }
LIBXSMM_API float libxsmm_convert_hf8_to_f32(libxsmm_hfloat8 in) { return hf8_to_f32(in); }

// stochastic rounding of the half's low byte: normal numbers add a random byte before truncation, subnormals round to nearest,
// infinities / NaNs as in the RNE form [ref: src/libxsmm_math.c:706-728]
static uint8_t bf8_stochastic(uint16_t h, unsigned int random_byte) {
  if ((h & 0x7c00u) == 0x7c00u) return (uint8_t)(((h & 0x3ffu) ? (h | 0x200u) : h) >> 8);
  if ((h & 0x7c00u) != 0) return (uint8_t)((uint16_t)(h + (random_byte & 0xffu)) >> 8);
  return (uint8_t)((uint16_t)(h + 0x7fu + ((h >> 8) & 1u)) >> 8);
}
LIBXSMM_API libxsmm_bfloat8 libxsmm_convert_f32_to_bf8_stochastic(float in, unsigned int seed) { return bf8_stochastic(f32_to_f16(in), seed & 0xffu); }

LIBXSMM_API void libxsmm_rne_convert_fp32_f16(const float* in, libxsmm_float16* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = f32_to_f16(in[i]); }
LIBXSMM_API void libxsmm_convert_f16_f32(const libxsmm_float16* in, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = f16_to_f32(in[i]); }
LIBXSMM_API void libxsmm_rne_convert_fp32_bf8(const float* in, libxsmm_bfloat8* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = f16_to_bf8_rne(f32_to_f16(in[i])); }
LIBXSMM_API void libxsmm_convert_bf8_f32(const libxsmm_bfloat8* in, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = bf8_to_f32(in[i]); }
LIBXSMM_API void libxsmm_rne_convert_fp32_hf8(const float* in, libxsmm_hfloat8* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = f16_to_hf8_rne(f32_to_f16(in[i])); }
LIBXSMM_API void libxsmm_convert_hf8_f32(const libxsmm_hfloat8* in, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = hf8_to_f32(in[i]); }
LIBXSMM_API void libxsmm_stochastic_convert_fp32_bf8(const float* in, libxsmm_bfloat8* out, unsigned int n, void* rng_state, unsigned int start_seed_idx) {
  for (unsigned int i = 0; i < n; ++i) {     // element i of a 16-wide group draws from lane (start + i % 16) % 16; the top byte decides
    const uint32_t r = xoshiro_lane((uint32_t*)rng_state, (start_seed_idx + (i & 15u)) & 15u);
    out[i] = bf8_stochastic(f32_to_f16(in[i]), r >> 24);
  }
}

LIBXSMM_API unsigned int* libxsmm_rng_create_extstate(unsigned int seed) {
  unsigned int* st = (unsigned int*)libxsmm_aligned_malloc(64 * sizeof(unsigned int), 64);
  if (!st) return nullptr;
  for (unsigned int w = 0; w < 4; ++w) for (unsigned int l = 0; l < 16; ++l) st[16 * w + l] = seed + 100u * w + 31u - l;
  for (unsigned int l = 0; l < 16; ++l) for (int warm = 0; warm < 64; ++warm) (void)xoshiro_lane(st, l);   // decorrelate the lanes
  return st;
}
LIBXSMM_API unsigned int libxsmm_rng_get_extstate_size(void) { return (unsigned int)(64 * sizeof(unsigned int)); }
LIBXSMM_API void libxsmm_rng_destroy_extstate(unsigned int* stateptr) { libxsmm_free(stateptr); }

LIBXSMM_API const char* libxsmm_stristrn(const char a[], const char b[], size_t maxlen) {
  if (!a || !b || !*a || !*b || maxlen == 0) return nullptr;
  for (const char* s = a; *s; ++s) {
    size_t i = 0;
    while (i < maxlen && b[i] && s[i] && std::tolower((unsigned char)s[i]) == std::tolower((unsigned char)b[i])) ++i;
    if (i == maxlen || !b[i]) return s;
  }
  return nullptr;
}
LIBXSMM_API const char* libxsmm_stristr(const char a[], const char b[]) { return libxsmm_stristrn(a, b, (size_t)-1); }

LIBXSMM_API double libxsmm_hip_matinit_value(double seed, double scale, libxsmm_blasint row, libxsmm_blasint col,
  libxsmm_blasint nrows, libxsmm_blasint ncols, libxsmm_blasint ld) {
  if (seed != 0) return row < nrows ? (seed * scale + scale) * (1.0 + (double)col * nrows + row) : seed;
  const size_t total = (size_t)ncols * (size_t)ld, k = (size_t)col * (size_t)ld + (size_t)row;
  const double half = (double)((total + 1) / 2);
  static thread_local size_t cached_n = 0, cached_c = 1;
  if (cached_n != total) { cached_c = a_coprime(total); cached_n = total; }
  return (scale / half) * ((double)((cached_c * k) % total) - half);
}
