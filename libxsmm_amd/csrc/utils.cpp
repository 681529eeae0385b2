// utils.cpp -- helpers of include/libxsmm_utils.h: 16/8-bit float conversions, external RNG state, strings, matrix init.
// Host-only code.  Behaviour follows the reference's documented semantics [ref: src/libxsmm_math.c:600-900 (conversions),
// src/libxsmm_rng.c:172-210, src/libxsmm_lpflt_quant.c:303-370, include/libxsmm_math.h:17-55]; the conversions themselves live in
// lowp.hpp (shared with the TPP kernels).
#include "../../include/libxsmm_utils.h"
#include "lowp.hpp"
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace {

using namespace lowp;

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

// 16 xoshiro128 lanes, word w of lane l seeded with seed + 100 w + 31 - l, each lane then moved 2^64 draws ahead with the generator's
// published jump polynomial (Blackman / Vigna) so that the lanes never overlap [ref: src/libxsmm_rng.c:30-60,172-197]
static void seed_lanes(uint32_t* st, unsigned int seed) {
  static const uint32_t jump_poly[4] = { 0x8764000bu, 0xf542d2d3u, 0x6fa035c3u, 0x77f2db5bu };
  for (unsigned int w = 0; w < 4; ++w) for (unsigned int l = 0; l < 16; ++l) st[16 * w + l] = seed + 100u * w + 31u - l;
  for (unsigned int l = 0; l < 16; ++l) {
    uint32_t acc[4] = { 0, 0, 0, 0 };
    for (unsigned int bit = 0; bit < 128; ++bit) {
      if ((jump_poly[bit >> 5] >> (bit & 31)) & 1u) for (unsigned int w = 0; w < 4; ++w) acc[w] ^= st[16 * w + l];
      (void)xoshiro_lane(st, l);
    }
    for (unsigned int w = 0; w < 4; ++w) st[16 * w + l] = acc[w];
  }
}
LIBXSMM_API unsigned int* libxsmm_rng_create_extstate(unsigned int seed) {
  unsigned int* st = (unsigned int*)libxsmm_aligned_malloc(64 * sizeof(unsigned int), 64);
  if (st) seed_lanes(st, seed);
  return st;
}

// the library's own lane state: libxsmm_rng_set_seed (re)seeds it together with the C library's generators; libxsmm_rng_f32_seq hands
// element i the next draw of lane i % 16, 23 random mantissa bits mapped to [0, 1) [ref: src/libxsmm_rng.c:62-119,211-260]
static uint32_t g_lanes[64];
LIBXSMM_API void libxsmm_rng_set_seed(unsigned int seed) { seed_lanes(g_lanes, seed); srand48((long)seed); srand(seed); }
LIBXSMM_API void libxsmm_rng_f32_seq(float* rngs, libxsmm_blasint count) {
  for (libxsmm_blasint i = 0; i < count; ++i) {
    const unsigned int l = (unsigned int)i & 15u;
    const uint32_t bits = 0x3f800000u | ((g_lanes[l] + g_lanes[48 + l]) >> 9);
    (void)xoshiro_lane(g_lanes, l);
    float f; std::memcpy(&f, &bits, 4);
    rngs[i] = f - 1.0f;
  }
}
LIBXSMM_API unsigned int libxsmm_rng_get_extstate_size(void) { return (unsigned int)(64 * sizeof(unsigned int)); }
LIBXSMM_API void libxsmm_rng_destroy_extstate(unsigned int* stateptr) { libxsmm_free(stateptr); }

LIBXSMM_API float libxsmm_sexp2_i8(signed char x) { return std::ldexp(1.0f, (int)x); }
LIBXSMM_API float libxsmm_sexp2_u8(unsigned char x) { return std::ldexp(1.0f, (int)x); }       // 2^128 and above: +infinity, like the f32 range demands
LIBXSMM_API float libxsmm_sexp2_i8i(int x) { return libxsmm_sexp2_i8((signed char)x); }
LIBXSMM_API double libxsmm_nearbyint(double x) { return std::nearbyint(x); }
LIBXSMM_API float libxsmm_nearbyintf(float x) { return std::nearbyintf(x); }
LIBXSMM_API double libxsmm_dsqrt(double x) { return std::sqrt(x); }
LIBXSMM_API float libxsmm_ssqrt(float x) { return std::sqrt(x); }

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
