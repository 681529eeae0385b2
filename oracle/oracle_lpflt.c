/*
 * oracle_lpflt.c -- low-precision float conversions and the comparison metric (test-only).
 * Restates  src/libxsmm_math.c:640-704  (bf16 truncate / RNE with denormals-are-zero and
 * NaN quieting) and the normf_rel metric of  src/libxsmm_matdiff.h:141-142  +
 * src/libxsmm_math.c:273  (sqrt of sum (r-t)^2 / sum r^2).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

static unsigned int f2u(float x) { unsigned int u; memcpy(&u, &x, 4); return u; }
static float u2f(unsigned int u) { float x; memcpy(&x, &u, 4); return x; }

float oracle_bf16_to_f32(unsigned short x) { return u2f((unsigned int)x << 16); }

/* shared front end: flush denormal inputs to signed zero, quiet NaNs, leave inf alone */
static unsigned int bf16_prepare(unsigned int u, int* special) {
  if ((u & 0x7f800000u) == 0) u &= 0x80000000u;                    /* DAZ */
  *special = ((u & 0x7f800000u) == 0x7f800000u);
  if (*special && (u & 0x007fffffu) != 0) u |= 0x00400000u;         /* quiet the NaN */
  return u;
}

unsigned short oracle_f32_to_bf16_trunc(float x) {
  int special; const unsigned int u = bf16_prepare(f2u(x), &special);
  return (unsigned short)(u >> 16);
}

unsigned short oracle_f32_to_bf16_rne(float x) {
  int special; unsigned int u = bf16_prepare(f2u(x), &special);
  if (!special) u += 0x00007fffu + ((u >> 16) & 1u);                /* round to nearest even */
  return (unsigned short)(u >> 16);
}

double oracle_normf_rel(int dtype, long long count, const void* ref, const void* tst) {
  double num = 0.0, den = 0.0; long long i;
  for (i = 0; i < count; ++i) {
    double r, t;
    switch (dtype) {
      case LIBXSMM_DATATYPE_F64: r = ((const double*)ref)[i]; t = ((const double*)tst)[i]; break;
      case LIBXSMM_DATATYPE_F32: r = ((const float*)ref)[i]; t = ((const float*)tst)[i]; break;
      case LIBXSMM_DATATYPE_BF16: r = oracle_bf16_to_f32(((const unsigned short*)ref)[i]);
                                  t = oracle_bf16_to_f32(((const unsigned short*)tst)[i]); break;
      case LIBXSMM_DATATYPE_I32: r = ((const int*)ref)[i]; t = ((const int*)tst)[i]; break;
      default: return -1.0;
    }
    num += (r - t) * (r - t); den += r * r;
  }
  if (den <= 0.0) return sqrt(num);
  return sqrt(num / den);
}

/* ---- 8-bit floats  [ref: src/libxsmm_math.c:546-585] ---------------------------------------
 * BF8 = E5M2 = the upper byte of an IEEE half; HF8 = E4M3 (bias 7, no infinities, S.1111.111 = NaN). */
static float half_bits_to_f32(unsigned short h) {
  const unsigned int s = (unsigned int)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  union { unsigned int u; float f; } r;
  if (e == 0) {
    if (m == 0) r.u = s;
    else {                                   /* subnormal half: value = m * 2^-24 (exact in f32) */
      r.f = (float)m * 5.9604644775390625e-08f;
      r.u |= s;
    }
  } else if (e == 31) r.u = s | 0x7f800000u | (m << 13);
  else r.u = s | ((e + 112u) << 23) | (m << 13);
  return r.f;
}
float oracle_bf8_to_f32(unsigned char x) { return half_bits_to_f32((unsigned short)((unsigned short)x << 8)); }
float oracle_hf8_to_f32(unsigned char in) {
  const unsigned int s = (unsigned int)(in & 0x80u) << 24, e = (in & 0x78u) >> 3;
  unsigned int m = in & 0x07u, e_norm = e + (127u - 7u);
  union { unsigned int u; float f; } r;
  if (e == 0 && m != 0) {                    /* subnormal: renormalise */
    unsigned int lz = 2;
    lz = (m > 0x1u) ? 1 : lz; lz = (m > 0x3u) ? 0 : lz;
    e_norm -= lz; m = (m << (lz + 1)) & 0x07u;
  } else if (e == 0 && m == 0) e_norm = 0;
  else if (e == 0xfu && m == 0x7u) { e_norm = 0xffu; m = 0x4u; }
  r.u = (e_norm << 23) | (m << 20) | s;
  return r.f;
}

/* ---- IEEE half and the narrowing conversions of the 8-bit floats  [ref: src/libxsmm_math.c:600-636 (f16 -> f32), :824-900 (f32 -> f16),
 * :731-746 (f32 -> bf8), :749-821 (f16 -> hf8)] ------------------------------------------------------------------------------------
 * Restated case by case like the reference: special values, overflow, flush below half of the smallest subnormal, subnormal results
 * with a sticky bit, normal results; RNE by adding (half ulp - 1 + lsb) to the bit pattern. */
float oracle_f16_to_f32(unsigned short h) {
  union { unsigned int u; float f; } r;
  const unsigned int s = (unsigned int)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu;
  if (e == 31u) { const unsigned int m = h & 0x3ffu; r.u = s | 0x7f800000u | ((m ? (m | 0x200u) : 0u) << 13); return r.f; }   /* NaNs quieted [:629-632] */
  return half_bits_to_f32(h);
}
unsigned short oracle_f32_to_f16(float x) {
  unsigned int u = f2u(x), s, e32, m32, e, m, fix;
  if ((u & 0x7f800000u) == 0) u &= 0x80000000u;                                   /* DAZ [:835-838] */
  s = (u & 0x80000000u) >> 16; e32 = (u & 0x7f800000u) >> 23; m32 = u & 0x007fffffu;
  if (e32 == 0xffu) { e = 0x1fu; m = (m32 == 0) ? 0 : ((m32 >> 13) | 0x200u); }  /* inf / NaN [:845-848] */
  else if (e32 > 127u + 15u) { e = 0x1fu; m = 0; }                               /* overflow -> inf */
  else if (e32 < 127u - 15u - 10u) { e = 0; m = 0; }                             /* < 2^-25 -> 0 */
  else if (e32 <= 127u - 15u) {                                                  /* subnormal half [:857-868] */
    m = (m32 | 0x00800000u) >> ((127u - 15u) + 1u - e32);
    m |= ((m32 & 0x1fffu) + 0x1fffu) >> 13;                                      /* sticky */
    fix = (m >> 13) & 1u; m = (m + 0x0fffu + fix) >> 13; e = 0;
  } else {                                                                       /* normal [:877-885] */
    fix = (m32 >> 13) & 1u; u = u + 0x0fffu + fix;
    e = ((u & 0x7f800000u) >> 23) - (127u - 15u); m = (u & 0x007fffffu) >> 13;
  }
  return (unsigned short)(s | (e << 10) | m);
}
unsigned char oracle_f32_to_bf8_rne(float x) {
  unsigned short h = oracle_f32_to_f16(x);
  const unsigned int fix = (h >> 8) & 1u;
  if ((h & 0x7c00u) == 0x7c00u) h = (unsigned short)((h & 0x03ffu) == 0 ? h : (h | 0x0200u));   /* no rounding of inf / NaN [:740-742] */
  else h = (unsigned short)(h + 0x007fu + fix);
  return (unsigned char)(h >> 8);
}
unsigned char oracle_f16_to_hf8_rne(unsigned short in) {
  const unsigned int s = (in & 0x8000u) >> 8, e16 = (in & 0x7c00u) >> 10, m16 = in & 0x03ffu;
  unsigned int e, m, fix;
  if (e16 == 0x1fu || e16 > 15u - 7u + 15u || (e16 == 15u - 7u + 15u && m16 > 0x0340u)) { e = 0xfu; m = 0x7u; }   /* specials, overflow -> NaN [:762-771] */
  else if (e16 < 15u - 7u - 3u) { e = 0; m = 0; }
  else if (e16 <= 15u - 7u) {                                                    /* subnormal [:777-789] */
    m = (m16 | 0x0400u) >> ((15u - 7u) + 1u - e16);
    m |= ((m16 & 0x007fu) + 0x007fu) >> 7;
    fix = (m >> 7) & 1u; m = (m + 0x003fu + fix) >> 7; e = 0;
  } else {                                                                       /* normal [:792-801] */
    unsigned int h = in;
    fix = (m16 >> 7) & 1u; h = (h + 0x003fu + fix) & 0xffffu;
    e = ((h & 0x7c00u) >> 10) - (15u - 7u); m = (h & 0x03ffu) >> 7;
  }
  return (unsigned char)(s | (e << 3) | m);
}
unsigned char oracle_f32_to_hf8_rne(float x) { return oracle_f16_to_hf8_rne(oracle_f32_to_f16(x)); }
