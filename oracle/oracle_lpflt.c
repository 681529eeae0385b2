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
