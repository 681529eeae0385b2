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
