/*
 * oracle_gemm.c -- CPU restatement of the reference GEMM/BRGEMM semantics (test-only).
 *
 * Follows  src/generator_gemm_reference_impl.c  of the reference:
 *   :180-197   batch-reduce addressing (address / offset / stride)
 *   :375-661   decoding of the run-time param slots
 *   :1322-1357 f64, :1359-1426 f32, :2127-2170 bf16->f32, :2367-2419 bf16->bf16 loop nests
 *   :294-372   fused pre-op (column bias) and post-op (ReLU / sigmoid / down-convert)
 *   :2802-2853 optional C -> VNNI re-layout and the top-level driver
 * Instead of one loop nest per datatype there is a single nest over (j, i, r, s) that reads
 * A/B through index helpers; the *order* of the floating-point operations is kept exactly:
 * per output element, serially over r then s, product rounded then added (no FMA), with the
 * two halves of a VNNI k-pair consumed high-k first (reference :2144, :2391).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct br_cursor {            /* where batch-reduce element r lives */
  const char* a; const char* b;
} br_cursor;

typedef struct gemm_view {
  const oracle_gemm_desc* d;
  const char* a0; const char* b0;     /* primary slots */
  const long long* offs_a; const long long* offs_b;
  void* const* addr_a; void* const* addr_b;
  long long br;
  int ta, tb, va, vb;
  int asz, bsz;
} gemm_view;

/* element sizes from the public X-table (the oracle does not link the product library) */
static int tsize(int t) {
  static const unsigned char sizes[] = {
#define ORACLE_X_(NAME, SIZE) SIZE,
    LIBXSMM_DATATYPE_TABLE(ORACLE_X_)
#undef ORACLE_X_
    0 };
  return (t >= 0 && t < (int)LIBXSMM_DATATYPE_COUNT_) ? (int)sizes[t] : 0;
}
/* VNNI pack factor of the *host* the reference runs on (x86: bf16 -> 2) [ref: src/libxsmm_cpuid_x86.c:775] */
enum { ORACLE_BF16_PACK = 2 };

static br_cursor br_at(const gemm_view* v, long long r) {
  br_cursor c;
  const unsigned int f = v->d->flags;
  if (f & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS) {        /* :181-185 */
    c.a = (const char*)v->addr_a[r]; c.b = (const char*)v->addr_b[r];
  } else if (f & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET) {  /* :186-188 byte offsets */
    c.a = v->a0 + v->offs_a[r]; c.b = v->b0 + v->offs_b[r];
  } else if (f & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE) {  /* :189-191 */
    c.a = v->a0 + v->d->br_stride_a * r; c.b = v->b0 + v->d->br_stride_b * r;
  } else {
    c.a = v->a0; c.b = v->b0;
  }
  return c;
}

/* element index of A(i, s) / B(s, j) for the flag combination; kb = VNNI pack factor (1 or 2) */
static long long a_index(const gemm_view* v, int i, int s, int kb) {
  const long long lda = v->d->lda;
  if (!v->ta) return (long long)(s / kb) * (lda * kb) + (long long)i * kb + (s % kb);   /* :2149 (kb=1: :1378) */
  return (long long)i * lda + s;                                                          /* :2151, :1391 */
}
static long long b_index(const gemm_view* v, int s, int j, int kb) {
  const long long ldb = v->d->ldb;
  if (v->tb && v->vb) return (long long)j * kb + (long long)(s / kb) * (ldb * kb) + (s % kb); /* :2157 */
  if (v->tb) return (long long)s * ldb + j;                                                      /* :2159, :1403 */
  return (long long)j * ldb + s;                                                                 /* :2161, :1379 */
}

static float load_f32(const char* base, long long idx, int type) {
  if (type == LIBXSMM_DATATYPE_F32) return ((const float*)base)[idx];
  if (type == LIBXSMM_DATATYPE_BF32) return oracle_bf16_to_f32(oracle_f32_to_bf16_rne(((const float*)base)[idx]));   /* f32 storage, bf16 precision [ref: :1366,:1384-1389] */
  if (type == LIBXSMM_DATATYPE_BF8) return oracle_bf8_to_f32(((const unsigned char*)base)[idx]);
  if (type == LIBXSMM_DATATYPE_HF8) return oracle_hf8_to_f32(((const unsigned char*)base)[idx]);
  return oracle_bf16_to_f32(((const unsigned short*)base)[idx]);
}

/* C (or the f32 scratch) accumulation for f32 / bf16 inputs. acc points at an m x n f32 image with
 * leading dimension ldacc which already holds the start value (0, C, bias, bias + C). */
static void contract_f32(const gemm_view* v, float* acc, long long ldacc) {
  const oracle_gemm_desc* d = v->d;
  const int kb = ((d->a_type == LIBXSMM_DATATYPE_BF16 || ((d->a_type == LIBXSMM_DATATYPE_BF8 || d->a_type == LIBXSMM_DATATYPE_HF8) && d->b_type == LIBXSMM_DATATYPE_BF16)) && v->va)
               ? ORACLE_BF16_PACK : 1;     /* 8-bit float weights x bf16 activations run the bf16 loop with A decoded from a byte [ref: :2171-2366] */
  int i, j, s, k2; long long r;
  for (j = 0; j < d->n; ++j) {
    for (i = 0; i < d->m; ++i) {
      float c = acc[j * ldacc + i];
      for (r = 0; r < v->br; ++r) {
        const br_cursor cur = br_at(v, r);
        for (s = 0; s < d->k / kb; ++s) {
          for (k2 = kb - 1; k2 >= 0; --k2) {               /* high half of the pair first */
            const int kk = s * kb + k2;
            const float av = load_f32(cur.a, a_index(v, i, kk, kb), d->a_type);
            const float bv = load_f32(cur.b, b_index(v, kk, j, kb), d->b_type);
            const float prod = av * bv;
            c = c + prod;
          }
        }
      }
      acc[j * ldacc + i] = c;
    }
  }
}

/* 8-bit float GEMM (BF8 = E5M2, HF8 = E4M3), f32 accumulate and output; A VNNI-4 under VNNI_A, k consumed in ascending
 * order inside a quad [ref: :2420-2470 (BF8), :2471-2510 (HF8)] */
static int is_fp8(int t) { return t == LIBXSMM_DATATYPE_BF8 || t == LIBXSMM_DATATYPE_HF8; }
static float load_fp8(const char* base, long long idx, int type) {
  const unsigned char x = ((const unsigned char*)base)[idx];
  return type == LIBXSMM_DATATYPE_BF8 ? oracle_bf8_to_f32(x) : oracle_hf8_to_f32(x);
}
static void contract_fp8(const gemm_view* v, void* cptr, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int kb = v->va ? 4 : 1;
  const int c_f32 = (d->c_type == LIBXSMM_DATATYPE_F32);          /* else C has the operands' 8-bit type [ref: :2511-2619] */
  float* cmat = (float*)cptr; unsigned char* c8 = (unsigned char*)cptr;
  int i, j, s; long long r;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    float c = beta0 ? 0.0f : (c_f32 ? cmat[(long long)j * d->ldc + i] : load_fp8((const char*)cptr, (long long)j * d->ldc + i, d->c_type));
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (s = 0; s < d->k; ++s) {
        const float prod = load_fp8(cur.a, a_index(v, i, s, kb), d->a_type) * load_fp8(cur.b, b_index(v, s, j, kb), d->b_type);
        c = c + prod;
      }
    }
    if (c_f32) cmat[(long long)j * d->ldc + i] = c;
    else c8[(long long)j * d->ldc + i] = d->c_type == LIBXSMM_DATATYPE_BF8 ? oracle_f32_to_bf8_rne(c) : oracle_f32_to_hf8_rne(c);
  }
}

/* 16-bit integers -> i32, A optionally VNNI-2 [ref: :1427-1450] */
static void contract_i16(const gemm_view* v, int* cmat, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int kb = v->va ? 2 : 1;
  int i, j, s; long long r;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    int acc = beta0 ? 0 : cmat[(long long)j * d->ldc + i];
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (s = 0; s < d->k; ++s)
        acc += (int)((const short*)cur.a)[(long long)(s / kb) * ((long long)d->lda * kb) + (long long)i * kb + (s % kb)] * (int)((const short*)cur.b)[(long long)j * d->ldb + s];
    }
    cmat[(long long)j * d->ldc + i] = acc;
  }
}

/* 8-bit integer weights with one f32 scale per row (a.tertiary) x bf16 activations -> bf16 / f32 [ref: :1684-1730]: the scaled weight is
 * rounded to bf16, products are summed from 0 in k order, beta * C is added after the sum */
static void contract_i8_bf16(const gemm_view* v, const libxsmm_gemm_param* p, void* cptr, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const float* scf = (const float*)p->a.tertiary;
  int i, j, s; long long r;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    float acc = 0.0f;
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (s = 0; s < d->k; ++s) {
        float a_use = (float)(int)((const signed char*)cur.a)[(long long)s * d->lda + i];
        a_use = a_use * scf[i];
        a_use = oracle_bf16_to_f32(oracle_f32_to_bf16_rne(a_use));
        { const float prod = a_use * oracle_bf16_to_f32(((const unsigned short*)cur.b)[(long long)j * d->ldb + s]); acc = acc + prod; }
      }
    }
    if (d->c_type == LIBXSMM_DATATYPE_BF16) {
      unsigned short* c = (unsigned short*)cptr + (long long)j * d->ldc + i;
      if (!beta0) acc = acc + oracle_bf16_to_f32(*c);
      *c = oracle_f32_to_bf16_rne(acc);
    } else {
      float* c = (float*)cptr + (long long)j * d->ldc + i;
      if (!beta0) acc = acc + *c;
      *c = acc;
    }
  }
}

/* F16 x F16 -> F16 or F32, f32 accumulation [ref: :2025-2124].  Unlike bf16: the pair of a VNNI-2 A is consumed LOW k first, the start
 * value is 0 and beta * C is added AFTER the sum, and an f32 C is rounded to f16 on the way in.  (comp_type F16: the running sum is rounded to f16 after
 * every product; IMPLICIT means that on AVX512-FP16 hosts only -- restated, like the device library, as f32.) */
static void contract_f16(const gemm_view* v, void* cmat, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int kb = v->va ? 2 : 1;
  int i, j, s; long long r;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    float c = 0.0f;
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (s = 0; s < d->k; ++s) {
        const float av = oracle_f16_to_f32(((const unsigned short*)cur.a)[a_index(v, i, s, kb)]);
        const float bv = oracle_f16_to_f32(((const unsigned short*)cur.b)[b_index(v, s, j, kb)]);
        const float prod = av * bv;
        c = c + prod;
        if (d->comp_type == LIBXSMM_DATATYPE_F16) c = oracle_f16_to_f32(oracle_f32_to_f16(c));      /* comp F16: the running sum lives in a half [ref: :2042,:2059-2062] */
      }
    }
    if (d->c_type == LIBXSMM_DATATYPE_F32) {
      float* cf = (float*)cmat + (long long)j * d->ldc + i;
      if (!beta0) c = c + oracle_f16_to_f32(oracle_f32_to_f16(*cf));
      *cf = c;
    } else {
      unsigned short* ch = (unsigned short*)cmat + (long long)j * d->ldc + i;
      if (!beta0) c = c + oracle_f16_to_f32(*ch);
      *ch = oracle_f32_to_f16(c);
    }
  }
}

static void contract_f64(const gemm_view* v, double* cmat) {
  const oracle_gemm_desc* d = v->d;
  int i, j, s; long long r;
  for (j = 0; j < d->n; ++j) {
    for (i = 0; i < d->m; ++i) {
      double c = (d->flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 0.0 : cmat[(long long)j * d->ldc + i];
      for (r = 0; r < v->br; ++r) {
        const br_cursor cur = br_at(v, r);
        for (s = 0; s < d->k; ++s) {
          const double prod = ((const double*)cur.a)[a_index(v, i, s, 1)] * ((const double*)cur.b)[b_index(v, s, j, 1)];
          c = c + prod;
        }
      }
      cmat[(long long)j * d->ldc + i] = c;
    }
  }
}

static void setup_view(gemm_view* v, const void* param, const oracle_gemm_desc* d) {
  /* both param flavours start with {op, a, b, c}: the ext struct only appends slots */
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  memset(v, 0, sizeof(*v));
  v->d = d;
  v->ta = (d->flags & LIBXSMM_GEMM_FLAG_TRANS_A) ? 1 : 0;
  v->tb = (d->flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? 1 : 0;
  v->va = (d->flags & LIBXSMM_GEMM_FLAG_VNNI_A) ? 1 : 0;
  v->vb = (d->flags & LIBXSMM_GEMM_FLAG_VNNI_B) ? 1 : 0;
  v->asz = tsize(d->a_type); v->bsz = tsize(d->b_type);
  v->a0 = (const char*)p->a.primary; v->b0 = (const char*)p->b.primary;
  v->br = 1;
  if (d->flags & (LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE)) {
    v->br = (long long)*(const unsigned long long*)p->op.tertiary;        /* :490-492 */
  }
  if (d->flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS) {                /* :494-498 */
    v->addr_a = (void* const*)p->a.primary; v->addr_b = (void* const*)p->b.primary;
  } else if (d->flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET) {          /* :509-513 */
    v->offs_a = (const long long*)p->a.secondary; v->offs_b = (const long long*)p->b.secondary;
  }
}

static float act_apply(int act, float x) {
  if (act == 1 || act == 2) return (x <= 0.0f) ? 0.0f : x;                /* mateltwise ref :2148 */
  if (act == 3) return (tanhf(x / 2.0f) + 1.0f) / 2.0f;                   /* mateltwise ref :18-20 */
  return x;
}

/* NORM -> VNNI2 of a 16-bit m x n matrix, in place through a copy [ref: mateltwise ref :532-557] */
static void c_to_vnni2_16bit(unsigned short* c, int m, int n, int ldc) {
  const long long nn = n + (n % 2);
  unsigned short* tmp = (unsigned short*)malloc(sizeof(unsigned short) * (size_t)ldc * (size_t)nn);
  long long i, j, j2;
  memset(tmp, 0, sizeof(unsigned short) * (size_t)ldc * (size_t)nn);
  memcpy(tmp, c, sizeof(unsigned short) * (size_t)ldc * (size_t)n);
  for (i = 0; i < (long long)ldc * nn; ++i) c[i] = 0;
  for (j = 0; j < nn / 2; ++j) for (i = 0; i < m; ++i) for (j2 = 0; j2 < 2; ++j2) {
    c[j * ldc * 2 + i * 2 + j2] = tmp[(j * 2 + j2) * ldc + i];
  }
  free(tmp);
}

/* NORM -> VNNI4 of an 8-bit m x n matrix, in place through a copy [ref: mateltwise ref :737-759] */
static void c_to_vnni4_08bit(unsigned char* c, int m, int n, int ldc) {
  const long long nn = ((n % 4) == 0) ? n : (n + 4 - n % 4);
  unsigned char* tmp = (unsigned char*)malloc((size_t)ldc * (size_t)nn);
  long long i, j, j2;
  memset(tmp, 0, (size_t)ldc * (size_t)nn);
  memcpy(tmp, c, (size_t)ldc * (size_t)n);
  for (i = 0; i < (long long)ldc * nn; ++i) c[i] = 0;
  for (j = 0; j < nn / 4; ++j) for (i = 0; i < m; ++i) for (j2 = 0; j2 < 4; ++j2) {
    c[j * ldc * 4 + i * 4 + j2] = tmp[(j * 4 + j2) * ldc + i];
  }
  free(tmp);
}

/* 8-bit integer GEMM: A and B i8/u8 (signedness in the datatype), i32 accumulation over all (r, k).
 * A is VNNI-4 [k/4][lda][4] under VNNI_A (flat [k][lda] otherwise, i32 output only), B flat [n][ldb].
 * C i32: C = beta*C + sum [ref: :1452-1555]; C f32: C = float(sum) * scf (+ C), scf = *(float*)c.tertiary,
 * A always VNNI-4 [ref: :1556-1683, scf :591-592]. */
static int is_int8(int t) { return t == LIBXSMM_DATATYPE_I8 || t == LIBXSMM_DATATYPE_U8; }
static void contract_int8(const gemm_view* v, const libxsmm_gemm_param* p, void* cptr, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int ua = (d->a_type == LIBXSMM_DATATYPE_U8), ub = (d->b_type == LIBXSMM_DATATYPE_U8);
  const int c_f32 = (d->c_type == LIBXSMM_DATATYPE_F32);
  const long long kb = (c_f32 || v->va) ? 4 : 1;
  const float scf = c_f32 ? *(const float*)p->c.tertiary : 1.0f;
  long long i, j, r, s, k2;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    int acc = 0;
    if (!c_f32 && !beta0) acc = ((int*)cptr)[j * d->ldc + i];
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (s = 0; s < d->k / kb; ++s) for (k2 = 0; k2 < kb; ++k2) {
        const long long ai = s * ((long long)d->lda * kb) + i * kb + k2, bi = j * (long long)d->ldb + s * kb + k2;
        const int av = ua ? (int)((const unsigned char*)cur.a)[ai] : (int)((const signed char*)cur.a)[ai];
        const int bv = ub ? (int)((const unsigned char*)cur.b)[bi] : (int)((const signed char*)cur.b)[bi];
        acc += av * bv;
      }
    }
    if (c_f32) {
      float f = (float)acc;
      f *= scf;
      if (!beta0) f += ((float*)cptr)[j * d->ldc + i];
      ((float*)cptr)[j * d->ldc + i] = f;
    } else ((int*)cptr)[j * d->ldc + i] = acc;
  }
}

/* 1-bit and 2-bit weights times 8-bit activations -> i32 [ref: gemm ref :1100-1300].
 * I1X8: a byte holds four k of two rows -- bit (k % 4) of the low nibble for the even row, of the high nibble for the odd row, 0 = +1, 1 = -1 --
 *       at (k / 4) * lda / 2 + i / 2.
 * I2X4 (interleaved): the m rows form four groups of m / 4; the byte at (k / 4) * lda + 4 * (i % (m / 4)) + k % 4 holds that k of row
 *       i % (m / 4) of every group, group g in bits 2 g, 2 g + 1: 0 -> 0, 1 -> +1, 2 and 3 -> -1. */
static int is_lowbit_a(int t) { return t == LIBXSMM_DATATYPE_I1X8 || t == LIBXSMM_DATATYPE_I2X4; }
static void contract_lowbit(const gemm_view* v, void* cptr, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int ub = (d->b_type == LIBXSMM_DATATYPE_U8), one_bit = (d->a_type == LIBXSMM_DATATYPE_I1X8);
  const long long mq = d->m / 4;
  long long i, j, r, kk;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    int acc = beta0 ? 0 : ((int*)cptr)[j * d->ldc + i];
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (kk = 0; kk < d->k; ++kk) {
        const long long bi = j * (long long)d->ldb + kk;
        const int bv = ub ? (int)((const unsigned char*)cur.b)[bi] : (int)((const signed char*)cur.b)[bi];
        int w;
        if (one_bit) {
          const unsigned int byte = ((const unsigned char*)cur.a)[((kk / 4) * (long long)d->lda) / 2 + i / 2];
          w = ((byte >> (4 * (i & 1) + (kk & 3))) & 1u) ? -1 : 1;
        } else {
          const unsigned int byte = ((const unsigned char*)cur.a)[(kk / 4) * (long long)d->lda + 4 * (i % mq) + (kk & 3)];
          const unsigned int code = (byte >> (2 * (i / mq))) & 3u;
          w = (code == 0) ? 0 : (code == 1) ? 1 : -1;
        }
        acc += w * bv;
      }
    }
    ((int*)cptr)[j * d->ldc + i] = acc;
  }
}

/* 4-bit weights in the INTERLEAVED layout times 8-bit activations [ref: gemm ref :1009-1088 (MXFP4), :1272-1330 (I4X2)]: a dword holds eight k of
 * one row, [k/8][lda][4 bytes]; byte b carries k = 8 o + b in its low and k = 8 o + 4 + b in its high nibble.  No batch-reduce or stride mode.
 * I4X2 x u8 -> i32: weight = nibble - zero point of the row (a.quaternary, one byte per row and block); B is read as UNSIGNED bytes.
 * MXFP4 x i8 -> f32 / bf16: E2M1 codes become integers through the table {0, 11, 21, 32, 42, 64, 85, 127} (sign in bit 3), the integer sum of a
 *   32-deep block is scaled by 2^(sa - 127) (a.tertiary, [k/32][lda] bytes) and by an f32 per (column, block) (b.tertiary, [n][ldb/32]). */
static float e8m0(unsigned char s);
static const signed char kFp4AsInt[16] = {0, 11, 21, 32, 42, 64, 85, 127, 0, -11, -21, -32, -42, -64, -85, -127};
static void contract_i4_intlv(const gemm_view* v, const libxsmm_gemm_param* p, void* cptr, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int mx = (d->a_type == LIBXSMM_DATATYPE_MXFP4X2);
  const long long lda = d->lda, ldb = d->ldb, k = d->k;
  long long i, j, r, s, kk;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    int iacc = (mx || beta0) ? 0 : ((int*)cptr)[j * d->ldc + i];
    float facc = 0.0f;
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      const unsigned char* a = (const unsigned char*)cur.a;
      if (!mx) {
        const int zpt = ((const unsigned char*)p->a.quaternary)[((d->br_stride_a * 2) / k) * r + i];
        for (kk = 0; kk < k; ++kk) {
          const unsigned int byte = a[(kk / 8) * lda * 4 + i * 4 + (kk & 3)];
          const int w = (int)((kk & 4) ? (byte >> 4) : (byte & 15u)) - zpt;
          iacc += (int)(signed char)w * (int)((const unsigned char*)cur.b)[j * ldb + kk];
        }
      } else {
        for (s = 0; s < k / 32; ++s) {
          const float sca = e8m0(((const unsigned char*)p->a.tertiary)[((d->br_stride_a * 2) / 32) * r + s * lda + i]);
          const float scb = ((const float*)p->b.tertiary)[(d->br_stride_b / 32) * r + j * (ldb / 32) + s];
          int tmp = 0;
          for (kk = 32 * s; kk < 32 * s + 32; ++kk) {
            const unsigned int byte = a[(kk / 8) * lda * 4 + i * 4 + (kk & 3)];
            tmp += (int)kFp4AsInt[(kk & 4) ? (byte >> 4) : (byte & 15u)] * (int)((const signed char*)cur.b)[j * ldb + kk];
          }
          { float t2 = (float)tmp * sca; t2 = t2 * scb; facc = facc + t2; }
        }
      }
    }
    if (!mx) ((int*)cptr)[j * d->ldc + i] = iacc;
    else if (d->c_type == LIBXSMM_DATATYPE_F32) { float* c = (float*)cptr + j * d->ldc + i; *c = (beta0 ? 0.0f : *c) + facc; }
    else { unsigned short* c = (unsigned short*)cptr + j * d->ldc + i; *c = oracle_f32_to_bf16_rne((beta0 ? 0.0f : oracle_bf16_to_f32(*c)) + facc); }
  }
}

/* MXFP4 weights: A = packed E2M1 pairs [k/2][lda] bytes (low nibble = even k) with one E8M0 scale per (32-deep k-block, row)
 * in a.tertiary ([k/32][lda] bytes; per batch-reduce element: pointer array / offset*2/32 / stride*2/32); B bf16 or f32 flat;
 * C f32 or bf16; C = (beta ? C : 0) + sum, one RNE for bf16 C [ref: :949-1008, scale :200-222, LUT :60-64, slots :565-569].
 * A scale byte of 0 decodes to 0.0f (bits << 23), not 2^-127, exactly like the reference. */
static float mxfp4_value(unsigned char x) {
  static const float lut[16] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f, -0.0f, -0.5f, -1.0f, -1.5f, -2.0f, -3.0f, -4.0f, -6.0f};
  return lut[x & 15];
}
static float mxfp4_scale(const gemm_view* v, const libxsmm_gemm_param* p, long long r, long long s, long long i) {
  const oracle_gemm_desc* d = v->d;
  const unsigned char* base = (const unsigned char*)p->a.tertiary;
  union { unsigned int u; float f; } cv;
  if (d->flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS) base = ((const unsigned char* const*)p->a.tertiary)[r];
  else if (d->flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET) base += (v->offs_a[r] * 2) / 32;
  else if (d->flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE) base += ((d->br_stride_a * 2) / 32) * r;
  cv.u = ((unsigned int)base[s * d->lda + i]) << 23;
  return cv.f;
}
static void contract_mxfp4(const gemm_view* v, const libxsmm_gemm_param* p, void* cptr, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int b_bf16 = (d->b_type == LIBXSMM_DATATYPE_BF16), c_f32 = (d->c_type == LIBXSMM_DATATYPE_F32);
  long long i, j, r, s, k2;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    float acc = 0.0f;
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (s = 0; s < d->k / 32; ++s) {
        const float scf = mxfp4_scale(v, p, r, s, i);
        for (k2 = 0; k2 < 32; k2 += 2) {
          const unsigned char pk = ((const unsigned char*)cur.a)[(s * 32 + k2) * (long long)d->lda / 2 + i];
          const float ev = mxfp4_value(pk & 15) * scf, od = mxfp4_value(pk >> 4) * scf;
          const long long bi = j * (long long)d->ldb + s * 32 + k2;
          const float b0 = b_bf16 ? oracle_bf16_to_f32(((const unsigned short*)cur.b)[bi]) : ((const float*)cur.b)[bi];
          const float b1 = b_bf16 ? oracle_bf16_to_f32(((const unsigned short*)cur.b)[bi + 1]) : ((const float*)cur.b)[bi + 1];
          float prod = ev * b0; acc = acc + prod;
          prod = od * b1; acc = acc + prod;
        }
      }
    }
    if (c_f32) { float* c = (float*)cptr + j * d->ldc + i; float base = beta0 ? 0.0f : *c; *c = base + acc; }
    else {
      unsigned short* c = (unsigned short*)cptr + j * d->ldc + i;
      float base = beta0 ? 0.0f : oracle_bf16_to_f32(*c);
      base = base + acc;
      *c = oracle_f32_to_bf16_rne(base);
    }
  }
}

/* MX x MX GEMM (OCP microscaling, E8M0 scale per 32 k and row): A and B in the SAME k-grouped layout -- a dword per (row, k-group):
 * MXFP8 (MXBF8 = E5M2, MXHF8 = E4M3) [k/4][ld][4 bytes], MXFP4 [k/8][ld][4 bytes = 8 nibbles, low nibble first]; B is "VNNI and
 * transposed", i.e. indexed by the column j with ldb >= n.  Scales: a.tertiary [k/32][lda], b.tertiary [k/32][ldb].  C f32 =
 * (beta ? C : 0) + sum over (r, 32-blocks) of partial * scale_a * scale_b, where fp8 adds the 4 products of a k-group high k
 * first and fp4 adds 8 products ascending; batch-reduce elements are contiguous blocks (r * ld * k elements), the reference's
 * MX x MX path knows no other addressing [ref: gemm ref :2620-2665 (fp8), :2731-2785 (fp4), :836-845]. */
static int is_mxmx(const oracle_gemm_desc* d) {
  return d->b_type == d->a_type && (d->a_type == LIBXSMM_DATATYPE_MXFP4X2 || d->a_type == LIBXSMM_DATATYPE_MXBF8 || d->a_type == LIBXSMM_DATATYPE_MXHF8 ||
                                    d->a_type == LIBXSMM_DATATYPE_MXBF6 || d->a_type == LIBXSMM_DATATYPE_MXHF6);
}
/* 6-bit floats of the MX formats (no infinities, no NaNs): E2M3 ("HF6": bias 1, subnormals m / 8) and E3M2 ("BF6": bias 3, subnormals m / 16).
 * The reference goes through E4M3 with two look-up tables [ref: gemm ref :70-92]; every 6-bit value is exact there, so this is the same number. */
static float fp6_value(unsigned int v, int e3m2) {
  const unsigned int sign = (v >> 5) & 1u;
  const unsigned int e = e3m2 ? ((v >> 2) & 7u) : ((v >> 3) & 3u), m = e3m2 ? (v & 3u) : (v & 7u);
  float mag;
  if (e3m2) mag = (e == 0) ? (float)m * 0.0625f : (1.0f + (float)m * 0.25f) * (float)(1u << e) * 0.125f;
  else mag = (e == 0) ? (float)m * 0.125f : (1.0f + (float)m * 0.125f) * (float)(1u << e) * 0.5f;
  return sign ? -mag : mag;
}
static float e8m0(unsigned char s) { union { unsigned int u; float f; } cv; cv.u = ((unsigned int)s) << 23; return cv.f; }
static void contract_mxmx(const gemm_view* v, const libxsmm_gemm_param* p, float* cmat, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const unsigned char* a = (const unsigned char*)p->a.primary; const unsigned char* b = (const unsigned char*)p->b.primary;
  const unsigned char* sa = (const unsigned char*)p->a.tertiary; const unsigned char* sb = (const unsigned char*)p->b.tertiary;
  const long long lda = d->lda, ldb = d->ldb, k = d->k;
  const int fp4 = (d->a_type == LIBXSMM_DATATYPE_MXFP4X2), hf8 = (d->a_type == LIBXSMM_DATATYPE_MXHF8);
  long long i, j, r, s, g, k2;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    float acc = 0.0f;
    float* c = cmat + j * d->ldc + i;
    if (beta0) *c = 0.0f;
    for (r = 0; r < v->br; ++r) {
      if (fp4) {
        for (s = 0; s < k / 32; ++s) {
          const float sca = e8m0(sa[r * lda * (k / 32) + s * lda + i]), scb = e8m0(sb[r * ldb * (k / 32) + s * ldb + j]);
          for (g = 0; g < 4; ++g) {
            float tmp = 0.0f;
            for (k2 = 0; k2 < 8; ++k2) {
              const long long kblk = s * 4 + g, byte = k2 / 2;
              const unsigned char ba = a[r * lda * (k / 2) + kblk * lda * 4 + i * 4 + byte], bb = b[r * ldb * (k / 2) + kblk * ldb * 4 + j * 4 + byte];
              const float prod = mxfp4_value((k2 & 1) ? (ba >> 4) : (ba & 15)) * mxfp4_value((k2 & 1) ? (bb >> 4) : (bb & 15));
              tmp = tmp + prod;
            }
            { float t2 = tmp * sca; t2 = t2 * scb; acc = acc + t2; }
          }
        }
      } else if (d->a_type == LIBXSMM_DATATYPE_MXBF6 || d->a_type == LIBXSMM_DATATYPE_MXHF6) {
        /* four 6-bit values of a row's k-group in three bytes, [k/4][ld][3]; a batch-reduce element is (ld * 6 / 8) * k bytes [ref: :2680-2727] */
        const int e3m2 = (d->a_type == LIBXSMM_DATATYPE_MXBF6);
        const long long slab_a = ((lda * 6) / 8) * k, slab_b = ((ldb * 6) / 8) * k;
        for (s = 0; s < k / 4; ++s) {
          const float sca = e8m0(sa[r * lda * (k / 32) + (s / 8) * lda + i]), scb = e8m0(sb[r * ldb * (k / 32) + (s / 8) * ldb + j]);
          const unsigned char* pa = a + r * slab_a + s * lda * 3 + i * 3; const unsigned char* pb = b + r * slab_b + s * ldb * 3 + j * 3;
          const unsigned int va = (unsigned int)pa[0] | ((unsigned int)pa[1] << 8) | ((unsigned int)pa[2] << 16);
          const unsigned int vb = (unsigned int)pb[0] | ((unsigned int)pb[1] << 8) | ((unsigned int)pb[2] << 16);
          float tmp = 0.0f;
          for (k2 = 3; k2 >= 0; --k2) {
            const float prod = fp6_value((va >> (6 * k2)) & 0x3f, e3m2) * fp6_value((vb >> (6 * k2)) & 0x3f, e3m2);
            tmp = tmp + prod;
          }
          { float t2 = tmp * sca; t2 = t2 * scb; acc = acc + t2; }
        }
      } else {
        for (s = 0; s < k / 4; ++s) {
          const float sca = e8m0(sa[r * lda * (k / 32) + (s / 8) * lda + i]), scb = e8m0(sb[r * ldb * (k / 32) + (s / 8) * ldb + j]);
          float tmp = 0.0f;
          for (k2 = 3; k2 >= 0; --k2) {
            const unsigned char ba = a[r * lda * k + s * lda * 4 + i * 4 + k2], bb = b[r * ldb * k + s * ldb * 4 + j * 4 + k2];
            const float prod = (hf8 ? oracle_hf8_to_f32(ba) : oracle_bf8_to_f32(ba)) * (hf8 ? oracle_hf8_to_f32(bb) : oracle_bf8_to_f32(bb));
            tmp = tmp + prod;
          }
          { float t2 = tmp * sca; t2 = t2 * scb; acc = acc + t2; }
        }
      }
    }
    *c = *c + acc;
  }
}

/* MX-typed C of an MX x MX GEMM [ref: gemm ref :661-817]: 32 consecutive rows of a column share one E8M0 scale.  The reference mimics its JIT,
 * which works in bf16: inputs rounded to bf16, shared exponent = exponent(amax) - emax_elem (2 for E2M1, 15 for E5M2) clamped to 0..254 (NaN
 * sticks in amax), scale and reciprocal are powers of two passed through bf16, every scaled value is rounded to bf16 again and then encoded
 * (E2M1 by thresholds with the sign of the input; E5M2 by RNE, infinities and NaNs saturate to the largest normal). */
static unsigned char e2m1_abs_code(float a) {
  return (a != a) ? 7 : (a > 5.0f) ? 7 : (a >= 3.5f) ? 6 : (a > 2.5f) ? 5 : (a >= 1.75f) ? 4 : (a > 1.25f) ? 3 : (a >= 0.75f) ? 2 : (a > 0.25f) ? 1 : 0;
}
static void mx_out_block(const float* in, unsigned char* out, unsigned char* out_scale, int fp4) {
  union { float f; unsigned int u; } cv;
  float x[32], amax = 0.0f, scale, rcp;
  int i, e;
  for (i = 0; i < 32; ++i) x[i] = oracle_bf16_to_f32(oracle_f32_to_bf16_rne(in[i]));
  for (i = 0; i < 32; ++i) { const float a = fabsf(x[i]); if (a > amax || a != a) amax = a; }
  cv.f = amax;
  e = (amax == 0.0f) ? 0 : (int)((cv.u >> 23) & 0xffu);
  e -= fp4 ? 2 : 15;
  if (e < 0) e = 0;
  if (e > 254) e = 254;
  *out_scale = (unsigned char)e;
  cv.u = ((unsigned int)e << 23) | (e == 0 ? (1u << 22) : 0u);
  scale = oracle_bf16_to_f32(oracle_f32_to_bf16_rne(cv.f));
  rcp = 1.0f / scale;
  rcp = oracle_bf16_to_f32(oracle_f32_to_bf16_rne(rcp));
  for (i = 0; i < 32; ++i) {
    float v = x[i] * rcp;
    v = oracle_bf16_to_f32(oracle_f32_to_bf16_rne(v));
    if (fp4) {
      cv.f = x[i];
      { const unsigned char code = (unsigned char)(((cv.u >> 31) ? 8u : 0u) | e2m1_abs_code(fabsf(v)));
        if (i & 1) out[i / 2] = (unsigned char)((out[i / 2] & 0x0f) | (code << 4)); else out[i / 2] = code; }
    } else {
      unsigned char b = oracle_f32_to_bf8_rne(v);
      if ((b & 0x7c) == 0x7c) b = (unsigned char)((b & 0x80) | 0x7b);
      out[i] = b;
    }
  }
}

/* A compressed by bitmask (LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK) [ref: generator_gemm_reference_impl.c:857-948]: a.primary holds only the
 * non-zeros, in the order k-group s, row i, k inside the group (= memory order of the VNNI image for 16-bit types); a.secondary holds one bit
 * per element, row s of the bit matrix being m * kb bits (bit order: LSB first, mateltwise ref :170-178).  Products are added in that order,
 * k ascending inside a pair -- NOT the high-half-first order of the dense 16-bit loop. */
static void contract_spmm(const oracle_gemm_desc* d, const libxsmm_gemm_param* p, int beta0) {
  const int m = d->m, n = d->n, k = d->k;
  const int kb = (d->a_type == LIBXSMM_DATATYPE_F32) ? 1 : ORACLE_BF16_PACK;
  const unsigned char* bitmap = (const unsigned char*)p->a.secondary;
  const int c_f32 = (d->c_type == LIBXSMM_DATATYPE_F32);
  float* scratch = c_f32 ? NULL : (float*)malloc(sizeof(float) * (size_t)m * (size_t)n);
  unsigned long long at = 0;
  int i, j, s, k2;
  for (s = 0; s < k / kb; ++s) for (i = 0; i < m; ++i) {
    if (s == 0) for (j = 0; j < n; ++j) {
      if (c_f32) { if (beta0) ((float*)p->c.primary)[(long long)j * d->ldc + i] = 0.0f; }
      else scratch[(long long)j * m + i] = beta0 ? 0.0f : (d->c_type == LIBXSMM_DATATYPE_BF16 ? oracle_bf16_to_f32(((const unsigned short*)p->c.primary)[(long long)j * d->ldc + i])
                                                                                              : oracle_f16_to_f32(((const unsigned short*)p->c.primary)[(long long)j * d->ldc + i]));
    }
    for (k2 = 0; k2 < kb; ++k2) {
      const long long q = (long long)i * kb + k2;
      if ((bitmap[q / 8 + (long long)s * ((long long)m * kb / 8)] >> (q % 8)) & 1) {
        const float av = d->a_type == LIBXSMM_DATATYPE_F32 ? ((const float*)p->a.primary)[at]
                       : d->a_type == LIBXSMM_DATATYPE_BF16 ? oracle_bf16_to_f32(((const unsigned short*)p->a.primary)[at]) : oracle_f16_to_f32(((const unsigned short*)p->a.primary)[at]);
        for (j = 0; j < n; ++j) {
          const long long bi = (long long)j * d->ldb + (long long)s * kb + k2;
          const float bv = d->b_type == LIBXSMM_DATATYPE_F32 ? ((const float*)p->b.primary)[bi]
                         : d->b_type == LIBXSMM_DATATYPE_BF16 ? oracle_bf16_to_f32(((const unsigned short*)p->b.primary)[bi]) : oracle_f16_to_f32(((const unsigned short*)p->b.primary)[bi]);
          const float prod = av * bv;
          float* c = c_f32 ? (float*)p->c.primary + (long long)j * d->ldc + i : scratch + (long long)j * m + i;
          *c = *c + prod;
        }
        ++at;
      }
    }
  }
  if (!c_f32) {
    for (i = 0; i < m; ++i) for (j = 0; j < n; ++j) {
      const float y = scratch[(long long)j * m + i];
      ((unsigned short*)p->c.primary)[(long long)j * d->ldc + i] = d->c_type == LIBXSMM_DATATYPE_BF16 ? oracle_f32_to_bf16_rne(y) : oracle_f32_to_f16(y);
    }
    free(scratch);
  }
}

/* Fused column bias / ReLU (+ bitmask) / sigmoid on IEEE-half and 8-bit-float GEMMs [ref: :294-372, :2826-2839].  The reference swaps C for an f32 scratch when C is
 * not f32 [:255-266], fills it with the bias (of C's type) [+ C] [:296-317] or with C alone when beta = 1 [:320-329], runs the matmul of the (A, B) pair with c_type = F32
 * and beta = 1 on that image -- so the F16 loop adds the start value AFTER the sum and rounds it to a half on the way in [:2112-2117], the 8-bit float loops start from
 * it [:2432-2434, :2483-2485] -- and hands the image to the unary TPP that applies the activation and converts to C's type [:335-370, mateltwise ref :303-322]. */
static void contract_fused_lowp(const gemm_view* v, const libxsmm_gemm_ext_param* pe, int beta0) {
  const oracle_gemm_desc* d = v->d;
  const int f16 = (d->a_type == LIBXSMM_DATATYPE_F16);
  const int kb = v->va ? (f16 ? 2 : 4) : 1;
  const long long mask_ld = LIBXSMM_UPDIV(d->ldc, 16) * 16;             /* mateltwise ref :2142 */
  unsigned char* mask = (d->act == 2) ? (unsigned char*)pe->c.secondary : NULL;
  char* cptr = (char*)pe->c.primary;
  const int have_start = d->colbias || !beta0;
  int i, j, s; long long r;
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    const long long ci = (long long)j * d->ldc + i;
    float start = 0.0f, c, y;
    if (!beta0) start = (d->c_type == LIBXSMM_DATATYPE_F16) ? oracle_f16_to_f32(((const unsigned short*)cptr)[ci]) : load_f32(cptr, ci, d->c_type);
    if (d->colbias) {
      const float bias = (d->c_type == LIBXSMM_DATATYPE_F16) ? oracle_f16_to_f32(((const unsigned short*)pe->d.primary)[i]) : load_f32((const char*)pe->d.primary, i, d->c_type);   /* D has C's type */
      start = beta0 ? bias : (bias + start);
    }
    c = (f16 || !have_start) ? 0.0f : start;
    for (r = 0; r < v->br; ++r) {
      const br_cursor cur = br_at(v, r);
      for (s = 0; s < d->k; ++s) {
        float prod;
        if (f16) {
          prod = oracle_f16_to_f32(((const unsigned short*)cur.a)[a_index(v, i, s, kb)]) * oracle_f16_to_f32(((const unsigned short*)cur.b)[b_index(v, s, j, kb)]);
          c = c + prod;
          if (d->comp_type == LIBXSMM_DATATYPE_F16) c = oracle_f16_to_f32(oracle_f32_to_f16(c));
        } else {
          prod = load_fp8(cur.a, a_index(v, i, s, kb), d->a_type) * load_fp8(cur.b, b_index(v, s, j, kb), d->b_type);
          c = c + prod;
        }
      }
    }
    if (f16 && have_start) c = c + oracle_f16_to_f32(oracle_f32_to_f16(start));
    y = act_apply(d->act, c);
    if (d->c_type == LIBXSMM_DATATYPE_F32) ((float*)cptr)[ci] = y;
    else if (d->c_type == LIBXSMM_DATATYPE_F16) ((unsigned short*)cptr)[ci] = oracle_f32_to_f16(y);
    else ((unsigned char*)cptr)[ci] = d->c_type == LIBXSMM_DATATYPE_BF8 ? oracle_f32_to_bf8_rne(y) : oracle_f32_to_hf8_rne(y);
    if (mask) {
      unsigned char* byte = mask + i / 8 + j * (mask_ld / 8);
      const unsigned char bit = (unsigned char)(1u << (i % 8));
      if (c <= 0.0f) *byte = (unsigned char)(*byte & ~bit); else *byte = (unsigned char)(*byte | bit);
    }
  }
}

static void oracle_gemm_plain(const void* param, const oracle_gemm_desc* d);
/* libxsmm_reference_gemm [:2817-2850]: the contraction, then -- VNNI_C -- the finished C re-laid through the NORM_TO_VNNI TPP [:2802-2815]: VNNI-2 for 16-bit results
 * (bf16 inside oracle_gemm_plain; IEEE halves here), VNNI-4 for results of an 8-bit float type. */
void oracle_gemm(const void* param, const oracle_gemm_desc* d) {
  oracle_gemm_plain(param, d);
  if (((d->flags & LIBXSMM_GEMM_FLAG_NO_RESET_TILECONFIG) != 0) != ((d->flags & LIBXSMM_GEMM_FLAG_NO_SETUP_TILECONFIG) != 0)) return;
  if (d->flags & LIBXSMM_GEMM_FLAG_VNNI_C) {
    void* cptr = ((const libxsmm_gemm_param*)param)->c.primary;
    if (d->c_type == LIBXSMM_DATATYPE_F16) c_to_vnni2_16bit((unsigned short*)cptr, d->m, d->n, d->ldc);
    else if (d->c_type == LIBXSMM_DATATYPE_BF8 || d->c_type == LIBXSMM_DATATYPE_HF8) c_to_vnni4_08bit((unsigned char*)cptr, d->m, d->n, d->ldc);
  }
}
static void oracle_gemm_plain(const void* param, const oracle_gemm_desc* d) {
  gemm_view v;
  const int is_ext = (d->flags & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI) ? 1 : 0;
  const int beta0 = (d->flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 1 : 0;
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  const libxsmm_gemm_ext_param* pe = (const libxsmm_gemm_ext_param*)param;
  void* cptr = p->c.primary;
  int i, j;
  /* tile-config pseudo kernels are no-ops  [:2821-2826] */
  if (((d->flags & LIBXSMM_GEMM_FLAG_NO_RESET_TILECONFIG) != 0) != ((d->flags & LIBXSMM_GEMM_FLAG_NO_SETUP_TILECONFIG) != 0)) return;
  setup_view(&v, param, d);

  if (d->flags & LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK) { contract_spmm(d, p, beta0); return; }
  if (d->a_type == LIBXSMM_DATATYPE_F64) { contract_f64(&v, (double*)cptr); return; }
  if (is_int8(d->a_type) && is_int8(d->b_type)) { contract_int8(&v, p, cptr, beta0); return; }
  if (is_lowbit_a(d->a_type) && is_int8(d->b_type) && d->c_type == LIBXSMM_DATATYPE_I32) { contract_lowbit(&v, cptr, beta0); return; }
  if ((d->flags & LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT) && is_int8(d->b_type) && (d->a_type == LIBXSMM_DATATYPE_I4X2 || d->a_type == LIBXSMM_DATATYPE_U4X2 || d->a_type == LIBXSMM_DATATYPE_MXFP4X2)) {
    contract_i4_intlv(&v, p, cptr, beta0); return;
  }
  if (is_ext && (d->colbias || d->act) &&
      ((is_fp8(d->a_type) && d->b_type == d->a_type && (d->c_type == LIBXSMM_DATATYPE_F32 || d->c_type == d->a_type)) ||
       (d->a_type == LIBXSMM_DATATYPE_F16 && d->b_type == LIBXSMM_DATATYPE_F16 && (d->c_type == LIBXSMM_DATATYPE_F16 || d->c_type == LIBXSMM_DATATYPE_F32)))) {
    contract_fused_lowp(&v, pe, beta0); return;
  }
  if (is_fp8(d->a_type) && d->b_type == d->a_type && (d->c_type == LIBXSMM_DATATYPE_F32 || d->c_type == d->a_type)) { contract_fp8(&v, cptr, beta0); return; }
  if (d->a_type == LIBXSMM_DATATYPE_I16 && d->b_type == LIBXSMM_DATATYPE_I16 && d->c_type == LIBXSMM_DATATYPE_I32) { contract_i16(&v, (int*)cptr, beta0); return; }
  if (d->a_type == LIBXSMM_DATATYPE_I8 && d->b_type == LIBXSMM_DATATYPE_BF16) { contract_i8_bf16(&v, p, cptr, beta0); return; }
  if (d->a_type == LIBXSMM_DATATYPE_F16 && d->b_type == LIBXSMM_DATATYPE_F16 && (d->c_type == LIBXSMM_DATATYPE_F16 || d->c_type == LIBXSMM_DATATYPE_F32)) {
    contract_f16(&v, cptr, beta0); return;
  }
  if (is_mxmx(d) && d->c_type == LIBXSMM_DATATYPE_F32) { contract_mxmx(&v, p, (float*)cptr, beta0); return; }
  if (is_mxmx(d) && d->c_type == d->a_type && (d->a_type == LIBXSMM_DATATYPE_MXFP4X2 || d->a_type == LIBXSMM_DATATYPE_MXBF8) && beta0) {
    /* C of the operands' MX type: the f32 result is quantised block by block, data to c.primary, scales to c.tertiary [ref: :2666-2678, :2787-2798] */
    const int fp4 = (d->a_type == LIBXSMM_DATATYPE_MXFP4X2);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)d->ldc * (size_t)d->n);
    oracle_gemm_desc df = *d;
    gemm_view vf;
    df.c_type = LIBXSMM_DATATYPE_F32;
    setup_view(&vf, param, &df);
    contract_mxmx(&vf, p, tmp, 1);
    for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; i += 32)
      mx_out_block(tmp + (long long)j * d->ldc + i, (unsigned char*)cptr + (fp4 ? ((long long)j * (d->ldc / 2) + i / 2) : ((long long)j * d->ldc + i)),
                   (unsigned char*)p->c.tertiary + (long long)j * (d->ldc / 32) + i / 32, fp4);
    free(tmp);
    return;
  }
  if (d->a_type == LIBXSMM_DATATYPE_MXFP4X2) { contract_mxfp4(&v, p, cptr, beta0); return; }

  {
    /* f32 working image: C itself for f32 output, otherwise a scratch of ldc x n floats [:262-272] */
    const int c_is_f32 = (d->c_type == LIBXSMM_DATATYPE_F32);
    const long long ldacc = d->ldc;
    float* acc = c_is_f32 ? (float*)cptr : (float*)malloc(sizeof(float) * (size_t)d->ldc * (size_t)d->n);
    const int colbias = is_ext ? d->colbias : 0;
    const int act = is_ext ? d->act : 0;
    /* start value  [:294-332] */
    for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
      float start = 0.0f;
      if (!beta0) start = load_f32((const char*)cptr, (long long)j * d->ldc + i, d->c_type);
      if (colbias) {
        const float bias = load_f32((const char*)pe->d.primary, i, d->c_type);   /* D has C's type */
        start = beta0 ? bias : (bias + start);
      }
      acc[j * ldacc + i] = start;
    }
    contract_f32(&v, acc, ldacc);
    /* post-op + store  [:335-372] */
    if (act != 0 || !c_is_f32) {
      const long long mask_ld = LIBXSMM_UPDIV(d->ldc, 16) * 16;             /* mateltwise ref :2142 */
      unsigned char* mask = (act == 2) ? (unsigned char*)pe->c.secondary : NULL;
      for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
        const float x = acc[j * ldacc + i];
        const float y = act_apply(act, x);
        if (c_is_f32) ((float*)cptr)[(long long)j * d->ldc + i] = y;
        else ((unsigned short*)cptr)[(long long)j * d->ldc + i] = oracle_f32_to_bf16_rne(y);
        if (mask) {
          unsigned char* byte = mask + i / 8 + j * (mask_ld / 8);
          const unsigned char bit = (unsigned char)(1u << (i % 8));
          if (x <= 0.0f) *byte = (unsigned char)(*byte & ~bit); else *byte = (unsigned char)(*byte | bit);
        }
      }
    }
    if (!c_is_f32) free(acc);
    if ((d->flags & LIBXSMM_GEMM_FLAG_VNNI_C) && d->c_type == LIBXSMM_DATATYPE_BF16) {
      c_to_vnni2_16bit((unsigned short*)cptr, d->m, d->n, d->ldc);          /* :2802-2815 */
    }
  }
}

/* k-ordered fmaf chain: what a v_mfma_f32_32x32x2_f32 accumulation computes when k is
 * consumed in natural order; used to cross-check kernels more tightly than the tolerance. */
void oracle_gemm_f32_fma(const void* param, const oracle_gemm_desc* d) {
  gemm_view v; int i, j, s; long long r;
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  float* c = (float*)p->c.primary;
  setup_view(&v, param, d);
  for (j = 0; j < d->n; ++j) for (i = 0; i < d->m; ++i) {
    float acc = (d->flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 0.0f : c[(long long)j * d->ldc + i];
    for (r = 0; r < v.br; ++r) {
      const br_cursor cur = br_at(&v, r);
      for (s = 0; s < d->k; ++s) {
        acc = fmaf(((const float*)cur.a)[a_index(&v, i, s, 1)], ((const float*)cur.b)[b_index(&v, s, j, 1)], acc);
      }
    }
    c[(long long)j * d->ldc + i] = acc;
  }
}
