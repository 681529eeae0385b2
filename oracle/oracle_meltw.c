/*
 * oracle_meltw.c -- CPU restatement of the reference's element-wise TPP semantics (test-only).
 *
 * Follows  src/generator_mateltwise_reference_impl.c  of the reference:
 *   :241-272   operand indexing under ROW / COL / SCALAR broadcast
 *   :274-324   load-as-f32 and store-with-rounding per datatype
 *   :83-130    f32 unary math; :181-214 f32 binary math
 *   :376-1062  layout transforms (transpose, NORM<->VNNI, padding)
 *   :1065-1441 reductions, :1444-1790 gather / scatter
 *   :2074-2502 unary driver (ReLU family with bitmask, UNZIP, ...), :2505-2593 binary,
 *   :2596-2660 ternary
 * Scope: the operations the GPU library implements (F32 / BF16 / F64 data, 8/16/32/64-bit
 * payloads for pure data movement).  Unsupported combinations abort loudly.
 */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_DIE(msg) do { fprintf(stderr, "oracle_meltw: %s (line %d)\n", msg, __LINE__); abort(); } while (0)

static int tsz(int t) {
  static const unsigned char sizes[] = {
#define ORACLE_X_(NAME, SIZE) SIZE,
    LIBXSMM_DATATYPE_TABLE(ORACLE_X_)
#undef ORACLE_X_
    0 };
  return (t >= 0 && t < (int)LIBXSMM_DATATYPE_COUNT_) ? (int)sizes[t] : 0;
}

/* which broadcast applies to operand `op` (0..2) of a unary/binary/ternary kernel  [:241-260] */
enum { BC_NONE = 0, BC_ROW = 1, BC_COL = 2, BC_SCALAR = 3 };
static int bcast_kind(const oracle_meltw_desc* d, int op) {
  const unsigned int f = d->flags;
  if (d->operation == LIBXSMM_MELTW_OPERATION_UNARY) {
    if (op != 0) return BC_NONE;
    if (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_ROW) return BC_ROW;
    if ((f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_COL) || d->type == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR) return BC_COL;
    if (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_SCALAR) return BC_SCALAR;
  } else if (d->operation == LIBXSMM_MELTW_OPERATION_BINARY) {
    if (op > 1) return BC_NONE;
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_ROW_IN_0 << op)) return BC_ROW;
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0 << op)) return BC_COL;
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_SCALAR_IN_0 << op)) return BC_SCALAR;
  } else if (d->operation == LIBXSMM_MELTW_OPERATION_TERNARY) {
    if (op > 2) return BC_NONE;
    if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_ROW_IN_0 << op)) return BC_ROW;
    if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_COL_IN_0 << op)) return BC_COL;
    if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_SCALAR_IN_0 << op)) return BC_SCALAR;
  }
  return BC_NONE;
}
static long long elem_index(int kind, long long i, long long j, long long ld) {   /* [:262-270] */
  switch (kind) { case BC_ROW: return j * ld; case BC_COL: return i; case BC_SCALAR: return 0; default: return i + j * ld; }
}
static float get_f32(const void* p, long long idx, int type) {                     /* [:274-297] */
  if (type == LIBXSMM_DATATYPE_F32) return ((const float*)p)[idx];
  if (type == LIBXSMM_DATATYPE_BF16) return oracle_bf16_to_f32(((const unsigned short*)p)[idx]);
  if (type == LIBXSMM_DATATYPE_F16) return oracle_f16_to_f32(((const unsigned short*)p)[idx]);
  if (type == LIBXSMM_DATATYPE_BF8) return oracle_bf8_to_f32(((const unsigned char*)p)[idx]);
  if (type == LIBXSMM_DATATYPE_HF8) return oracle_hf8_to_f32(((const unsigned char*)p)[idx]);
  ORACLE_DIE("unsupported input datatype"); return 0.0f;
}
static void put_f32(void* p, long long idx, int type, float v) {                   /* [:299-324] */
  if (type == LIBXSMM_DATATYPE_F32) ((float*)p)[idx] = v;
  else if (type == LIBXSMM_DATATYPE_BF16) ((unsigned short*)p)[idx] = oracle_f32_to_bf16_rne(v);
  else if (type == LIBXSMM_DATATYPE_F16) ((unsigned short*)p)[idx] = oracle_f32_to_f16(v);
  else if (type == LIBXSMM_DATATYPE_BF8) ((unsigned char*)p)[idx] = oracle_f32_to_bf8_rne(v);
  else if (type == LIBXSMM_DATATYPE_HF8) ((unsigned char*)p)[idx] = oracle_f32_to_hf8_rne(v);
  else ORACLE_DIE("unsupported output datatype");
}
static void bit_put(unsigned char* bits, long long i, long long j, long long ld_bits, int on) {   /* [:150-166] */
  unsigned char* byte = bits + i / 8 + j * (ld_bits / 8);
  const unsigned char m = (unsigned char)(1u << (i % 8));
  *byte = on ? (unsigned char)(*byte | m) : (unsigned char)(*byte & ~m);
}
static int bit_get(const unsigned char* bits, long long i, long long j, long long ld_bits) {     /* [:168-177] */
  return (bits[i / 8 + j * (ld_bits / 8)] >> (i % 8)) & 1;
}

/* f32 -> BF8 with stochastic rounding [ref: src/libxsmm_lpflt_quant.c:303-365]: the value goes through f16, a normal number gets a random
 * byte added below the kept bits (one xoshiro128++ draw of stream `lane` of the 16-stream state), subnormals round to nearest even,
 * infinities stay, NaNs are quieted.  Element number e of a TPP call uses stream e % 16 [mateltwise ref :2095, :2485-2486]. */
static unsigned char f32_to_bf8_stochastic(float x, unsigned int* st, unsigned int lane) {
  unsigned short h = oracle_f32_to_f16(x);
  unsigned int s0 = st[lane], s1 = st[lane + 16], s2 = st[lane + 32], s3 = st[lane + 48], t0;
  const unsigned int sum = s0 + s3, vrng = ((sum << 7) | (sum >> 25)) + s0;
  const unsigned short rnd = (unsigned short)((vrng >> 24) & 0xffu), fixup = (unsigned short)((h >> 8) & 1u);
  t0 = s1 << 9; s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t0; s3 = (s3 << 11) | (s3 >> 21);
  st[lane] = s0; st[lane + 16] = s1; st[lane + 32] = s2; st[lane + 48] = s3;
  if ((h & 0x7c00u) == 0x7c00u) h = ((h & 0x03ffu) == 0) ? h : (unsigned short)(h | 0x0200u);
  else if ((h & 0x7c00u) == 0) h = (unsigned short)(h + 0x007fu + fixup);
  else h = (unsigned short)(h + rnd);
  return (unsigned char)(h >> 8);
}
/* store of the generic element-wise loops: stochastic only for BF8 output with the flag set [:299-320] */
static void put_elem(void* out, long long idx, int type, float v, int stoch, void* state, unsigned int seed_idx) {
  if (stoch && type == LIBXSMM_DATATYPE_BF8) ((unsigned char*)out)[idx] = f32_to_bf8_stochastic(v, (unsigned int*)state, seed_idx % 16u);
  else put_f32(out, idx, type, v);
}

static float sigmoidf_ref(float x) { return (tanhf(x / 2.0f) + 1.0f) / 2.0f; }     /* [:18-20] */
static float unary_f32(int type, float x) {                                       /* [:83-127] */
  switch (type) {
    case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: case LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR: return x;
    case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: return -1.0f * x;
    case LIBXSMM_MELTW_TYPE_UNARY_X2: return x * x;
    case LIBXSMM_MELTW_TYPE_UNARY_XOR: return 0.0f;
    case LIBXSMM_MELTW_TYPE_UNARY_TANH: return tanhf(x);
    case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID: return sigmoidf_ref(x);
    case LIBXSMM_MELTW_TYPE_UNARY_GELU: return (erff(x / sqrtf(2.0f)) + 1.0f) * 0.5f * x;
    case LIBXSMM_MELTW_TYPE_UNARY_GELU_INV:
      return (0.5f + 0.5f * erff(x / sqrtf(2.0f)) + x / (sqrtf(2.0f * (float)M_PI)) * expf(-0.5f * x * x));
    case LIBXSMM_MELTW_TYPE_UNARY_TANH_INV: return 1.0f - tanhf(x) * tanhf(x);
    case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID_INV: return sigmoidf_ref(x) * (1.0f - sigmoidf_ref(x));
    case LIBXSMM_MELTW_TYPE_UNARY_SQRT: return sqrtf(x);
    case LIBXSMM_MELTW_TYPE_UNARY_INC: return x + 1.0f;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: return 1.0f / x;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: return 1.0f / sqrtf(x);
    case LIBXSMM_MELTW_TYPE_UNARY_EXP: return expf(x);
    default: ORACLE_DIE("unsupported unary op"); return 0.0f;
  }
}
static double unary_f64(int type, double x) {                                     /* [:129-152] */
  switch (type) {
    case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: return x;
    case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: return -1.0 * x;
    case LIBXSMM_MELTW_TYPE_UNARY_X2: return x * x;
    case LIBXSMM_MELTW_TYPE_UNARY_XOR: return 0.0;
    case LIBXSMM_MELTW_TYPE_UNARY_SQRT: return sqrt(x);
    case LIBXSMM_MELTW_TYPE_UNARY_INC: return x + 1.0;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: return 1.0 / x;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: return 1.0 / sqrt(x);
    default: ORACLE_DIE("unsupported f64 unary op"); return 0.0;
  }
}

/* ---- layout transforms: pure data movement on 1/2/4/8-byte payloads ----------------- */
static void move_elem(void* out, long long oi, const void* in, long long ii, int sz) {
  memcpy((char*)out + oi * sz, (const char*)in + ii * sz, (size_t)sz);
}
static int is_transform(int t) {
  switch (t) {
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_NORMT: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2_PAD: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4_PAD: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8_PAD: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI2_TO_VNNI2T:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_VNNI4T: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI8_TO_VNNI8T:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2T: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4T:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8T: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI2T_TO_NORM:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4T_TO_NORM: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI8T_TO_NORM:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_NORM: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_VNNI2:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADM_MOD2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADN_MOD2:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADNM_MOD2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADM_MOD4:
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADN_MOD4: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADNM_MOD4: return 1;
    default: return 0;
  }
}
static void transform(const libxsmm_meltw_unary_param* p, const oracle_meltw_desc* d) {
  const long long M = d->m, N = d->n, ldi = d->ldi, ldo = d->ldo;
  const int sz = tsz(d->in0_type);
  const void* in = p->in.primary; void* out = p->out.primary;
  long long i, j, i2, j2, v = 0, pad_n = 0;
  switch (d->type) {
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_NORMT:                     /* [:376-424] */
      for (i = 0; i < N; ++i) for (j = 0; j < M; ++j) move_elem(out, j * ldo + i, in, i * ldi + j, sz);
      return;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2_PAD: v = 2; break;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4_PAD: v = 4; break;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8_PAD: v = 8; break;
    default: break;
  }
  if (v != 0) {   /* NORM -> VNNI<v>: zero-fill ldo * roundup(N, v) elements first  [:532-557, :686-760] */
    const long long Nn = LIBXSMM_UP(N, v);
    memset(out, 0, (size_t)(ldo * Nn * sz));
    for (j = 0; j < Nn / v; ++j) for (i = 0; i < M; ++i) for (j2 = 0; j2 < v; ++j2) {
      if (j * v + j2 < N) move_elem(out, j * ldo * v + i * v + j2, in, (j * v + j2) * ldi + i, sz);
      /* rows beyond N read unspecified input in the reference; here they stay zero */
    }
    return;
  }
  switch (d->type) {
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI2_TO_VNNI2T: v = 2; break;      /* [:427-446] */
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_VNNI4T: v = 4; break;      /* [:449-470] */
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI8_TO_VNNI8T: v = 8; break;      /* [:493-512] */
    default: break;
  }
  if (v != 0) {
    for (j = 0; j < M / v; ++j) for (i = 0; i < N / v; ++i) for (j2 = 0; j2 < v; ++j2) for (i2 = 0; i2 < v; ++i2) {
      move_elem(out, j * ldo * v + j2 + (i * v + i2) * v, in, i * ldi * v + i2 + (j * v + j2) * v, sz);
    }
    return;
  }
  switch (d->type) {
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2T: v = 2; break;       /* [:560-578] */
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4T: v = 4; break;       /* [:643-661] */
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8T: v = 8; break;       /* [:664-683] */
    default: break;
  }
  if (v != 0) {
    for (i = 0; i < M / v; ++i) for (j = 0; j < N; ++j) for (i2 = 0; i2 < v; ++i2) {
      move_elem(out, i * ldo * v + j * v + i2, in, j * ldi + i * v + i2, sz);
    }
    return;
  }
  switch (d->type) {
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI2T_TO_NORM: v = 2; break;       /* [:623-640] */
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4T_TO_NORM: v = 4; break;       /* [:602-620] */
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI8T_TO_NORM: v = 8; break;       /* [:581-599] */
    default: break;
  }
  if (v != 0) {   /* note the swapped roles of m and n in the reference */
    const long long Mm = d->n, Nn = d->m;
    for (i = 0; i < Mm / v; ++i) for (j = 0; j < Nn; ++j) for (i2 = 0; i2 < v; ++i2) {
      move_elem(out, j * ldo + i * v + i2, in, i * ldi * v + j * v + i2, sz);
    }
    return;
  }
  if (d->type == LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_NORM) {            /* [:788-804] */
    for (i = 0; i < N; ++i) for (j = 0; j < M; ++j) move_elem(out, i * ldo + j, in, (i / 4) * ldi * 4 + j * 4 + (i % 4), sz);
    return;
  }
  if (d->type == LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_VNNI2) {           /* [:807-823] */
    for (i = 0; i < N; ++i) for (j = 0; j < M; ++j) move_elem(out, (i / 2) * ldo * 2 + j * 2 + (i % 2), in, (i / 4) * ldi * 4 + j * 4 + (i % 4), sz);
    return;
  }
  /* padding copies  [:826-964]: zero ldo x N' then copy the m x n block */
  switch (d->type) {
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADN_MOD2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADNM_MOD2: pad_n = LIBXSMM_UP(N, 2); break;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADN_MOD4: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADNM_MOD4: pad_n = LIBXSMM_UP(N, 4); break;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADM_MOD2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADM_MOD4: pad_n = N; break;
    default: ORACLE_DIE("unsupported transform");
  }
  memset(out, 0, (size_t)(ldo * pad_n * sz));
  for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) move_elem(out, j * ldo + i, in, j * ldi + i, sz);
}

/* ---- gather / scatter  [:1444-1790] ------------------------------------------------------ */
static void gather_scatter(const libxsmm_meltw_unary_param* p, const oracle_meltw_desc* d) {
  const long long m = d->m, n = d->n, ldi = d->ldi, ldo = d->ldo;
  const int sz = tsz(d->in0_type);
  const int is_gather = (d->type == LIBXSMM_MELTW_TYPE_UNARY_GATHER);
  const int idx64 = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_8BYTES) ? 1 : 0;
  const void* idxp = is_gather ? p->in.secondary : p->out.secondary;
  const void* in = p->in.primary; void* out = p->out.primary;
  long long a, b;
#define IDX(q) (idx64 ? (long long)((const unsigned long long*)idxp)[q] : (long long)((const unsigned int*)idxp)[q])
  if (d->flags & LIBXSMM_MELTW_FLAG_UNARY_GS_COLS) {
    for (a = 0; a < n; ++a) for (b = 0; b < m; ++b) {
      if (is_gather) move_elem(out, b + a * ldo, in, b + IDX(a) * ldi, sz);
      else move_elem(out, b + IDX(a) * ldo, in, b + a * ldi, sz);
    }
  } else if (d->flags & LIBXSMM_MELTW_FLAG_UNARY_GS_ROWS) {
    for (a = 0; a < m; ++a) for (b = 0; b < n; ++b) {
      if (is_gather) move_elem(out, a + b * ldo, in, IDX(a) + b * ldi, sz);
      else move_elem(out, IDX(a) + b * ldo, in, a + b * ldi, sz);
    }
  } else {   /* GS_OFFS: per-element linear offsets */
    for (b = 0; b < n; ++b) for (a = 0; a < m; ++a) {
      if (is_gather) move_elem(out, a + b * ldo, in, IDX(a + b * m), sz);
      else move_elem(out, IDX(a + b * m), in, a + b * ldi, sz);
    }
  }
#undef IDX
}

/* rows per draw of the DROPOUT generator (see there) */
static long long oracle_rng_width = 16;
void oracle_set_rng_width(int w) { oracle_rng_width = (w >= 1 && w <= 16) ? w : 16; }

/* ---- reductions  [:1065-1441], f32 compute ------------------------------------------------ */
static int is_reduce(int t) {
  switch (t) {
    case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD: case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD:
    case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD: case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX:
    case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN: case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX: return 1;
    default: return 0;
  }
}
/* Reduction over a LIST of columns (REDUCE_COLS_IDX_OP_ADD / MAX / MIN: out[i] = op_jj in(i, idx[jj]), jj < *in.tertiary) and the
 * column reductions MAX / ABSMAX / MIN that record WHERE the extremum was found (REDUCE_RECORD_ARGOP: the column index goes to
 * out.secondary; a later equal value wins) [:1346-1430].  Indices are 4 bytes with IDX_SIZE_4BYTES, else 8 [:1088]. */
static int is_reduce_cols_idx(int t) {
  return t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_ADD || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MAX || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MIN;
}
static void reduce_cols_listed(const libxsmm_meltw_unary_param* p, const oracle_meltw_desc* d) {
  const long long m = d->m, ldi = d->ldi;
  const int idx4 = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_4BYTES) ? 1 : 0;
  const int record = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_RECORD_ARGOP) ? 1 : 0;
  const int listed = is_reduce_cols_idx(d->type);
  const unsigned long long n_cols = listed ? *(const unsigned long long*)p->in.tertiary : (unsigned long long)d->n;
  const unsigned int* idx32 = (const unsigned int*)p->in.secondary; const unsigned long long* idx64 = (const unsigned long long*)p->in.secondary;
  unsigned int* arg32 = (unsigned int*)p->out.secondary; unsigned long long* arg64 = (unsigned long long*)p->out.secondary;
  const int op = (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_ADD) ? 0
               : (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MAX || d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) ? 1
               : (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX) ? 3 : 2;
  long long i; unsigned long long jj;
  for (i = 0; i < m; ++i) {
    float acc = (op == 0) ? 0.0f : (op == 1) ? -FLT_MAX : (op == 3) ? 0.0f : FLT_MAX;
    for (jj = 0; jj < n_cols; ++jj) {
      const unsigned long long j = listed ? (idx4 ? (unsigned long long)idx32[jj] : idx64[jj]) : jj;
      float x = get_f32(p->in.primary, i + (long long)j * ldi, d->in0_type);
      if (op == 0) { acc = acc + x; continue; }
      if (op == 3) x = fabsf(x);
      if (op == 1 || op == 3) {
        if (record) { if (x >= acc) { acc = x; if (idx4) arg32[i] = (unsigned int)j; else arg64[i] = j; } }
        else acc = (x < acc) ? acc : x;                                      /* LIBXSMM_MAX(in_val, acc) */
      } else {
        if (record) { if (x <= acc) { acc = x; if (idx4) arg32[i] = (unsigned int)j; else arg64[i] = j; } }
        else acc = (x < acc) ? x : acc;                                      /* LIBXSMM_MIN(in_val, acc) */
      }
    }
    put_f32(p->out.primary, i, d->out_type, acc);
  }
}
static void reduce(const libxsmm_meltw_unary_param* p, const oracle_meltw_desc* d) {
  const long long m = d->m, n = d->n, ldi = d->ldi;
  const int rows = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) ? 1 : 0;   /* 1: collapse i, result per column */
  const long long count = rows ? n : m, inner = rows ? m : n;
  const long long result_size = rows ? n : d->ldo;                              /* [:1073] */
  const int init_acc = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_INIT_ACC) ? 1 : 0;
  const int want_x = (d->type != LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD);
  const int want_x2 = (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD || d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD);
  const int is_add = (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD || want_x2);
  void* out_x = p->out.primary;
  void* out_x2 = (want_x && want_x2) ? (void*)((char*)p->out.primary + result_size * tsz(d->out_type)) : p->out.primary;
  long long q, t;
  for (q = 0; q < count; ++q) {
    float sx = 0.0f, sx2 = 0.0f;
    if (!is_add) {
      if (rows) sx = get_f32(p->in.primary, 0 + q * ldi, d->in0_type);                         /* [:1369] */
      else sx = (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) ? -FLT_MAX
              : (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN) ? FLT_MAX : 0.0f;         /* [:1386,:1415] */
    }
    for (t = 0; t < inner; ++t) {
      const long long i = rows ? t : q, j = rows ? q : t;
      float x = get_f32(p->in.primary, i + j * ldi, d->in0_type);
      if (is_add) { sx = sx + x; sx2 = sx2 + x * x; }
      else if (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) sx = (sx < x) ? x : sx;
      else if (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN) sx = (sx > x) ? x : sx;
      else { const float ax = fabsf(x), as = fabsf(sx); sx = (as < ax) ? ax : as; }
    }
    if (is_add && init_acc) {                                                                   /* [:1313-1324] */
      if (want_x) sx = sx + get_f32(out_x, q, d->out_type);
      if (want_x2) sx2 = sx2 + get_f32(out_x2, q, d->out_type);
    }
    if (want_x) put_f32(out_x, q, d->out_type, sx);
    if (want_x2) put_f32(out_x2, q, d->out_type, sx2);
  }
}

/* ---- drivers --------------------------------------------------------------------------------- */
void oracle_meltw_unary(const libxsmm_meltw_unary_param* p, const oracle_meltw_desc* d) {
  const long long M = d->m, ldi = d->ldi, ldo = d->ldo;
  const long long N = (d->type == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR) ? (long long)*(const unsigned long long*)p->op.primary : d->n;
  const int bc = bcast_kind(d, 0);
  long long i, j;
  if (is_reduce_cols_idx(d->type)) { reduce_cols_listed(p, d); return; }
  if (is_reduce(d->type) && (d->flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_RECORD_ARGOP) && !(d->flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) &&
      (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX || d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN || d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX)) {
    reduce_cols_listed(p, d); return;
  }
  if (is_reduce(d->type)) { reduce(p, d); return; }
  if (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_TO_SCALAR_OP_ADD) {                /* [:2097-2116] one serial sum over the whole block, column by column */
    if (d->in0_type == LIBXSMM_DATATYPE_F64 && d->out_type == LIBXSMM_DATATYPE_F64 && d->comp_type == LIBXSMM_DATATYPE_F64) {
      double acc = 0.0;
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) acc += ((const double*)p->in.primary)[elem_index(bc, i, j, ldi)];
      ((double*)p->out.primary)[0] = acc;
    } else {
      float acc = 0.0f;
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) acc += get_f32(p->in.primary, elem_index(bc, i, j, ldi), d->in0_type);
      put_f32(p->out.primary, 0, d->out_type, acc);
    }
    return;
  }
  if (d->type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD_NCNC_FORMAT) {            /* [:2118-2141] blocked [N / bn][C / bc][bn][bc] input: m = bc, n = bn, ldi = C, ldo = N */
    const long long bc_ = d->m, bn = d->n, C = d->ldi, NN = d->ldo;
    long long iC, ic, iN, i_n;
    for (iC = 0; iC < C / bc_; ++iC) for (ic = 0; ic < bc_; ++ic) {
      float tmp = 0.0f;
      for (iN = 0; iN < NN / bn; ++iN) for (i_n = 0; i_n < bn; ++i_n)
        tmp += get_f32(p->in.primary, iN * C * bn + iC * bn * bc_ + i_n * bc_ + ic, d->in0_type);
      put_f32(p->out.primary, iC * bc_ + ic, d->out_type, tmp);
    }
    return;
  }
  if (d->type == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X2 || d->type == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X3) {   /* [:2437-2470] */
    /* f32 -> two / three bf16 that add up to it: the leading pieces by truncation, the last by RNE of the remainder; byte offsets of the pieces in out.secondary */
    const unsigned long long* strides = (const unsigned long long*)p->out.secondary;
    unsigned short* o16 = (unsigned short*)p->out.primary;
    for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
      const float x = ((const float*)p->in.primary)[elem_index(bc, i, j, ldi)];
      unsigned int u; float t, r1;
      memcpy(&u, &x, 4); u &= 0xffff0000u; memcpy(&t, &u, 4);
      o16[j * ldo + i] = (unsigned short)(u >> 16);
      r1 = x - t;
      if (d->type == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X3) {
        float r2;
        memcpy(&u, &r1, 4); u &= 0xffff0000u; memcpy(&t, &u, 4);
        o16[j * ldo + i + (long long)(strides[0] / 2)] = (unsigned short)(u >> 16);
        r2 = r1 - t;
        put_f32(o16, j * ldo + i + (long long)(strides[1] / 2), LIBXSMM_DATATYPE_BF16, r2);
      } else put_f32(o16, j * ldo + i + (long long)(strides[0] / 2), LIBXSMM_DATATYPE_BF16, r1);
    }
    return;
  }
  if (d->type == LIBXSMM_MELTW_TYPE_UNARY_GATHER || d->type == LIBXSMM_MELTW_TYPE_UNARY_SCATTER) { gather_scatter(p, d); return; }
  if (is_transform(d->type)) { transform(p, d); return; }
  switch (d->type) {
    case LIBXSMM_MELTW_TYPE_UNARY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_ELU: {   /* [:2136-2167] */
      const int bitm = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) ? 1 : 0;
      const long long mask_ld = LIBXSMM_UPDIV(ldo, 16) * 16;
      const float alpha = (d->type == LIBXSMM_MELTW_TYPE_UNARY_RELU) ? 1.0f : *(const float*)p->op.primary;
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
        const float x = get_f32(p->in.primary, elem_index(bc, i, j, ldi), d->in0_type);
        float y;
        if (d->type == LIBXSMM_MELTW_TYPE_UNARY_RELU) y = (x <= 0.0f) ? 0.0f : x;
        else if (d->type == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU) y = (x <= 0.0f) ? alpha * x : x;
        else y = (x <= 0.0f) ? alpha * (expf(x) - 1.0f) : x;
        put_f32(p->out.primary, i + j * ldo, d->out_type, y);
        if (bitm) bit_put((unsigned char*)p->out.secondary, i, j, mask_ld, !(x <= 0.0f));
      }
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_DROPOUT: {        /* [:2361-2407]; the generator of [:43-72] */
      /* 16 independent xoshiro128+ streams side by side (state word s of stream l at rng_state[l + 16 s], op.secondary).  The reference
       * draws for `w` rows at a time, w = the 32-bit vector length of the CPU it runs on: row i of column j gets draw number
       * j * ceil(M / w) + i / w of stream i % w.  The width is therefore part of the semantics: oracle_set_rng_width (default 16, the
       * AVX-512 width, which is what the device library implements). */
      const int bitm = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) ? 1 : 0;
      const long long mask_ld = bitm ? LIBXSMM_UPDIV(ldo, 16) * 16 : ldo;
      const float prob = *(const float*)p->op.primary, pn = 1 - prob, pi = 1 / pn;
      unsigned int* st = (unsigned int*)p->op.secondary;
      const long long w = oracle_rng_width;
      float draw[16];
      long long l;
      for (j = 0; j < N; ++j) for (i = 0; i < M; i += w) {
        for (l = 0; l < w; ++l) {                                       /* one step of every stream [:43-72] */
          unsigned int s0 = st[l], s1 = st[l + 16], s2 = st[l + 32], s3 = st[l + 48], t0;
          union { unsigned int u; float f; } r;
          r.u = 0x3f800000u | ((s3 + s0) >> 9);
          draw[l] = r.f - 1.0f;
          t0 = s1 << 9; s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t0; s3 = (s3 << 11) | (s3 >> 21);
          st[l] = s0; st[l + 16] = s1; st[l + 32] = s2; st[l + 48] = s3;
        }
        for (l = 0; l < w && i + l < M; ++l) {
          const float x = get_f32(p->in.primary, elem_index(bc, i + l, j, ldi), d->in0_type);
          const int keep = draw[l] < pn;
          put_f32(p->out.primary, (i + l) + j * ldo, d->out_type, keep ? pi * x : 0.0f);
          if (bitm) bit_put((unsigned char*)p->out.secondary, i + l, j, mask_ld, keep);
        }
      }
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_DROPOUT_INV: {    /* [:2408-2424] */
      const int bitm = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) ? 1 : 0;
      const long long mask_ld = bitm ? LIBXSMM_UPDIV(ldi, 16) * 16 : ldi;
      const float prob = *(const float*)p->op.primary, pn = 1.0f - prob, pi = 1.0f / pn;
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
        const float x = get_f32(p->in.primary, elem_index(bc, i, j, ldi), d->in0_type) * pi;
        put_f32(p->out.primary, i + j * ldo, d->out_type, bit_get((const unsigned char*)p->in.secondary, i, j, mask_ld) ? x : 0.0f);
      }
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_QUANT: {          /* f32 -> i8 / i16 / i32, round to nearest even  [:2195-2240] */
      const float scf = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_NO_SCF_QUANT) ? 1.0f : *(const float*)p->in.secondary;
      const int sat = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_SIGN_SAT_QUANT) ? 1 : 0;
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
        const float x = ((const float*)p->in.primary)[elem_index(bc, i, j, ldi)];
        float t = nearbyintf(x * scf);
        if (d->out_type == LIBXSMM_DATATYPE_I8) {
          if (sat) { if (t < -128) t = -128.0f; if (t > 127) t = 127.0f; ((signed char*)p->out.primary)[i + j * ldo] = (signed char)t; }
          else ((signed char*)p->out.primary)[i + j * ldo] = (signed char)(0x000000ff & (int)t);
        } else if (d->out_type == LIBXSMM_DATATYPE_I16) {
          if (sat) { if (t < -32768) t = -32768.0f; if (t > 32767) t = 32767.0f; ((short*)p->out.primary)[i + j * ldo] = (short)t; }
          else ((short*)p->out.primary)[i + j * ldo] = (short)(0x0000ffff & (int)t);
        } else ((int*)p->out.primary)[i + j * ldo] = (int)t;
      }
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_DEQUANT: {        /* i8 / i16 / i32 -> f32  [:2330-2360] */
      const float scf = (d->flags & LIBXSMM_MELTW_FLAG_UNARY_NO_SCF_QUANT) ? 1.0f : *(const float*)p->in.secondary;
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
        const long long idx = elem_index(bc, i, j, ldi);
        float v;
        if (d->in0_type == LIBXSMM_DATATYPE_I8) v = (float)((const signed char*)p->in.primary)[idx];
        else if (d->in0_type == LIBXSMM_DATATYPE_I16) v = (float)((const short*)p->in.primary)[idx];
        else v = (float)((const int*)p->in.primary)[idx];
        ((float*)p->out.primary)[i + j * ldo] = v * scf;
      }
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_RELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_ELU_INV: {   /* [:2168-2194] */
      const long long mask_ld = LIBXSMM_UPDIV(ldi, 16) * 16;
      const float alpha = (d->type == LIBXSMM_MELTW_TYPE_UNARY_RELU_INV) ? 1.0f : *(const float*)p->op.primary;
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
        const float x = get_f32(p->in.primary, elem_index(bc, i, j, ldi), d->in0_type);
        float y;
        if (d->type == LIBXSMM_MELTW_TYPE_UNARY_ELU_INV) {
          const float fwd = get_f32(p->in.secondary, elem_index(bc, i, j, ldi), d->in0_type);
          y = (fwd > 0) ? x : x * (fwd + alpha);
        } else {
          const int bit = bit_get((const unsigned char*)p->in.secondary, i, j, mask_ld);
          y = (d->type == LIBXSMM_MELTW_TYPE_UNARY_RELU_INV) ? (bit ? x : 0.0f) : (bit ? x : alpha * x);
        }
        put_f32(p->out.primary, i + j * ldo, d->out_type, y);
      }
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_UNZIP: {                                         /* [:2419-2432] */
      const unsigned long long offset = *(const unsigned long long*)p->out.secondary;
      unsigned short* lo = (unsigned short*)p->out.primary;
      unsigned short* hi = (unsigned short*)((char*)p->out.primary + offset);
      for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
        unsigned int u; const float f = ((const float*)p->in.primary)[elem_index(bc, i, j, ldi)];
        memcpy(&u, &f, 4);
        lo[j * ldo + i] = (unsigned short)(u & 0xffffu); hi[j * ldo + i] = (unsigned short)(u >> 16);
      }
      return;
    }
    default: break;
  }
  for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {                                 /* [:2469-2497] */
    if (d->in0_type == LIBXSMM_DATATYPE_F64 && d->out_type == LIBXSMM_DATATYPE_F64) {
      ((double*)p->out.primary)[i + j * ldo] = unary_f64(d->type, ((const double*)p->in.primary)[elem_index(bc, i, j, ldi)]);
    } else {
      const float x = get_f32(p->in.primary, elem_index(bc, i, j, ldi), d->in0_type);
      put_elem(p->out.primary, i + j * ldo, d->out_type, unary_f32(d->type, x), (d->flags & LIBXSMM_MELTW_FLAG_UNARY_STOCHASTIC_ROUND) != 0, p->op.secondary, (unsigned int)(j * M + i));
    }
  }
}

static float binary_f32(int type, float a, float b, float out) {                    /* [:181-214] */
  switch (type) {
    case LIBXSMM_MELTW_TYPE_BINARY_ADD: return a + b;
    case LIBXSMM_MELTW_TYPE_BINARY_SUB: return a - b;
    case LIBXSMM_MELTW_TYPE_BINARY_MUL: return a * b;
    case LIBXSMM_MELTW_TYPE_BINARY_DIV: return a / b;
    case LIBXSMM_MELTW_TYPE_BINARY_MULADD: { const float prod = a * b; return out + prod; }
    case LIBXSMM_MELTW_TYPE_BINARY_MAX: return (a > b) ? a : b;
    case LIBXSMM_MELTW_TYPE_BINARY_MIN: return (a > b) ? b : a;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GT: return (a > b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GE: return (a >= b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_LT: return (a < b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_LE: return (a <= b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_EQ: return (a == b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_NE: return (a != b) ? 1.0f : 0.0f;
    default: ORACLE_DIE("unsupported binary op"); return 0.0f;
  }
}
static int is_cmp(int t) { return t >= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GT && t <= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_NE; }

void oracle_meltw_binary(const libxsmm_meltw_binary_param* p, const oracle_meltw_desc* d) {
  const long long M = d->m, N = d->n, ldi = d->ldi, ldi1 = d->ldi2, ldo = d->ldo;
  const int bc0 = bcast_kind(d, 0), bc1 = bcast_kind(d, 1);
  long long i, j;
  if (d->type == LIBXSMM_MELTW_TYPE_BINARY_ZIP) {                                    /* [:2543-2556] */
    for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
      const unsigned int lo = ((const unsigned short*)p->in0.primary)[elem_index(bc0, i, j, ldi)];
      const unsigned int hi = ((const unsigned short*)p->in1.primary)[elem_index(bc1, i, j, ldi1)];
      const unsigned int u = lo | (hi << 16); float f; memcpy(&f, &u, 4);
      ((float*)p->out.primary)[j * ldo + i] = f;
    }
    return;
  }
  if (d->type == LIBXSMM_MELTW_TYPE_BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD) {           /* a dot product into out[0], serial f32 sum [:2523-2542] */
    float acc = 0.0f;
    for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
      const float prod = get_f32(p->in0.primary, elem_index(bc0, i, j, ldi), d->in0_type) * get_f32(p->in1.primary, elem_index(bc1, i, j, ldi1), d->in1_type);
      acc = acc + prod;
    }
    put_f32(p->out.primary, 0, d->out_type, acc);
    return;
  }
  for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {                                   /* [:2558-2590] */
    if (d->in0_type == LIBXSMM_DATATYPE_F64 && d->out_type == LIBXSMM_DATATYPE_F64) {
      const double a = ((const double*)p->in0.primary)[elem_index(bc0, i, j, ldi)];
      const double b = ((const double*)p->in1.primary)[elem_index(bc1, i, j, ldi1)];
      double* o = (double*)p->out.primary + i + j * ldo; double prod;
      switch (d->type) {
        case LIBXSMM_MELTW_TYPE_BINARY_ADD: *o = a + b; break;
        case LIBXSMM_MELTW_TYPE_BINARY_SUB: *o = a - b; break;
        case LIBXSMM_MELTW_TYPE_BINARY_MUL: *o = a * b; break;
        case LIBXSMM_MELTW_TYPE_BINARY_DIV: *o = a / b; break;
        case LIBXSMM_MELTW_TYPE_BINARY_MULADD: prod = a * b; *o = *o + prod; break;
        case LIBXSMM_MELTW_TYPE_BINARY_MAX: *o = (a > b) ? a : b; break;
        case LIBXSMM_MELTW_TYPE_BINARY_MIN: *o = (a > b) ? b : a; break;
        default: ORACLE_DIE("unsupported f64 binary op");
      }
    } else {
      const float a = get_f32(p->in0.primary, elem_index(bc0, i, j, ldi), d->in0_type);
      const float b = get_f32(p->in1.primary, elem_index(bc1, i, j, ldi1), d->in1_type);
      if (is_cmp(d->type)) {
        bit_put((unsigned char*)p->out.primary, i, j, LIBXSMM_UPDIV(ldo, 16) * 16, binary_f32(d->type, a, b, 0.0f) > 0.1f);
      } else {
        const float prev = (d->type == LIBXSMM_MELTW_TYPE_BINARY_MULADD) ? get_f32(p->out.primary, i + j * ldo, d->out_type) : 0.0f;
        put_elem(p->out.primary, i + j * ldo, d->out_type, binary_f32(d->type, a, b, prev), (d->flags & LIBXSMM_MELTW_FLAG_BINARY_STOCHASTIC_ROUND) != 0, p->op.secondary, (unsigned int)(j * M + i));
      }
    }
  }
}

void oracle_meltw_ternary(const libxsmm_meltw_ternary_param* p, const oracle_meltw_desc* d) {
  const long long M = d->m, N = d->n, ldi = d->ldi, ldi1 = d->ldi2, ldi2 = d->ldi3, ldo = d->ldo;
  const int bc0 = bcast_kind(d, 0), bc1 = bcast_kind(d, 1), bc2 = bcast_kind(d, 2);
  long long i, j;
  for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) {
    if (d->type == LIBXSMM_MELTW_TYPE_TERNARY_SELECT) {                                /* [:2617-2640] */
      const int bit = bit_get((const unsigned char*)p->in2.primary, i, j, LIBXSMM_UPDIV(ldi2, 16) * 16);
      if (d->in0_type == LIBXSMM_DATATYPE_F64) {
        const double a = ((const double*)p->in0.primary)[elem_index(bc0, i, j, ldi)];
        const double b = ((const double*)p->in1.primary)[elem_index(bc1, i, j, ldi1)];
        ((double*)p->out.primary)[i + j * ldo] = bit ? b : a;
      } else {
        const float a = get_f32(p->in0.primary, elem_index(bc0, i, j, ldi), d->in0_type);
        const float b = get_f32(p->in1.primary, elem_index(bc1, i, j, ldi1), d->in1_type);
        put_elem(p->out.primary, i + j * ldo, d->out_type, bit ? b : a, (d->flags & LIBXSMM_MELTW_FLAG_TERNARY_STOCHASTIC_ROUND) != 0, p->op.secondary, (unsigned int)(j * M + i));
      }
    } else if (d->type == LIBXSMM_MELTW_TYPE_TERNARY_MULADD || d->type == LIBXSMM_MELTW_TYPE_TERNARY_NMULADD) {   /* [:2641-2655] */
      const float a = get_f32(p->in0.primary, elem_index(bc0, i, j, ldi), d->in0_type);
      const float b = get_f32(p->in1.primary, elem_index(bc1, i, j, ldi1), d->in1_type);
      const float c = get_f32(p->in2.primary, elem_index(bc2, i, j, ldi2), d->in2_type);
      float prod, r;
      if (d->type == LIBXSMM_MELTW_TYPE_TERNARY_MULADD) { prod = a * b; r = c + prod; }
      else { prod = a * c; r = b - prod; }
      put_elem(p->out.primary, i + j * ldo, d->out_type, r, (d->flags & LIBXSMM_MELTW_FLAG_TERNARY_STOCHASTIC_ROUND) != 0, p->op.secondary, (unsigned int)(j * M + i));
    } else {
      ORACLE_DIE("unsupported ternary op");
    }
  }
}
