// meltw_kernels.hip -- element-wise Tensor Processing Primitives (unary / binary / ternary) for gfx950.
//
// Semantics restated from the reference [ref: src/generator_mateltwise_reference_impl.c]:
//   out[i + j*ldo] = f(in0[idx0(i,j)], in1[idx1(i,j)], ...)  over an m x n column-major matrix, with
//   ROW / COL / SCALAR broadcast of any input (:241-272), fp32 compute for F32/BF16 data and one
//   RNE-with-DAZ rounding at a bf16 store (:299-324); pure data movement (copy, transpose, VNNI
//   re-layouts, gather/scatter, ZIP/UNZIP) is bit-exact on 1/2/4/8-byte payloads.
// All of these are HBM-bound streaming kernels: lanes run along i (the contiguous dimension), the
// batch axis of the batched launchers is the outermost grid dimension, and the contiguous f32/bf16
// case uses 16-byte (f32x4 / bf16x8) accesses.  The op is a run-time switch: the arithmetic is
// irrelevant next to the memory traffic.
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include "internal.hpp"
#include "lowp.hpp"

// a*b+c below means two roundings unless fma()/MFMA is spelled out: parity with the reference's C
// loops (built without FMA contraction) depends on it.
#pragma clang fp contract(off)

namespace xamd {

// kernel-argument-block pointers are generic to the compiler; everything dereferenced here is global memory
#define GM __attribute__((address_space(1)))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mw_bf2f(unsigned short x) { return __uint_as_float((unsigned int)x << 16); }
__device__ __forceinline__ unsigned short mw_f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7f800000u) == 0u) u &= 0x80000000u;
  if ((u & 0x7f800000u) == 0x7f800000u) { if (u & 0x007fffffu) u |= 0x00400000u; }
  else u += 0x00007fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
typedef GM const char* gcptr;
typedef GM char* gptr;
// element access of the general kernels: f32, bf16 and -- through lowp.hpp, bit-identical to the reference's conversions -- f16, bf8, hf8
__device__ __forceinline__ int mw_size(int type) {
  return type == LIBXSMM_DATATYPE_F32 ? 4 : (type == LIBXSMM_DATATYPE_BF8 || type == LIBXSMM_DATATYPE_HF8) ? 1 : 2;
}
__device__ __forceinline__ float mw_load(gcptr p, long long idx, int type) {
  switch (type) {
    case LIBXSMM_DATATYPE_F32: return ((GM const float*)p)[idx];
    case LIBXSMM_DATATYPE_F16: return lowp::f16_to_f32(((GM const unsigned short*)p)[idx]);
    case LIBXSMM_DATATYPE_BF8: return lowp::bf8_to_f32(((GM const unsigned char*)p)[idx]);
    case LIBXSMM_DATATYPE_HF8: return lowp::hf8_to_f32(((GM const unsigned char*)p)[idx]);
    default: return mw_bf2f(((GM const unsigned short*)p)[idx]);
  }
}
__device__ __forceinline__ void mw_store(gptr p, long long idx, int type, float v) {
  switch (type) {
    case LIBXSMM_DATATYPE_F32: ((GM float*)p)[idx] = v; break;
    case LIBXSMM_DATATYPE_F16: ((GM unsigned short*)p)[idx] = lowp::f32_to_f16(v); break;
    case LIBXSMM_DATATYPE_BF8: ((GM unsigned char*)p)[idx] = lowp::f16_to_bf8_rne(lowp::f32_to_f16(v)); break;
    case LIBXSMM_DATATYPE_HF8: ((GM unsigned char*)p)[idx] = lowp::f16_to_hf8_rne(lowp::f32_to_f16(v)); break;
    default: ((GM unsigned short*)p)[idx] = mw_f2bf(v);
  }
}

enum { BC_NONE = 0, BC_ROW = 1, BC_COL = 2, BC_SCALAR = 3 };
// broadcast kind of operand `op` [ref: :241-260]
__host__ __device__ inline int bcast_kind(int operation, int type, unsigned int f, int op) {
  if (operation == LIBXSMM_MELTW_OPERATION_UNARY) {
    if (op != 0) return BC_NONE;
    if (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_ROW) return BC_ROW;
    if ((f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_COL) || type == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR) return BC_COL;
    if (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_SCALAR) return BC_SCALAR;
  } else if (operation == LIBXSMM_MELTW_OPERATION_BINARY) {
    if (op > 1) return BC_NONE;
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_ROW_IN_0 << op)) return BC_ROW;
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0 << op)) return BC_COL;
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_SCALAR_IN_0 << op)) return BC_SCALAR;
  } else {
    if (op > 2) return BC_NONE;
    if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_ROW_IN_0 << op)) return BC_ROW;
    if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_COL_IN_0 << op)) return BC_COL;
    if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_SCALAR_IN_0 << op)) return BC_SCALAR;
  }
  return BC_NONE;
}
__device__ __forceinline__ long long bc_index(int kind, long long i, long long j, long long ld) {
  return kind == BC_ROW ? j * ld : kind == BC_COL ? i : kind == BC_SCALAR ? 0 : i + j * ld;
}

__device__ __forceinline__ float sigmoid_ref(float x) { return (tanhf(x * 0.5f) + 1.0f) * 0.5f; }
__device__ float unary_math(int type, float x, float alpha) {   // [ref: :83-127, :2148-2152]
  switch (type) {
    case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: case LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR: return x;
    case LIBXSMM_MELTW_TYPE_UNARY_XOR: return 0.0f;
    case LIBXSMM_MELTW_TYPE_UNARY_X2: return x * x;
    case LIBXSMM_MELTW_TYPE_UNARY_SQRT: return sqrtf(x);
    case LIBXSMM_MELTW_TYPE_UNARY_RELU: return (x <= 0.0f) ? 0.0f : x;
    case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU: return (x <= 0.0f) ? alpha * x : x;
    case LIBXSMM_MELTW_TYPE_UNARY_ELU: return (x <= 0.0f) ? alpha * (expf(x) - 1.0f) : x;
    case LIBXSMM_MELTW_TYPE_UNARY_TANH: return tanhf(x);
    case LIBXSMM_MELTW_TYPE_UNARY_TANH_INV: { const float t = tanhf(x); return 1.0f - t * t; }
    case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID: return sigmoid_ref(x);
    case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID_INV: { const float s = sigmoid_ref(x); return s * (1.0f - s); }
    case LIBXSMM_MELTW_TYPE_UNARY_GELU: return (erff(x / sqrtf(2.0f)) + 1.0f) * 0.5f * x;
    case LIBXSMM_MELTW_TYPE_UNARY_GELU_INV:
      return 0.5f + 0.5f * erff(x / sqrtf(2.0f)) + x / sqrtf(2.0f * 3.14159265358979323846f) * expf(-0.5f * x * x);
    case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: return -1.0f * x;
    case LIBXSMM_MELTW_TYPE_UNARY_INC: return x + 1.0f;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: return 1.0f / x;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: return 1.0f / sqrtf(x);
    case LIBXSMM_MELTW_TYPE_UNARY_EXP: return expf(x);
    default: return x;
  }
}
__device__ double unary_math_f64(int type, double x) {          // [ref: :129-152]
  switch (type) {
    case LIBXSMM_MELTW_TYPE_UNARY_XOR: return 0.0;
    case LIBXSMM_MELTW_TYPE_UNARY_X2: return x * x;
    case LIBXSMM_MELTW_TYPE_UNARY_SQRT: return sqrt(x);
    case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: return -1.0 * x;
    case LIBXSMM_MELTW_TYPE_UNARY_INC: return x + 1.0;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: return 1.0 / x;
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: return 1.0 / sqrt(x);
    default: return x;
  }
}
__device__ float binary_math(int type, float a, float b, float prev) {   // [ref: :181-214]
  switch (type) {
    case LIBXSMM_MELTW_TYPE_BINARY_ADD: return a + b;
    case LIBXSMM_MELTW_TYPE_BINARY_SUB: return a - b;
    case LIBXSMM_MELTW_TYPE_BINARY_MUL: return a * b;
    case LIBXSMM_MELTW_TYPE_BINARY_DIV: return a / b;
    case LIBXSMM_MELTW_TYPE_BINARY_MULADD: { const float prod = a * b; return prev + prod; }
    case LIBXSMM_MELTW_TYPE_BINARY_MAX: return (a > b) ? a : b;
    case LIBXSMM_MELTW_TYPE_BINARY_MIN: return (a > b) ? b : a;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GT: return (a > b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GE: return (a >= b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_LT: return (a < b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_LE: return (a <= b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_EQ: return (a == b) ? 1.0f : 0.0f;
    case LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_NE: return (a != b) ? 1.0f : 0.0f;
    default: return a;
  }
}

// A tile of 64 (i) x 4 (j): one wave = 64 consecutive rows of one column so bitmask bytes are ballots.
struct EwJob { unsigned int bidx; int i, j; bool valid; };
__device__ __forceinline__ EwJob ew_job(const MeltwArgs& p, int m, int n) {
  const int tiles_i = (m + 63) / 64, tiles_j = (n + 3) / 4;
  const long long per = (long long)tiles_i * tiles_j;
  const long long blk = blockIdx.x;
  EwJob e; e.bidx = (unsigned int)(blk / per);
  const int t = (int)(blk % per);
  e.i = (t % tiles_i) * 64 + threadIdx.x; e.j = (t / tiles_i) * 4 + threadIdx.y;
  e.valid = e.i < m && e.j < n;
  return e;
}
// write bit `on` for element (i,j) into a bit matrix with leading dimension ld_bits; ballot-combined
__device__ __forceinline__ void put_bits(GM unsigned char* bits, int i, int j, long long ld_bits, bool valid, bool on) {
  const int lane = threadIdx.x;
  const unsigned long long pos = __ballot(valid && on), val = __ballot(valid);
  if ((lane & 7) == 0 && valid) {
    GM unsigned char* byte = bits + i / 8 + (long long)j * (ld_bits / 8);
    const unsigned char vm = (unsigned char)((val >> lane) & 0xffu), nb = (unsigned char)((pos >> lane) & 0xffu);
    *byte = (unsigned char)((*byte & ~vm) | (nb & vm));
  }
}
__device__ __forceinline__ int get_bit(GM const unsigned char* bits, int i, int j, long long ld_bits) {
  return (bits[i / 8 + (long long)j * (ld_bits / 8)] >> (i % 8)) & 1;
}

__global__ __launch_bounds__(256) void meltw_unary_kernel(MeltwArgs p) {
  const int n_eff = (p.type == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR) ? (int)p.scalar_u64 : p.n;
  const EwJob e = ew_job(p, p.m, n_eff);
  gcptr in = (gcptr)p.in0 + (long long)e.bidx * p.bs_in0;
  gptr out = (gptr)p.out + (long long)e.bidx * p.bs_out;
  const int bc = bcast_kind(p.operation, p.type, p.flags, 0);
  const bool bitm = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) != 0;
  const int i = e.i, j = e.j;
  if (p.in0_type == LIBXSMM_DATATYPE_F64) {
    if (e.valid) ((GM double*)out)[i + (long long)j * p.ldo] = unary_math_f64(p.type, ((GM const double*)in)[bc_index(bc, i, j, p.ldi)]);
    return;
  }
  switch (p.type) {
    case LIBXSMM_MELTW_TYPE_UNARY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_ELU: {
      const float x = e.valid ? mw_load(in, bc_index(bc, i, j, p.ldi), p.in0_type) : 0.0f;
      if (e.valid) mw_store(out, i + (long long)j * p.ldo, p.out_type, unary_math(p.type, x, p.scalar_f32));
      if (bitm) put_bits((GM unsigned char*)p.aux_out + (long long)e.bidx * p.bs_aux, i, j, ((p.ldo + 15) / 16) * 16, e.valid, !(x <= 0.0f));
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_RELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_ELU_INV: {
      if (!e.valid) return;
      const float x = mw_load(in, bc_index(bc, i, j, p.ldi), p.in0_type);
      gcptr aux = (gcptr)p.aux_in + (long long)e.bidx * p.bs_aux;
      float y;
      if (p.type == LIBXSMM_MELTW_TYPE_UNARY_ELU_INV) {
        const float fwd = mw_load(aux, bc_index(bc, i, j, p.ldi), p.in0_type);
        y = (fwd > 0.0f) ? x : x * (fwd + p.scalar_f32);
      } else {
        const int bit = get_bit((GM const unsigned char*)aux, i, j, ((p.ldi + 15) / 16) * 16);
        y = (p.type == LIBXSMM_MELTW_TYPE_UNARY_RELU_INV) ? (bit ? x : 0.0f) : (bit ? x : p.scalar_f32 * x);
      }
      mw_store(out, i + (long long)j * p.ldo, p.out_type, y);
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_QUANT: {                      // f32 -> i8 / i16 / i32, round to nearest even [ref: :2195-2240]
      if (!e.valid) return;
      const bool sat = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_SIGN_SAT_QUANT) != 0;
      float t = rintf(((GM const float*)in)[bc_index(bc, i, j, p.ldi)] * p.scalar_f32);
      const long long o = i + (long long)j * p.ldo;
      if (p.out_type == LIBXSMM_DATATYPE_I8) {
        if (sat) { t = t < -128.0f ? -128.0f : t; t = t > 127.0f ? 127.0f : t; }
        ((GM signed char*)out)[o] = (signed char)(0xff & (int)t);
      } else if (p.out_type == LIBXSMM_DATATYPE_I16) {
        if (sat) { t = t < -32768.0f ? -32768.0f : t; t = t > 32767.0f ? 32767.0f : t; }
        ((GM short*)out)[o] = (short)(0xffff & (int)t);
      } else ((GM int*)out)[o] = (int)t;
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_DEQUANT: {                    // i8 / i16 / i32 -> f32 [ref: :2330-2360]
      if (!e.valid) return;
      const long long idx = bc_index(bc, i, j, p.ldi);
      const float v = p.in0_type == LIBXSMM_DATATYPE_I8 ? (float)((GM const signed char*)in)[idx] : p.in0_type == LIBXSMM_DATATYPE_I16 ? (float)((GM const short*)in)[idx] : (float)((GM const int*)in)[idx];
      ((GM float*)out)[i + (long long)j * p.ldo] = v * p.scalar_f32;
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_DUMP: {                       // identity that also lands in out.secondary, same ldo and type [ref: :2478-2494]
      if (!e.valid) return;
      const float x = mw_load(in, bc_index(bc, i, j, p.ldi), p.in0_type);
      mw_store(out, i + (long long)j * p.ldo, p.out_type, x);
      mw_store((gptr)p.aux_out + (long long)e.bidx * p.bs_aux, i + (long long)j * p.ldo, p.out_type, x);
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X2: case LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X3: {      // [ref: :2437-2470]
      // f32 -> a sum of two / three bf16: the leading piece(s) by TRUNCATION (upper 16 bits), the last by RNE of what is left (both subtractions are exact);
      // the pieces lie strides[0] / strides[1] BYTES behind the first (out.secondary, read on the host)
      if (!e.valid) return;
      const float x = ((GM const float*)in)[bc_index(bc, i, j, p.ldi)];
      const long long o = i + (long long)j * p.ldo;
      GM unsigned short* o16 = (GM unsigned short*)out;
      const unsigned int u1 = __float_as_uint(x) & 0xffff0000u;
      const float r1 = x - __uint_as_float(u1);
      o16[o] = (unsigned short)(u1 >> 16);
      if (p.type == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X3) {
        const unsigned int u2 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(u2);
        o16[o + (long long)(p.scalar_u64 / 2)] = (unsigned short)(u2 >> 16);
        o16[o + (long long)(p.scalar_u64b / 2)] = mw_f2bf(r2);
      } else o16[o + (long long)(p.scalar_u64 / 2)] = mw_f2bf(r1);
      return;
    }
    case LIBXSMM_MELTW_TYPE_UNARY_UNZIP: {                      // [ref: :2419-2432]
      if (!e.valid) return;
      const unsigned int u = ((GM const unsigned int*)in)[bc_index(bc, i, j, p.ldi)];
      ((GM unsigned short*)out)[i + (long long)j * p.ldo] = (unsigned short)(u & 0xffffu);
      ((GM unsigned short*)(out + p.scalar_u64))[i + (long long)j * p.ldo] = (unsigned short)(u >> 16);
      return;
    }
    default: break;
  }
  if (!e.valid) return;
  // pure copies / zero fill of same-width types stay bit-exact (no float round trip)
  if (p.in0_type == p.out_type && (p.type == LIBXSMM_MELTW_TYPE_UNARY_IDENTITY || p.type == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR)) {
    const int sz = mw_size(p.in0_type);
    if (sz == 4) ((GM unsigned int*)out)[i + (long long)j * p.ldo] = ((GM const unsigned int*)in)[bc_index(bc, i, j, p.ldi)];
    else if (sz == 2) ((GM unsigned short*)out)[i + (long long)j * p.ldo] = ((GM const unsigned short*)in)[bc_index(bc, i, j, p.ldi)];
    else ((GM unsigned char*)out)[i + (long long)j * p.ldo] = ((GM const unsigned char*)in)[bc_index(bc, i, j, p.ldi)];
    return;
  }
  const float x = mw_load(in, bc_index(bc, i, j, p.ldi), p.in0_type);
  mw_store(out, i + (long long)j * p.ldo, p.out_type, unary_math(p.type, x, p.scalar_f32));
}

// contiguous fast path (no broadcast, f32 or bf16 in == out type, m % 4 == 0, lds % 4 == 0, aligned):
// one thread = four consecutive rows.  Covers copy/zero/relu/... of the streaming TPPs.
template <typename VT, bool BF16>
__global__ __launch_bounds__(256) void meltw_unary_vec4_kernel(MeltwArgs p) {
  const int m4 = p.m / 4;
  const long long per = (long long)m4 * p.n;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= per * p.nbatch) return;
  const unsigned int bidx = (unsigned int)(gid / per);
  const long long t = gid % per;
  const int i = (int)(t % m4) * 4, j = (int)(t / m4);
  gcptr in = (gcptr)p.in0 + (long long)bidx * p.bs_in0;
  gptr out = (gptr)p.out + (long long)bidx * p.bs_out;
  float x[4];
  if (BF16) { const u16x4 v = *(GM const u16x4*)((GM const unsigned short*)in + i + (long long)j * p.ldi); for (int e = 0; e < 4; ++e) x[e] = mw_bf2f(v[e]); }
  else { const f32x4 v = *(GM const f32x4*)((GM const float*)in + i + (long long)j * p.ldi); for (int e = 0; e < 4; ++e) x[e] = v[e]; }
  if (BF16) { u16x4 o; for (int e = 0; e < 4; ++e) o[e] = mw_f2bf(unary_math(p.type, x[e], p.scalar_f32)); *(GM u16x4*)((GM unsigned short*)out + i + (long long)j * p.ldo) = o; }
  else { f32x4 o; for (int e = 0; e < 4; ++e) o[e] = unary_math(p.type, x[e], p.scalar_f32); *(GM f32x4*)((GM float*)out + i + (long long)j * p.ldo) = o; }
}

// ------------------------------------------------------------------------------------------------
// Streaming form of the element-wise TPPs: one thread = 8 consecutive rows of one column (16 bytes of bf16,
// 32 bytes of f32 per operand), f32/bf16 operands in any mix, every broadcast kind, unary / binary / ternary
// arithmetic (no bit-matrix in or out).  The per-element arithmetic is the same unary_math / binary_math as the
// general kernels, so results are bit-identical to them; the f32 -> bf16 store uses v_cvt_pk_bf16_f32, which
// equals the reference's rounding for every input except f32 denormals (probed over all 2^32 patterns,
// tools/cvt_probe.hip) -- those are flushed first (DAZ) [ref: src/libxsmm_math.c:684-704].
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
typedef __bf16 mwbf16x2 __attribute__((ext_vector_type(2)));
typedef float mwf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float mw_daz(float x) { return ((__float_as_uint(x) & 0x7f800000u) == 0u) ? __uint_as_float(__float_as_uint(x) & 0x80000000u) : x; }
__device__ __forceinline__ unsigned int mw_f2bf_pk(float lo, float hi) {
  const mwf32x2 v = {mw_daz(lo), mw_daz(hi)};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, mwbf16x2));
}
// NT (round 6): non-temporal policy for launches whose operands cannot be cache resident (MeltwArgs::nt) -- a bare copy of this footprint gains 7-10 % from it
// (tools/copy_floor.hip: 0.72 -> 0.80 of 8 TB/s)
template <bool NT, typename V> __device__ __forceinline__ V ld_pol(GM const V* p) { if (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT, typename V> __device__ __forceinline__ void st_pol(GM V* p, V v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT = false, int E = 8>      // E = 4: all operands f32, ONE 16-byte access per lane (a wave covers 1 KiB without gaps)
__device__ __forceinline__ void ew8_load(float (&x)[E], gcptr base, int type, int kind, long long i, long long j, long long ld) {
  if (kind == BC_ROW || kind == BC_SCALAR) {
    const float v = mw_load(base, kind == BC_ROW ? j * ld : 0, type);
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = v;
    return;
  }
  const long long idx = (kind == BC_COL) ? i : i + j * ld;
  if constexpr (E == 4) {
    // (a cacheable load for a broadcast column next to the non-temporal one was tried and taken back: the compiler folded the two loads of this function into ONE
    //  without the non-temporal bit -- the f32 copy lost its 5 %, found in the round's last closing run)
    const f32x4 a = ld_pol<NT>((GM const f32x4*)((GM const float*)base + idx));
    x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3];
  } else
  if (type == LIBXSMM_DATATYPE_F32) {
    const f32x4 a = ld_pol<NT>((GM const f32x4*)((GM const float*)base + idx)), b = ld_pol<NT>((GM const f32x4*)((GM const float*)base + idx + 4));
    x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
  } else {
    const u32x4e v = ld_pol<NT>((GM const u32x4e*)((GM const unsigned short*)base + idx));
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(v[e] << 16); x[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
  }
}
template <bool NT = false, int E = 8>
__device__ __forceinline__ void ew8_store(gptr base, int type, long long idx, const float (&y)[E]) {
  if constexpr (E == 4) {
    f32x4 a; a[0] = y[0]; a[1] = y[1]; a[2] = y[2]; a[3] = y[3];
    st_pol<NT>((GM f32x4*)((GM float*)base + idx), a);
  } else
  if (type == LIBXSMM_DATATYPE_F32) {
    f32x4 a, b; a[0] = y[0]; a[1] = y[1]; a[2] = y[2]; a[3] = y[3]; b[0] = y[4]; b[1] = y[5]; b[2] = y[6]; b[3] = y[7];
    st_pol<NT>((GM f32x4*)((GM float*)base + idx), a); st_pol<NT>((GM f32x4*)((GM float*)base + idx + 4), b);
  } else {
    u32x4e v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mw_f2bf_pk(y[2 * e], y[2 * e + 1]);
    st_pol<NT>((GM u32x4e*)((GM unsigned short*)base + idx), v);
  }
}
// Round 4, measured and NOT adopted (4096 x 8192 f32 copy 0.66, bf16 ReLU tiles 0.77 with this kernel): (a) four chunks per thread a grid apart with all loads
// issued first, 16 bytes per lane for f32-only TPPs (whole-line accesses instead of two half-line ones): 0.60 / 0.70 -- fewer, longer-lived waves lose, as in the
// ragged GEMM kernels (DESIGN decision 11); (b) the f32 transpose as 4 x 4 register blocks with 16-byte LDS traffic only (8 LDS instructions per thread instead
// of 32): 0.595 against 0.604 -- LDS is not what holds the transposes at 0.9 of the copy's rate.
template <int NIN, bool NT = false, int E = 8>
__global__ __launch_bounds__(256) void meltw_ew8_kernel(MeltwArgs p, unsigned int m8, unsigned int total) {
  const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= total) return;
  const unsigned int i8 = gid % m8, t = gid / m8, j = t % (unsigned int)p.n, bidx = t / (unsigned int)p.n;
  const long long i = (long long)E * i8;
  gptr out = (gptr)p.out + (long long)bidx * p.bs_out;
  const long long oidx = i + (long long)j * p.ldo;
  float x0[E], x1[E], x2[E], y[E];
  ew8_load<NT, E>(x0, (gcptr)p.in0 + (long long)bidx * p.bs_in0, p.in0_type, bcast_kind(p.operation, p.type, p.flags, 0), i, j, p.ldi);
  if (NIN >= 2) ew8_load<NT, E>(x1, (gcptr)p.in1 + (long long)bidx * p.bs_in1, p.in1_type, bcast_kind(p.operation, p.type, p.flags, 1), i, j, p.ldi1);
  if (NIN >= 3) ew8_load<NT, E>(x2, (gcptr)p.in2 + (long long)bidx * p.bs_in2, p.in2_type, bcast_kind(p.operation, p.type, p.flags, 2), i, j, p.ldi2);
  if (NIN == 1) {
    if (p.type == LIBXSMM_MELTW_TYPE_UNARY_IDENTITY) {
#pragma unroll
      for (int e = 0; e < E; ++e) y[e] = x0[e];
    } else if (p.type == LIBXSMM_MELTW_TYPE_UNARY_RELU) {
#pragma unroll
      for (int e = 0; e < E; ++e) y[e] = (x0[e] <= 0.0f) ? 0.0f : x0[e];
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) y[e] = unary_math(p.type, x0[e], p.scalar_f32);
    }
  } else if (NIN == 2) {
    if (p.type == LIBXSMM_MELTW_TYPE_BINARY_ADD) {
#pragma unroll
      for (int e = 0; e < E; ++e) y[e] = x0[e] + x1[e];
    } else if (p.type == LIBXSMM_MELTW_TYPE_BINARY_MUL) {
#pragma unroll
      for (int e = 0; e < E; ++e) y[e] = x0[e] * x1[e];
    } else {
      float prev[E];
      if (p.type == LIBXSMM_MELTW_TYPE_BINARY_MULADD) ew8_load<false, E>(prev, (gcptr)out, p.out_type, BC_NONE, i, j, p.ldo);
#pragma unroll
      for (int e = 0; e < E; ++e) y[e] = binary_math(p.type, x0[e], x1[e], p.type == LIBXSMM_MELTW_TYPE_BINARY_MULADD ? prev[e] : 0.0f);
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float prod = (p.type == LIBXSMM_MELTW_TYPE_TERNARY_MULADD) ? x0[e] * x1[e] : x0[e] * x2[e];
      y[e] = (p.type == LIBXSMM_MELTW_TYPE_TERNARY_MULADD) ? x2[e] + prod : x1[e] - prod;
    }
  }
  ew8_store<NT, E>(out, p.out_type, oidx, y);
}

static bool is_float_type(int t) { return t == LIBXSMM_DATATYPE_F32 || t == LIBXSMM_DATATYPE_BF16; }      // the types the vector kernels know
// ... and the ones the general kernels convert element by element [ref: mateltwise ref :262-324: F16, BF8, HF8 in and out]
static bool is_tpp_float(int t) { return is_float_type(t) || t == LIBXSMM_DATATYPE_F16 || t == LIBXSMM_DATATYPE_BF8 || t == LIBXSMM_DATATYPE_HF8; }
// elements per thread of meltw_ew8_kernel: eight (16 bytes of bf16); FOUR when every operand and the result are f32 -- eight f32 are two 16-byte accesses 32 bytes apart
// per lane, each instruction then touches every other 16 bytes of a 2 KiB span (round 6: f32 copy 0.665 -> see DESIGN)
static int ew_elems(const MeltwArgs& a) {
  const int nin = a.operation == LIBXSMM_MELTW_OPERATION_UNARY ? 1 : a.operation == LIBXSMM_MELTW_OPERATION_BINARY ? 2 : 3;
  const int types[3] = {a.in0_type, a.in1_type, a.in2_type};
  bool f32 = a.out_type == LIBXSMM_DATATYPE_F32;
  for (int o = 0; o < nin; ++o) f32 = f32 && types[o] == LIBXSMM_DATATYPE_F32;
  return f32 ? 4 : 8;
}
// is this TPP eligible for meltw_ew8_kernel?
static bool ew8_ok(const MeltwArgs& a) {
  const int nin = a.operation == LIBXSMM_MELTW_OPERATION_UNARY ? 1 : a.operation == LIBXSMM_MELTW_OPERATION_BINARY ? 2 : 3;
  const int g = ew_elems(a);                 // elements per thread: 4 when every operand is f32, else 8
  if (a.m % g != 0 || a.ldo % g != 0) return false;
  if (!is_float_type(a.out_type) || ((size_t)a.out % 16) || ((size_t)a.bs_out % 16)) return false;
  const void* ptrs[3] = {a.in0, a.in1, a.in2}; const long long lds[3] = {a.ldi, a.ldi1, a.ldi2}; const long long bss[3] = {a.bs_in0, a.bs_in1, a.bs_in2};
  const int types[3] = {a.in0_type, a.in1_type, a.in2_type};
  for (int o = 0; o < nin; ++o) {
    if (!is_float_type(types[o])) return false;
    const int k = bcast_kind(a.operation, a.type, a.flags, o);
    if (k == BC_NONE && lds[o] % g != 0) return false;
    if ((k == BC_NONE || k == BC_COL) && (((size_t)ptrs[o] % 16) || ((size_t)bss[o] % 16))) return false;
  }
  if ((long long)(a.m / g) * a.n * (long long)a.nbatch >= (1ll << 32) - 256) return false;
  if (nin == 1) {
    switch (a.type) {   // arithmetic TPPs without side channels
      case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: case LIBXSMM_MELTW_TYPE_UNARY_XOR: case LIBXSMM_MELTW_TYPE_UNARY_X2: case LIBXSMM_MELTW_TYPE_UNARY_SQRT:
      case LIBXSMM_MELTW_TYPE_UNARY_TANH: case LIBXSMM_MELTW_TYPE_UNARY_TANH_INV: case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID: case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID_INV:
      case LIBXSMM_MELTW_TYPE_UNARY_GELU: case LIBXSMM_MELTW_TYPE_UNARY_GELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: case LIBXSMM_MELTW_TYPE_UNARY_INC:
      case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: case LIBXSMM_MELTW_TYPE_UNARY_EXP:
        return true;
      case LIBXSMM_MELTW_TYPE_UNARY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_ELU:
        return !(a.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT);
      default: return false;
    }
  }
  if (nin == 2) return a.type == LIBXSMM_MELTW_TYPE_BINARY_ADD || a.type == LIBXSMM_MELTW_TYPE_BINARY_SUB || a.type == LIBXSMM_MELTW_TYPE_BINARY_MUL ||
                       a.type == LIBXSMM_MELTW_TYPE_BINARY_DIV || a.type == LIBXSMM_MELTW_TYPE_BINARY_MULADD || a.type == LIBXSMM_MELTW_TYPE_BINARY_MAX || a.type == LIBXSMM_MELTW_TYPE_BINARY_MIN;
  return a.type == LIBXSMM_MELTW_TYPE_TERNARY_MULADD || a.type == LIBXSMM_MELTW_TYPE_TERNARY_NMULADD;
}

__global__ __launch_bounds__(256) void meltw_binary_kernel(MeltwArgs p) {
  const EwJob e = ew_job(p, p.m, p.n);
  gcptr in0 = (gcptr)p.in0 + (long long)e.bidx * p.bs_in0;
  gcptr in1 = (gcptr)p.in1 + (long long)e.bidx * p.bs_in1;
  gptr out = (gptr)p.out + (long long)e.bidx * p.bs_out;
  const int bc0 = bcast_kind(p.operation, p.type, p.flags, 0), bc1 = bcast_kind(p.operation, p.type, p.flags, 1);
  const int i = e.i, j = e.j;
  const bool cmp = p.type >= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GT && p.type <= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_NE;
  if (p.type == LIBXSMM_MELTW_TYPE_BINARY_ZIP) {                // [ref: :2543-2556]
    if (!e.valid) return;
    const unsigned int lo = ((GM const unsigned short*)in0)[bc_index(bc0, i, j, p.ldi)];
    const unsigned int hi = ((GM const unsigned short*)in1)[bc_index(bc1, i, j, p.ldi1)];
    ((GM unsigned int*)out)[i + (long long)j * p.ldo] = lo | (hi << 16);
    return;
  }
  if (p.in0_type == LIBXSMM_DATATYPE_F64) {
    if (!e.valid) return;
    const double a = ((GM const double*)in0)[bc_index(bc0, i, j, p.ldi)], b = ((GM const double*)in1)[bc_index(bc1, i, j, p.ldi1)];
    GM double* o = (GM double*)out + i + (long long)j * p.ldo;
    switch (p.type) {
      case LIBXSMM_MELTW_TYPE_BINARY_ADD: *o = a + b; break;
      case LIBXSMM_MELTW_TYPE_BINARY_SUB: *o = a - b; break;
      case LIBXSMM_MELTW_TYPE_BINARY_MUL: *o = a * b; break;
      case LIBXSMM_MELTW_TYPE_BINARY_DIV: *o = a / b; break;
      case LIBXSMM_MELTW_TYPE_BINARY_MULADD: { const double prod = a * b; *o = *o + prod; } break;
      case LIBXSMM_MELTW_TYPE_BINARY_MAX: *o = (a > b) ? a : b; break;
      default: *o = (a > b) ? b : a; break;
    }
    return;
  }
  float a = 0.0f, b = 0.0f;
  if (e.valid) { a = mw_load(in0, bc_index(bc0, i, j, p.ldi), p.in0_type); b = mw_load(in1, bc_index(bc1, i, j, p.ldi1), p.in1_type); }
  if (cmp) {   // result is a bit matrix, ld rounded up to 16 [ref: :2575-2584]
    put_bits((GM unsigned char*)out, i, j, ((p.ldo + 15) / 16) * 16, e.valid, binary_math(p.type, a, b, 0.0f) > 0.1f);
    return;
  }
  if (!e.valid) return;
  const float prev = (p.type == LIBXSMM_MELTW_TYPE_BINARY_MULADD) ? mw_load(out, i + (long long)j * p.ldo, p.out_type) : 0.0f;
  mw_store(out, i + (long long)j * p.ldo, p.out_type, binary_math(p.type, a, b, prev));
}

__global__ __launch_bounds__(256) void meltw_ternary_kernel(MeltwArgs p) {
  const EwJob e = ew_job(p, p.m, p.n);
  if (!e.valid) return;
  gcptr in0 = (gcptr)p.in0 + (long long)e.bidx * p.bs_in0;
  gcptr in1 = (gcptr)p.in1 + (long long)e.bidx * p.bs_in1;
  gcptr in2 = (gcptr)p.in2 + (long long)e.bidx * p.bs_in2;
  gptr out = (gptr)p.out + (long long)e.bidx * p.bs_out;
  const int bc0 = bcast_kind(p.operation, p.type, p.flags, 0), bc1 = bcast_kind(p.operation, p.type, p.flags, 1), bc2 = bcast_kind(p.operation, p.type, p.flags, 2);
  const int i = e.i, j = e.j;
  if (p.type == LIBXSMM_MELTW_TYPE_TERNARY_SELECT) {            // [ref: :2617-2640]
    const int bit = get_bit((GM const unsigned char*)in2, i, j, ((p.ldi2 + 15) / 16) * 16);
    if (p.in0_type == LIBXSMM_DATATYPE_F64) {
      const double a = ((GM const double*)in0)[bc_index(bc0, i, j, p.ldi)], b = ((GM const double*)in1)[bc_index(bc1, i, j, p.ldi1)];
      ((GM double*)out)[i + (long long)j * p.ldo] = bit ? b : a;
    } else {
      const float a = mw_load(in0, bc_index(bc0, i, j, p.ldi), p.in0_type), b = mw_load(in1, bc_index(bc1, i, j, p.ldi1), p.in1_type);
      mw_store(out, i + (long long)j * p.ldo, p.out_type, bit ? b : a);
    }
    return;
  }
  const float a = mw_load(in0, bc_index(bc0, i, j, p.ldi), p.in0_type);
  const float b = mw_load(in1, bc_index(bc1, i, j, p.ldi1), p.in1_type);
  const float c = mw_load(in2, bc_index(bc2, i, j, p.ldi2), p.in2_type);
  const float prod = (p.type == LIBXSMM_MELTW_TYPE_TERNARY_MULADD) ? a * b : a * c;
  const float r = (p.type == LIBXSMM_MELTW_TYPE_TERNARY_MULADD) ? c + prod : b - prod;   // [ref: :2641-2655]
  mw_store(out, i + (long long)j * p.ldo, p.out_type, r);
}

// ------------------------------------------------------------------------------------------------
// data-movement kernels on S-byte payloads (bit-exact)
// ------------------------------------------------------------------------------------------------
template <int S> struct Payload;
template <> struct Payload<1> { typedef unsigned char type; };
template <> struct Payload<2> { typedef unsigned short type; };
template <> struct Payload<4> { typedef unsigned int type; };
template <> struct Payload<8> { typedef unsigned long long type; };

// out[j*ldo + i] = in[i*ldi + j] for i < n, j < m (in is m x n, out is n x m) [ref: :376-424].
// 32x32 tile through LDS so that both the read and the write are coalesced.
template <int S>
__global__ __launch_bounds__(256) void transpose_kernel(MeltwArgs p) {
  typedef typename Payload<S>::type T;
  __shared__ T tile[32][33];
  const int tm = (p.m + 31) / 32, tn = (p.n + 31) / 32;
  const long long per = (long long)tm * tn;
  const unsigned int bidx = (unsigned int)(blockIdx.x / per);
  const int t = (int)(blockIdx.x % per);
  const int r0 = (t % tm) * 32, c0 = (t / tm) * 32;       // r: index along m (contiguous in `in`), c: along n
  GM const T* in = (GM const T*)((gcptr)p.in0 + (long long)bidx * p.bs_in0);
  GM T* out = (GM T*)((gptr)p.out + (long long)bidx * p.bs_out);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int cc = ty; cc < 32; cc += 8) {
    const int r = r0 + tx, c = c0 + cc;
    if (r < p.m && c < p.n) tile[cc][tx] = in[(long long)c * p.ldi + r];
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int c = c0 + tx, r = r0 + rr;
    if (r < p.m && c < p.n) out[(long long)r * p.ldo + c] = tile[tx][rr];
  }
}

// (Round 6, measured and not adopted: 128 x 128 tiles -- 512-byte instead of 256-byte pieces per request, 65 KiB of LDS, two workgroups per CU: f32 4096 x 8192 0.615 -> 0.537;
//  padded leading dimensions change nothing either: the transposes are not held back by channel aliasing, profiles/r06_transpose_pitch.jsonl)
// Vector form of the transpose: 64x64 tiles, 16-byte global accesses on both sides (VEC = 16/S elements per thread
// access), the element shuffle happens in LDS (row pitch padded by one dword).  Needs m, n, ldi, ldo multiples of VEC
// and 16-byte aligned bases; tiles at the matrix edge are guarded per vector.
template <int S>
__global__ __launch_bounds__(256) void transpose_vec_kernel(MeltwArgs p, unsigned int tm, unsigned int tn) {
  typedef typename Payload<S>::type T;
  constexpr int VEC = 16 / S, PITCH = 64 + 4 / S, VPR = 64 / VEC;      // vectors per tile row
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ T tile[64 * PITCH];
  const unsigned int per = tm * tn, bidx = blockIdx.x / per, t = blockIdx.x % per;
  const int r0 = (int)(t % tm) * 64, c0 = (int)(t / tm) * 64;           // r: along m (contiguous in `in`), c: along n
  GM const T* in = (GM const T*)((gcptr)p.in0 + (long long)bidx * p.bs_in0);
  GM T* out = (GM T*)((gptr)p.out + (long long)bidx * p.bs_out);
#pragma unroll
  for (int q = 0; q < 64 * VPR / 256; ++q) {
    const int v = threadIdx.x + 256 * q, cc = v / VPR, rr = (v % VPR) * VEC;
    if (r0 + rr < p.m && c0 + cc < p.n) {
      GM const vec_t* src = (GM const vec_t*)(in + (long long)(c0 + cc) * p.ldi + r0 + rr);
      const vec_t x = p.nt ? __builtin_nontemporal_load(src) : *src;
#pragma unroll
      for (int e = 0; e < VEC; ++e) tile[cc * PITCH + rr + e] = x[e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 64 * VPR / 256; ++q) {
    const int v = threadIdx.x + 256 * q, rr = v / VPR, cc = (v % VPR) * VEC;
    if (r0 + rr < p.m && c0 + cc < p.n) {
      vec_t y;
#pragma unroll
      for (int e = 0; e < VEC; ++e) y[e] = tile[(cc + e) * PITCH + rr];
      GM vec_t* dst = (GM vec_t*)(out + (long long)(r0 + rr) * p.ldo + c0 + cc);
      if (p.nt) __builtin_nontemporal_store(y, dst); else *dst = y;
    }
  }
}

// NORM -> VNNI2 of 16-bit payloads without LDS: a thread reads 8 consecutive i of rows 2jp and 2jp+1 (16 bytes each)
// and writes the 8 interleaved pairs (32 contiguous bytes); i in [m, ldo) and the odd-n pad row are zero filled as the
// reference does [ref: mateltwise ref :532-557].  Needs m, ldi, ldo multiples of 8 and 16-byte aligned bases.
template <int E>      // E positions i per thread: 8 (two 16-byte loads, 32 bytes out) or 4 (round 6: two 8-byte loads, ONE 16-byte store -- a wave's stores are 1 KiB without gaps)
__global__ __launch_bounds__(256) void vnni2_vec_kernel(MeltwArgs p, unsigned int o8, unsigned int total) {
  typedef unsigned short u16xE __attribute__((ext_vector_type(E)));
  typedef unsigned int u32xH __attribute__((ext_vector_type(E / 2)));
  const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= total) return;
  const unsigned int np = (unsigned int)(p.n + 1) / 2u;
  const unsigned int i8 = gid % o8, t = gid / o8, jp = t % np, bidx = t / np;
  GM const unsigned short* in = (GM const unsigned short*)((gcptr)p.in0 + (long long)bidx * p.bs_in0);
  GM unsigned int* out = (GM unsigned int*)((gptr)p.out + (long long)bidx * p.bs_out);
  const long long i = (long long)E * i8;
  u16xE a = (u16xE)(unsigned short)0, b = (u16xE)(unsigned short)0;
  if (i < p.m) {
    GM const u16xE* pa = (GM const u16xE*)(in + (long long)(2 * jp) * p.ldi + i);
    a = p.nt ? __builtin_nontemporal_load(pa) : *pa;
    if ((int)(2 * jp + 1) < p.n) { GM const u16xE* pb = (GM const u16xE*)(in + (long long)(2 * jp + 1) * p.ldi + i); b = p.nt ? __builtin_nontemporal_load(pb) : *pb; }
  }
  u32xH lo, hi;
#pragma unroll
  for (int e = 0; e < E / 2; ++e) { lo[e] = (unsigned int)a[e] | ((unsigned int)b[e] << 16); hi[e] = (unsigned int)a[E / 2 + e] | ((unsigned int)b[E / 2 + e] << 16); }
  GM unsigned int* dst = out + (long long)jp * p.ldo + i;
  if constexpr (E == 4) {
    u32x4e v; v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    if (p.nt) __builtin_nontemporal_store(v, (GM u32x4e*)dst); else *(GM u32x4e*)dst = v;
  } else {
    if (p.nt) { __builtin_nontemporal_store(lo, (GM u32xH*)dst); __builtin_nontemporal_store(hi, (GM u32xH*)(dst + E / 2)); }
    else { *(GM u32xH*)dst = lo; *(GM u32xH*)(dst + E / 2) = hi; }
  }
}

// NORM -> VNNI4 of 8-bit payloads without LDS (the producer side of the 8-bit GEMMs / BCSC): a thread reads 4 consecutive i of the four
// rows 4jq .. 4jq+3 (one dword each: a wave reads whole 256-byte row segments) and writes the 4 x 4 byte transpose as 16 contiguous
// bytes (out[(jq*ldo + i)*4 + j2] = in[(4jq + j2)*ldi + i]); i in [m, ldo) and the rows past n are zero filled as the reference does
// [ref: mateltwise ref :532-557 with v = 4].  Needs m, ldi, ldo multiples of 4 and 16-byte aligned bases.
__global__ __launch_bounds__(256) void vnni4_vec_kernel(MeltwArgs p, unsigned int o4, unsigned int total) {
  const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= total) return;
  const unsigned int nq = (unsigned int)(p.n + 3) / 4u;
  const unsigned int i4 = gid % o4, t = gid / o4, jq = t % nq, bidx = t / nq;
  GM const unsigned char* in = (GM const unsigned char*)((gcptr)p.in0 + (long long)bidx * p.bs_in0);
  GM unsigned int* out = (GM unsigned int*)((gptr)p.out + (long long)bidx * p.bs_out);
  const long long i = 4ll * i4;
  unsigned int r[4] = {0u, 0u, 0u, 0u};
  if (i < p.m) {
#pragma unroll
    for (int c = 0; c < 4; ++c) if ((int)(4 * jq + c) < p.n) r[c] = *(GM const unsigned int*)(in + (long long)(4 * jq + c) * p.ldi + i);
  }
  // byte e of r[c] = element (row 4jq + c, i + e)  ->  output dword e = bytes (c = 0..3) of column i + e
  const unsigned int t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);   // (r0.b0 r1.b0 r0.b1 r1.b1), (r0.b2 r1.b2 r0.b3 r1.b3)
  const unsigned int t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
  u32x4e o;
  o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
  *(GM u32x4e*)(out + (long long)jq * p.ldo + i) = o;
}

// generic index-remapping transforms: one thread per OUTPUT element of the padded extent;
// `mode` encodes the reference loop nest being restated.
enum XformMode { XF_NORM_TO_VNNI = 1, XF_VNNI_TO_VNNIT, XF_NORM_TO_VNNIT, XF_VNNIT_TO_NORM, XF_VNNI4_TO_NORM, XF_VNNI4_TO_VNNI2, XF_PAD };
template <int S>
__global__ __launch_bounds__(256) void xform_kernel(MeltwArgs p, int mode, int v, int pad_m, int pad_n) {
  typedef typename Payload<S>::type T;
  GM const T* in = (GM const T*)((gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0);
  GM T* out = (GM T*)((gptr)p.out + (long long)blockIdx.y * p.bs_out);
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long M = p.m, N = p.n, ldi = p.ldi, ldo = p.ldo;
  switch (mode) {
    case XF_NORM_TO_VNNI: {   // out[(j*ldo*v) + i*v + j2] = in[(j*v+j2)*ldi + i]; zero fill of ldo*Nn first [ref: :532-557]
      const long long Nn = ((N + v - 1) / v) * v, total = ldo * Nn;
      if (gid >= total) return;
      const long long jb = gid / (ldo * v), rem = gid % (ldo * v), i = rem / v, j2 = rem % v, jsrc = jb * v + j2;
      out[gid] = (i < M && jsrc < N) ? in[jsrc * ldi + i] : (T)0;
      return;
    }
    case XF_VNNI_TO_VNNIT: {  // [ref: :427-446] out[j*ldo*v + j2 + (i*v+i2)*v] = in[i*ldi*v + i2 + (j*v+j2)*v]
      const long long total = (M / v) * (N / v) * v * v;
      if (gid >= total) return;
      const long long i2 = gid % v, j2 = (gid / v) % v, i = (gid / (v * v)) % (N / v), j = gid / (v * v * (N / v));
      out[j * ldo * v + j2 + (i * v + i2) * v] = in[i * ldi * v + i2 + (j * v + j2) * v];
      return;
    }
    case XF_NORM_TO_VNNIT: {  // [ref: :560-578] out[i*ldo*v + j*v + i2] = in[j*ldi + i*v + i2]
      const long long total = (M / v) * N * v;
      if (gid >= total) return;
      const long long i2 = gid % v, j = (gid / v) % N, i = gid / (v * N);
      out[i * ldo * v + j * v + i2] = in[j * ldi + i * v + i2];
      return;
    }
    case XF_VNNIT_TO_NORM: {  // [ref: :581-640] (m and n swap roles) out[j*ldo + i*v + i2] = in[i*ldi*v + j*v + i2]
      const long long Mm = N, Nn = M, total = (Mm / v) * Nn * v;
      if (gid >= total) return;
      const long long i2 = gid % v, j = (gid / v) % Nn, i = gid / (v * Nn);
      out[j * ldo + i * v + i2] = in[i * ldi * v + j * v + i2];
      return;
    }
    case XF_VNNI4_TO_NORM: {  // [ref: :788-804] out[i*ldo + j] = in[(i/4)*ldi*4 + j*4 + i%4]
      if (gid >= M * N) return;
      const long long j = gid % M, i = gid / M;
      out[i * ldo + j] = in[(i / 4) * ldi * 4 + j * 4 + (i % 4)];
      return;
    }
    case XF_VNNI4_TO_VNNI2: { // [ref: :807-823]
      if (gid >= M * N) return;
      const long long j = gid % M, i = gid / M;
      out[(i / 2) * ldo * 2 + j * 2 + (i % 2)] = in[(i / 4) * ldi * 4 + j * 4 + (i % 4)];
      return;
    }
    default: {                // XF_PAD [ref: :826-964]: zero ldo x pad_n, copy m x n
      const long long total = ldo * (long long)pad_n;
      if (gid >= total) return;
      const long long i = gid % ldo, j = gid / ldo;
      out[gid] = (i < M && j < N) ? in[j * ldi + i] : (T)0;
      (void)pad_m;
      return;
    }
  }
}

// NORM -> VNNI2 of 16-bit elements with any leading dimensions (round 3; the 16-byte kernel above needs multiples of 8): a thread owns ONE output dword =
// rows (2 jb, 2 jb + 1) of column position i, two 2-byte loads (coalesced along i) and one dword store; positions beyond m and a missing odd row are zero
// like the reference's fill [ref: :532-557].  32-bit index arithmetic, no division per element beyond one per thread.
__global__ __launch_bounds__(256) void vnni2_pair_kernel(MeltwArgs p, unsigned int per_batch) {
  const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= per_batch) return;
  const unsigned int ldo = (unsigned int)p.ldo, jb = gid / ldo, i = gid - jb * ldo;
  GM const unsigned short* in = (GM const unsigned short*)((gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0);
  GM unsigned int* out = (GM unsigned int*)((gptr)p.out + (long long)blockIdx.y * p.bs_out);
  unsigned int lo = 0, hi = 0;
  if (i < (unsigned int)p.m) {
    lo = in[(long long)(2u * jb) * p.ldi + i];
    if (2u * jb + 1u < (unsigned int)p.n) hi = in[(long long)(2u * jb + 1u) * p.ldi + i];
  }
  out[gid] = lo | (hi << 16);
}

// The same transform, FOUR output dwords per thread (round 4): the pair kernel moves 4 bytes per memory instruction (0.37 of the HBM roofline on 4090 x 8192);
// here a thread owns positions i0 .. i0 + 3 of one row pair and uses the widest access the addresses allow -- the even row of a pair starts 4-byte aligned
// whatever ldi is (2 * 2 jb * ldi bytes), the odd row when ldi is even; 8-byte loads when the row offset allows; the four output dwords leave as one 16-byte,
// two 8-byte or four 4-byte stores.  The alignment cases are the same for all lanes of a row, so the branches are (nearly) wave-uniform.
__device__ __forceinline__ void load4_u16(GM const unsigned short* src, unsigned int valid, unsigned int (&v)[4]) {
  v[0] = v[1] = v[2] = v[3] = 0u;
  if (valid == 0u) return;
  const unsigned long long a = (unsigned long long)(size_t)src;
  if (valid >= 4u && (a & 7ull) == 0ull) {
    const unsigned long long w = *(GM const unsigned long long*)src;
    v[0] = (unsigned int)(w & 0xffffu); v[1] = (unsigned int)((w >> 16) & 0xffffu); v[2] = (unsigned int)((w >> 32) & 0xffffu); v[3] = (unsigned int)(w >> 48);
  } else if (valid >= 4u && (a & 3ull) == 0ull) {
    const unsigned int w0 = ((GM const unsigned int*)src)[0], w1 = ((GM const unsigned int*)src)[1];
    v[0] = w0 & 0xffffu; v[1] = w0 >> 16; v[2] = w1 & 0xffffu; v[3] = w1 >> 16;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if ((unsigned int)e < valid) v[e] = src[e];
  }
}
__global__ __launch_bounds__(256) void vnni2_quad_kernel(MeltwArgs p, unsigned int q4, unsigned int per_batch) {
  typedef unsigned int u32x2q __attribute__((ext_vector_type(2)));
  const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= per_batch) return;
  const unsigned int jb = gid / q4, i0 = (gid - jb * q4) * 4u, ldo = (unsigned int)p.ldo, m = (unsigned int)p.m;
  GM const unsigned short* in = (GM const unsigned short*)((gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0);
  GM unsigned int* out = (GM unsigned int*)((gptr)p.out + (long long)blockIdx.y * p.bs_out) + (long long)jb * ldo + i0;
  const unsigned int valid = i0 < m ? (m - i0 < 4u ? m - i0 : 4u) : 0u;
  unsigned int lo[4], hi[4];
  load4_u16(in + (long long)(2u * jb) * p.ldi + i0, valid, lo);
  load4_u16(in + (long long)(2u * jb + 1u) * p.ldi + i0, (2u * jb + 1u < (unsigned int)p.n) ? valid : 0u, hi);
  u32x4e o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = lo[e] | (hi[e] << 16);
  const unsigned int nout = ldo - i0 < 4u ? ldo - i0 : 4u;
  const unsigned long long oa = (unsigned long long)(size_t)out;
  if (nout == 4u && (oa & 15ull) == 0ull) *(GM u32x4e*)out = o;
  else if (nout == 4u && (oa & 7ull) == 0ull) { u32x2q a2 = {o[0], o[1]}, b2 = {o[2], o[3]}; *(GM u32x2q*)out = a2; *(GM u32x2q*)(out + 2) = b2; }
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if ((unsigned int)e < nout) out[e] = o[e];
  }
}

// gather / scatter [ref: :1444-1790]; lanes along the contiguous (i) axis where there is one
template <int S>
__global__ __launch_bounds__(256) void gather_scatter_kernel(MeltwArgs p) {
  typedef typename Payload<S>::type T;
  GM const T* in = (GM const T*)((gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0);
  GM T* out = (GM T*)((gptr)p.out + (long long)blockIdx.y * p.bs_out);
  const bool gather = (p.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER);
  const void* idxp = gather ? p.aux_in : (const void*)p.aux_out;
  const bool idx64 = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_8BYTES) != 0;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)p.m * p.n) return;
  const long long i = gid % p.m, j = gid / p.m;
#define XIDX(q) (idx64 ? (long long)((GM const unsigned long long*)idxp)[q] : (long long)((GM const unsigned int*)idxp)[q])
  if (p.flags & LIBXSMM_MELTW_FLAG_UNARY_GS_COLS) {
    if (gather) out[i + j * p.ldo] = in[i + XIDX(j) * p.ldi]; else out[i + XIDX(j) * p.ldo] = in[i + j * p.ldi];
  } else if (p.flags & LIBXSMM_MELTW_FLAG_UNARY_GS_ROWS) {
    if (gather) out[i + j * p.ldo] = in[XIDX(i) + j * p.ldi]; else out[XIDX(i) + j * p.ldo] = in[i + j * p.ldi];
  } else {
    if (gather) out[i + j * p.ldo] = in[XIDX(i + j * p.m)]; else out[XIDX(i + j * p.m)] = in[i + j * p.ldi];
  }
#undef XIDX
}

// GS_OFFS (one linear offset per element), four consecutive elements per thread (round 3): the four offsets arrive as one vector load, the four random
// accesses are in flight together, the contiguous side moves as one vector.  m % 4 == 0, 4-byte offsets 16-byte aligned, contiguous side 4 * S aligned.
template <int S>
__global__ __launch_bounds__(256) void gs_offs_vec4_kernel(MeltwArgs p, unsigned int m4, unsigned int total) {
  typedef typename Payload<S>::type T;
  typedef T T4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4i __attribute__((ext_vector_type(4)));
  const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= total) return;
  const unsigned int j = gid / m4, i = (gid - j * m4) * 4u;
  GM const T* in = (GM const T*)((gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0);
  GM T* out = (GM T*)((gptr)p.out + (long long)blockIdx.y * p.bs_out);
  const bool gather = (p.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER);
  const u32x4i off = *(GM const u32x4i*)((GM const unsigned int*)(gather ? p.aux_in : (const void*)p.aux_out) + (long long)j * p.m + i);
  if (gather) {
    T4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = in[off[e]];
    *(GM T4*)(out + i + (long long)j * p.ldo) = v;
  } else {
    const T4 v = *(GM const T4*)(in + i + (long long)j * p.ldi);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[off[e]] = v[e];
  }
}

// whole-column gather / scatter (GS_COLS) with 16-byte accesses: a thread moves 16 bytes of one column; the column index is
// read once per thread (same address across the lanes of a column segment -> broadcast).  Needs (m * S) % 16 == 0, 16-byte
// aligned bases and leading dimensions that keep every column 16-byte aligned.
// (round 6, not adopted: a workgroup per 16 KiB run of one column, the index read once per thread and four loads in flight before the first store: 54.0 against 52.4 us
//  on 8192 columns of 16 KiB out of 16 384, profiles/r06_gather_run.jsonl)
__global__ __launch_bounds__(256) void gather_cols_vec_kernel(MeltwArgs p, int elem_size, unsigned int vpc, unsigned int total) {
  const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= total) return;
  const unsigned int v = gid % vpc, j = gid / vpc;
  gcptr in = (gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0;
  gptr out = (gptr)p.out + (long long)blockIdx.y * p.bs_out;
  const bool gather = (p.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER);
  const void* idxp = gather ? p.aux_in : (const void*)p.aux_out;
  const long long c = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_8BYTES) ? (long long)((GM const unsigned long long*)idxp)[j] : (long long)((GM const unsigned int*)idxp)[j];
  const long long src = (gather ? c * p.ldi : (long long)j * p.ldi) * elem_size + 16ll * v;
  const long long dst = (gather ? (long long)j * p.ldo : c * p.ldo) * elem_size + 16ll * v;
  *(GM u32x4e*)(out + dst) = *(GM const u32x4e*)(in + src);
}

// Row gather / scatter (GS_ROWS) through LDS: the direct form reads (gather) or writes (scatter) one random 4-byte word per lane -- 64
// cache lines per wave instruction, 0.20 of the HBM roofline.  Here a workgroup stages one whole source (gather) or destination (scatter)
// column in LDS with coalesced 16-byte accesses and does the random indexing THERE: HBM sees two streams.
//   gather : lds[0 .. ldi) = in[.. + j*ldi];  out[i + j*ldo] = lds[idx[i]]
//   scatter: lds = out column j (read-modify-write: rows that no index names keep their value);  lds[idx[i]] = in[i + j*ldi];  column written back
// `rows` = staged extent (ldi for gather, ldo for scatter), a multiple of 16 / S elements; used when at least half of the staged rows are touched.
// NC columns per workgroup (round 4; gather only): every workgroup of the one-column form reads the whole index list again -- as many bytes through L1 as the column
// it stages -- so a workgroup now stages NC columns (NC * rows * S <= 64 KiB) and a thread uses each index it reads for all of them.
template <int S, int NC>
__global__ __launch_bounds__(256) void gs_rows_lds_multi_kernel(MeltwArgs p, int rows) {
  typedef typename Payload<S>::type T;
  extern __shared__ __attribute__((aligned(16))) unsigned char gs_lds[];
  T* col = (T*)gs_lds;
  GM const T* in = (GM const T*)((gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0);
  GM T* out = (GM T*)((gptr)p.out + (long long)blockIdx.y * p.bs_out);
  const void* idxp = p.aux_in;
  const bool idx64 = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_8BYTES) != 0;
  const long long j0 = (long long)blockIdx.x * NC;
  const int nvec = rows * S / 16;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (j0 + c < p.n) {
      GM const u32x4e* src = (GM const u32x4e*)(in + (j0 + c) * p.ldi);
      for (int v = threadIdx.x; v < nvec; v += 256) ((u32x4e*)col)[c * nvec + v] = src[v];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.m; i += 256) {
    const long long r = idx64 ? (long long)((GM const unsigned long long*)idxp)[i] : (long long)((GM const unsigned int*)idxp)[i];
#pragma unroll
    for (int c = 0; c < NC; ++c) if (j0 + c < p.n) out[i + (j0 + c) * p.ldo] = col[(long long)c * rows + r];
  }
}

template <int S>
__global__ __launch_bounds__(256) void gs_rows_lds_kernel(MeltwArgs p, int rows) {
  typedef typename Payload<S>::type T;
  extern __shared__ __attribute__((aligned(16))) unsigned char gs_lds[];
  T* col = (T*)gs_lds;
  GM const T* in = (GM const T*)((gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0);
  GM T* out = (GM T*)((gptr)p.out + (long long)blockIdx.y * p.bs_out);
  const bool gather = (p.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER);
  const void* idxp = gather ? p.aux_in : (const void*)p.aux_out;
  const bool idx64 = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_8BYTES) != 0;
  const long long j = blockIdx.x;
  const int nvec = rows * S / 16;
  GM const u32x4e* src = gather ? (GM const u32x4e*)(in + j * p.ldi) : (GM const u32x4e*)(out + j * p.ldo);
  for (int v = threadIdx.x; v < nvec; v += 256) ((u32x4e*)col)[v] = src[v];
  __syncthreads();
#define XIDX(q) (idx64 ? (long long)((GM const unsigned long long*)idxp)[q] : (long long)((GM const unsigned int*)idxp)[q])
  if (gather) {
    for (int i = threadIdx.x; i < p.m; i += 256) out[i + j * p.ldo] = col[XIDX(i)];
  } else {
    for (int i = threadIdx.x; i < p.m; i += 256) col[XIDX(i)] = in[i + j * p.ldi];
    __syncthreads();
    GM u32x4e* dst = (GM u32x4e*)(out + j * p.ldo);
    for (int v = threadIdx.x; v < nvec; v += 256) dst[v] = ((const u32x4e*)col)[v];
  }
#undef XIDX
}

// reductions over rows (collapse i: one wave per column, shuffle tree) or columns (collapse j:
// one thread per row, serial over j -- reads stay coalesced along i) [ref: :1065-1441]
__global__ __launch_bounds__(256) void reduce_kernel(MeltwArgs p) {
  const bool rows = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) != 0;
  const bool init_acc = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_INIT_ACC) != 0;
  const int type = p.type;
  const bool want_x = type != LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD;
  const bool want_x2 = type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD || type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD;
  const bool is_add = type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD || want_x2;
  gcptr in = (gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0;
  gptr out = (gptr)p.out + (long long)blockIdx.y * p.bs_out;
  const long long result_size = rows ? p.n : p.ldo;
  gptr out2 = (want_x && want_x2) ? out + result_size * mw_size(p.out_type) : out;
  const float ident = (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) ? -3.402823466e+38f : (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN) ? 3.402823466e+38f : 0.0f;
  auto combine = [&](float a, float x) {
    if (is_add) return a + x;
    if (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) return (a < x) ? x : a;
    if (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN) return (a > x) ? x : a;
    return fmaxf(fabsf(a), fabsf(x));
  };
  if (rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= p.n) return;
    float sx = ident, sx2 = 0.0f;
    for (int i = lane; i < p.m; i += 64) {
      const float x = mw_load(in, i + (long long)j * p.ldi, p.in0_type);
      sx = combine(sx, x); sx2 += x * x;
    }
    for (int off = 32; off > 0; off >>= 1) { sx = combine(sx, __shfl_xor(sx, off)); sx2 += __shfl_xor(sx2, off); }
    if (lane == 0) {
      if (is_add && init_acc) { if (want_x) sx += mw_load(out, j, p.out_type); if (want_x2) sx2 += mw_load(out2, j, p.out_type); }
      if (want_x) mw_store(out, j, p.out_type, sx);
      if (want_x2) mw_store(out2, j, p.out_type, sx2);
    }
  } else {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.m) return;
    float sx = ident, sx2 = 0.0f;
    for (int j = 0; j < p.n; ++j) {
      const float x = mw_load(in, i + (long long)j * p.ldi, p.in0_type);
      sx = combine(sx, x); sx2 += x * x;
    }
    if (is_add && init_acc) { if (want_x) sx += mw_load(out, i, p.out_type); if (want_x2) sx2 += mw_load(out2, i, p.out_type); }
    if (want_x) mw_store(out, i, p.out_type, sx);
    if (want_x2) mw_store(out2, i, p.out_type, sx2);
  }
}

// Vector form of the reductions (m % 4 == 0, ldi % 4 == 0, 16-byte (f32) / 8-byte (bf16) aligned input).
//   REDUCE_ROWS (one result per column): a group of G = min(64, pow2(m/4)) lanes owns a column, every lane adds 4-element
//     vectors with stride G, the group is folded with xor-shuffles; a wave covers 64/G columns (small m keeps all lanes busy).
//   REDUCE_COLS (one result per 4 rows): a thread owns 4 consecutive rows and, when n >= 256, one of 16 column slices
//     (columns slice, slice+16, ...); the 16 partial vectors are combined in slice order through LDS.  For n < 256 there is
//     one slice and the sum runs in column order, bit-identical to the general kernel and the oracle.
template <bool BF16IN>
__device__ __forceinline__ void red_load4(float (&x)[4], gcptr in, long long idx) {
  if (BF16IN) { const u16x4 v = *(GM const u16x4*)((GM const unsigned short*)in + idx); for (int e = 0; e < 4; ++e) x[e] = mw_bf2f(v[e]); }
  else { const f32x4 v = *(GM const f32x4*)((GM const float*)in + idx); for (int e = 0; e < 4; ++e) x[e] = v[e]; }
}
// RGL: row groups per block of the REDUCE_COLS form (16: 16 column slices per block, tiles; 64: 4 slices, whole 1 KiB row segments per wave -- the
// two-pass form over one big matrix).  CPG: columns a lane group of the REDUCE_ROWS form handles per trip (4 when a column is ONE vector per lane).
template <bool BF16IN, int RGL = 16, int CPG = 1>
__global__ __launch_bounds__(256) void reduce_vec_kernel(MeltwArgs p, int G, int slices, int chunk, float* partial) {
  constexpr int NSL = 256 / RGL;
  __shared__ float part[2][NSL][RGL][4];
  const bool rows = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) != 0;
  const bool init_acc = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_INIT_ACC) != 0;
  const int type = p.type;
  const bool want_x = type != LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD;
  const bool want_x2 = type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD || type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD;
  const bool is_add = type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD || want_x2;
  gcptr in = (gcptr)p.in0 + (long long)blockIdx.y * p.bs_in0;
  gptr out = (gptr)p.out + (long long)blockIdx.y * p.bs_out;
  const long long result_size = rows ? p.n : p.ldo;
  gptr out2 = (want_x && want_x2) ? out + result_size * mw_size(p.out_type) : out;
  const float ident = (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) ? -3.402823466e+38f : (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN) ? 3.402823466e+38f : 0.0f;
  auto combine = [&](float a, float x) {
    if (is_add) return a + x;
    if (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) return (a < x) ? x : a;
    if (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN) return (a > x) ? x : a;
    return fmaxf(fabsf(a), fabsf(x));
  };
  const int m4 = p.m / 4;
  if (rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane % G, cpw = 64 / G;
    if constexpr (CPG > 1) {
      // short columns (m / 4 <= G: one vector per lane and column): CPG columns per lane group with all their loads in flight, then CPG folds
      const int j0 = ((blockIdx.x * 4 + wave) * cpw + lane / G) * CPG;
      float x[CPG][4];
#pragma unroll
      for (int u = 0; u < CPG; ++u) {
        if (j0 + u < p.n && l < m4) red_load4<BF16IN>(x[u], in, 4ll * l + (long long)(j0 + u) * p.ldi);
        else { x[u][0] = x[u][1] = x[u][2] = x[u][3] = ident; }
      }
#pragma unroll
      for (int u = 0; u < CPG; ++u) {
        float sx = ident, sx2 = 0.0f;
        if (l < m4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { sx = combine(sx, x[u][e]); sx2 += x[u][e] * x[u][e]; }
        }
        for (int off = G >> 1; off > 0; off >>= 1) { sx = combine(sx, __shfl_xor(sx, off)); sx2 += __shfl_xor(sx2, off); }
        const int j = j0 + u;
        if (l == 0 && j < p.n) {
          if (is_add && init_acc) { if (want_x) sx += mw_load(out, j, p.out_type); if (want_x2) sx2 += mw_load(out2, j, p.out_type); }
          if (want_x) mw_store(out, j, p.out_type, sx);
          if (want_x2) mw_store(out2, j, p.out_type, sx2);
        }
      }
      return;
    }
    const int j = (blockIdx.x * 4 + wave) * cpw + lane / G;
    float sx = ident, sx2 = 0.0f;
    if (j < p.n) {
      // four independent loads in flight per lane; the values are folded in the order of the plain loop (same rounding)
      int i4 = l;
      for (; i4 + 15 * G < m4; i4 += 16 * G) {      // a long column: sixteen vectors (16 KiB per wave) in flight, folded in the same order as below
        float x[16][4];
#pragma unroll
        for (int u = 0; u < 16; ++u) red_load4<BF16IN>(x[u], in, 4ll * (i4 + u * G) + (long long)j * p.ldi);
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { sx = combine(sx, x[u][e]); sx2 += x[u][e] * x[u][e]; }
      }
      for (; i4 + 3 * G < m4; i4 += 4 * G) {
        float x[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) red_load4<BF16IN>(x[u], in, 4ll * (i4 + u * G) + (long long)j * p.ldi);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { sx = combine(sx, x[u][e]); sx2 += x[u][e] * x[u][e]; }
      }
      for (; i4 < m4; i4 += G) {
        float x[4]; red_load4<BF16IN>(x, in, 4ll * i4 + (long long)j * p.ldi);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sx = combine(sx, x[e]); sx2 += x[e] * x[e]; }
      }
    }
    for (int off = G >> 1; off > 0; off >>= 1) { sx = combine(sx, __shfl_xor(sx, off)); sx2 += __shfl_xor(sx2, off); }
    if (l == 0 && j < p.n) {
      if (is_add && init_acc) { if (want_x) sx += mw_load(out, j, p.out_type); if (want_x2) sx2 += mw_load(out2, j, p.out_type); }
      if (want_x) mw_store(out, j, p.out_type, sx);
      if (want_x2) mw_store(out2, j, p.out_type, sx2);
    }
  } else {
    const int rg_l = threadIdx.x % RGL, sl = threadIdx.x / RGL;          // RGL row groups x NSL slices per block (tiles: 64 x 4 measured 1.5x slower than 16 x 16)
    const int rg = blockIdx.x * RGL + rg_l;
    float sx[4] = {ident, ident, ident, ident}, sx2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // two-pass form (partial != NULL): blockIdx.z owns columns [z*chunk, (z+1)*chunk) and writes raw partial sums
    const int jbeg = partial ? (int)blockIdx.z * chunk : 0, jend = partial ? ((jbeg + chunk < p.n) ? jbeg + chunk : p.n) : p.n;
    if (rg < m4 && sl < slices) {
      int j = jbeg + sl;
      // sixteen columns in flight per thread first (a wave then has 16 KiB on its way: with four, 2048 waves of the two-pass form kept 8 MB in flight and ran at
      // 4.2 TB/s -- the round trip, not the memory system), folded in column order like the loops below: the same sums
      for (; j + 15 * slices < jend; j += 16 * slices) {
        float x[16][4];
#pragma unroll
        for (int u = 0; u < 16; ++u) red_load4<BF16IN>(x[u], in, 4ll * rg + (long long)(j + u * slices) * p.ldi);
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { sx[e] = combine(sx[e], x[u][e]); sx2[e] += x[u][e] * x[u][e]; }
      }
      for (; j + 3 * slices < jend; j += 4 * slices) {       // four columns in flight, folded in column order
        float x[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) red_load4<BF16IN>(x[u], in, 4ll * rg + (long long)(j + u * slices) * p.ldi);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { sx[e] = combine(sx[e], x[u][e]); sx2[e] += x[u][e] * x[u][e]; }
      }
      for (; j < jend; j += slices) {
        float x[4]; red_load4<BF16IN>(x, in, 4ll * rg + (long long)j * p.ldi);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sx[e] = combine(sx[e], x[e]); sx2[e] += x[e] * x[e]; }
      }
    }
    if (slices > 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { part[0][sl][rg_l][e] = sx[e]; part[1][sl][rg_l][e] = sx2[e]; }
      __syncthreads();
      if (sl != 0) return;
      for (int s2 = 1; s2 < slices; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sx[e] = combine(sx[e], part[0][s2][rg_l][e]); sx2[e] += part[1][s2][rg_l][e]; }
    } else if (sl != 0) return;
    if (rg >= m4) return;
    if (partial) {
      GM float* px = (GM float*)partial + ((long long)blockIdx.z * 2) * p.m + 4ll * rg;
#pragma unroll
      for (int e = 0; e < 4; ++e) { px[e] = sx[e]; if (want_x2) px[p.m + e] = sx2[e]; }
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long i = 4ll * rg + e;
      float a = sx[e], b = sx2[e];
      if (is_add && init_acc) { if (want_x) a += mw_load(out, i, p.out_type); if (want_x2) b += mw_load(out2, i, p.out_type); }
      if (want_x) mw_store(out, i, p.out_type, a);
      if (want_x2) mw_store(out2, i, p.out_type, b);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Reduction over a LIST of columns -- REDUCE_COLS_IDX_OP_ADD / MAX / MIN: out[i] = op over jj < n_cols of in(i, idx[jj]) (an embedding
// bag: gather + reduce in one pass) -- and the column reductions MAX / ABSMAX / MIN that record the column of the extremum
// (REDUCE_RECORD_ARGOP) [ref: mateltwise ref :1346-1430].  One thread per row i, the listed columns in the caller's order (the sum is
// the reference's serial f32 sum; a later equal extremum wins like there); a wave reads 64 consecutive rows of a column: 256 bytes.
// p.scalar_u64 = number of listed columns (0: all p.n columns in order), p.aux_in = the list, p.aux_out = the recorded columns.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_cols_listed_kernel(MeltwArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.m) return;
  const unsigned int b = blockIdx.y;
  gcptr in = (gcptr)p.in0 + (long long)b * p.bs_in0;
  gptr out = (gptr)p.out + (long long)b * p.bs_out;
  const bool idx4 = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_4BYTES) != 0, record = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_RECORD_ARGOP) != 0;
  const bool listed = p.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_ADD || p.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MAX || p.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MIN;
  const int op = (p.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_ADD) ? 0
               : (p.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MAX || p.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) ? 1
               : (p.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX) ? 3 : 2;
  const unsigned long long n_cols = listed ? p.scalar_u64 : (unsigned long long)p.n;
  GM const unsigned int* idx32 = (GM const unsigned int*)((gcptr)p.aux_in + (long long)b * p.bs_aux);
  GM const unsigned long long* idx64 = (GM const unsigned long long*)((gcptr)p.aux_in + (long long)b * p.bs_aux);
  float acc = (op == 0) ? 0.0f : (op == 1) ? -3.402823466e+38f : (op == 3) ? 0.0f : 3.402823466e+38f;
  unsigned long long arg = 0; bool found = false;
  for (unsigned long long jj = 0; jj < n_cols; ++jj) {
    const unsigned long long j = listed ? (idx4 ? (unsigned long long)idx32[jj] : idx64[jj]) : jj;     // wave-uniform: one scalar load
    float x = mw_load(in, (long long)i + (long long)j * p.ldi, p.in0_type);
    if (op == 0) { acc += x; continue; }
    if (op == 3) x = fabsf(x);
    if (op == 1 || op == 3) {
      if (record) { if (x >= acc) { acc = x; arg = j; found = true; } }
      else acc = (x < acc) ? acc : x;
    } else {
      if (record) { if (x <= acc) { acc = x; arg = j; found = true; } }
      else acc = (x < acc) ? x : acc;
    }
  }
  mw_store(out, i, p.out_type, acc);
  if (record && found) {          // (no listed column qualified -- all NaN -- : the entry keeps its old value, like the reference)
    if (idx4) ((GM unsigned int*)p.aux_out)[i] = (unsigned int)arg; else ((GM unsigned long long*)p.aux_out)[i] = arg;
  }
}

// ------------------------------------------------------------------------------------------------
// DROPOUT [ref: mateltwise ref :2361-2407, generator :43-72].  The reference keeps 16 xoshiro128+ streams side by side (state word s of
// stream l at state[l + 16 s]) and draws for 16 rows at a time (the AVX-512 width; the width is part of the semantics):
// row i of column j of batch element b gets draw number g = (b * n + j) * ceil(m / 16) + i / 16 of stream i % 16, and every draw
// advances all 16 streams, also in a column's ragged tail.  A stream is a linear map over GF(2): draw g needs T^g state, and T^(2^k) is
// a 128 x 128 bit matrix that the host builds once (jump[k][bit] = T^(2^k) e_bit, 64 KiB).  So the draws are cut into SEGMENTS of L
// consecutive g: 16 threads (one per stream) jump to the segment's first draw -- one matrix-vector product per set bit of g -- and
// step through it; the segment that ends the launch writes the advanced state back, exactly the reference's state after its last draw.
// The bitmask comes from a wave ballot: the 16 streams of a segment are 16 adjacent lanes = two mask bytes.
// p.aux_in = the 64-dword state, p.aux_out = the mask (BITMASK_2BYTEMULT), p.scalar_f32 = the dropout probability.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void dropout_kernel(MeltwArgs p, const u32x4m* jump_tables, unsigned long long L) {
  const unsigned long long t = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  const unsigned int l = (unsigned int)(t & 15ull);
  const unsigned long long seg = t >> 4;
  const unsigned long long cpr = (unsigned long long)((p.m + 15) / 16), C = (unsigned long long)p.n * cpr, total = C * p.nbatch;
  const unsigned long long g0 = seg * L;
  if (g0 >= total) return;
  const unsigned long long g1 = (g0 + L < total) ? g0 + L : total;
  // read from the launch's snapshot of the state (p.ws, copied in stream order before the launch), written back to the caller's buffer:
  // no workgroup can observe the state the stream's last segment has already advanced
  GM unsigned int* st = (GM unsigned int*)p.aux_in;
  GM const unsigned int* st_in = (GM const unsigned int*)p.ws;
  GM const u32x4m* jump = (GM const u32x4m*)jump_tables;
  u32x4m s = {st_in[l], st_in[l + 16], st_in[l + 32], st_in[l + 48]};
  for (int k = 0; k < 64 && (g0 >> k) != 0ull; ++k) {              // s = T^g0 s
    if (!((g0 >> k) & 1ull)) continue;
    GM const u32x4m* col = jump + 128 * k;
    u32x4m acc = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      unsigned int bits = s[w];
      while (bits) { const int b = __builtin_ctz(bits); bits &= bits - 1u; acc ^= col[32 * w + b]; }
    }
    s = acc;
  }
  const float pn = 1.0f - p.scalar_f32, pi = 1.0f / pn;
  const bool bitm = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) != 0;
  const long long mask_ld8 = (((long long)p.ldo + 15) / 16) * 2;    // mask bytes per column
  const unsigned int shift = (threadIdx.x & 63u) & ~15u;             // this segment's 16 lanes inside the wave
  for (unsigned long long g = g0; g < g1; ++g) {
    const float draw = __uint_as_float(0x3f800000u | ((s[3] + s[0]) >> 9)) - 1.0f;
    { const unsigned int t0 = s[1] << 9; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t0; s[3] = (s[3] << 11) | (s[3] >> 21); }
    const unsigned long long b = g / C, c = g - b * C, j = c / cpr, ic = c - j * cpr;
    const long long i = (long long)(ic * 16ull + l);
    const bool valid = i < p.m, keep = draw < pn;
    if (valid) {
      const float x = mw_load((gcptr)p.in0 + (long long)b * p.bs_in0, i + (long long)j * p.ldi, p.in0_type);
      mw_store((gptr)p.out + (long long)b * p.bs_out, i + (long long)j * p.ldo, p.out_type, keep ? pi * x : 0.0f);
    }
    if (bitm) {
      const unsigned int kept = (unsigned int)((__ballot(valid && keep) >> shift) & 0xffffull), have = (unsigned int)((__ballot(valid) >> shift) & 0xffffull);
      if (l == 0) {
        GM unsigned char* mb = (GM unsigned char*)p.aux_out + (long long)b * p.bs_aux + (long long)j * mask_ld8 + (long long)ic * 2;
        if (have & 0x00ffu) mb[0] = (unsigned char)((mb[0] & ~(have & 0xffu)) | (kept & 0xffu));
        if (have & 0xff00u) mb[1] = (unsigned char)((mb[1] & ~(have >> 8)) | (kept >> 8));
      }
    }
  }
  if (g1 == total) { st[l] = s[0]; st[l + 16] = s[1]; st[l + 32] = s[2]; st[l + 48] = s[3]; }
}
// DROPOUT_INV [ref: :2408-2424]: out = mask bit ? in / (1 - p) : 0, the mask from in.secondary
__global__ __launch_bounds__(256) void dropout_inv_kernel(MeltwArgs p) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x, per = (long long)p.m * p.n;
  if (e >= per * p.nbatch) return;
  const long long b = e / per, r = e - b * per, j = r / p.m, i = r - j * p.m;
  const bool bitm = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) != 0;
  const long long mask_ld8 = bitm ? (((long long)p.ldi + 15) / 16) * 2 : p.ldi / 8;
  const float pi = 1.0f / (1.0f - p.scalar_f32);
  GM const unsigned char* mb = (GM const unsigned char*)p.aux_in + b * p.bs_aux;
  const int bit = (mb[i / 8 + j * mask_ld8] >> (i % 8)) & 1;
  const float x = mw_load((gcptr)p.in0 + b * p.bs_in0, i + j * p.ldi, p.in0_type) * pi;
  mw_store((gptr)p.out + b * p.bs_out, i + j * p.ldo, p.out_type, bit ? x : 0.0f);
}
// ------------------------------------------------------------------------------------------------
// Stochastic rounding to BF8 (UNARY / BINARY / TERNARY_STOCHASTIC_ROUND) [ref: src/libxsmm_lpflt_quant.c:303-365; mateltwise ref
// :2485-2486].  Element e (column-major order of the m x n result) of a call takes one xoshiro128++ draw of stream e % 16 of the same
// 16-stream state DROPOUT uses; every call of a batched launch starts again at e = 0 and continues the streams.  So stream l makes
// cnt(l) = |{e < m n : e % 16 == l}| draws per call and draw d of stream l belongs to call d / cnt(l), element 16 (d % cnt(l)) + l.
// The TPP itself ran into an f32 workspace (p.in0, [call][n][m] dense); this pass jumps every thread to its segment of draws (as in
// dropout_kernel) and rounds; the thread that makes a stream's last draw writes that stream's state back.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stochastic_bf8_kernel(MeltwArgs p, const u32x4m* jump_tables, unsigned long long L) {
  const unsigned long long t = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  const unsigned int l = (unsigned int)(t & 15ull);
  const unsigned long long seg = t >> 4, mn = (unsigned long long)p.m * p.n;
  if (l >= mn) return;
  const unsigned long long cnt = (mn - 1ull - l) / 16ull + 1ull, total = cnt * p.nbatch;
  const unsigned long long d0 = seg * L;
  if (d0 >= total) return;
  const unsigned long long d1 = (d0 + L < total) ? d0 + L : total;
  // the state is READ from the launch's snapshot (p.ws, copied in stream order before the launch) and WRITTEN to the caller's buffer: a
  // workgroup that starts late can never load what the segment ending the stream has already advanced
  GM unsigned int* st = (GM unsigned int*)p.aux_in;
  GM const unsigned int* st_in = (GM const unsigned int*)p.ws;
  GM const u32x4m* jump = (GM const u32x4m*)jump_tables;
  u32x4m s = {st_in[l], st_in[l + 16], st_in[l + 32], st_in[l + 48]};
  for (int k = 0; k < 64 && (d0 >> k) != 0ull; ++k) {
    if (!((d0 >> k) & 1ull)) continue;
    GM const u32x4m* col = jump + 128 * k;
    u32x4m acc = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      unsigned int bits = s[w];
      while (bits) { const int b = __builtin_ctz(bits); bits &= bits - 1u; acc ^= col[32 * w + b]; }
    }
    s = acc;
  }
  for (unsigned long long d = d0; d < d1; ++d) {
    const unsigned int sum = s[0] + s[3], vrng = ((sum << 7) | (sum >> 25)) + s[0];
    { const unsigned int t0 = s[1] << 9; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t0; s[3] = (s[3] << 11) | (s[3] >> 21); }
    const unsigned long long b = d / cnt, e = 16ull * (d - b * cnt) + l, j = e / (unsigned long long)p.m, i = e - j * (unsigned long long)p.m;
    const float x = ((GM const float*)p.in0)[b * mn + e];
    unsigned short h = __builtin_bit_cast(unsigned short, (_Float16)x);
    const unsigned short rnd = (unsigned short)((vrng >> 24) & 0xffu), fixup = (unsigned short)((h >> 8) & 1u);
    if ((h & 0x7c00u) == 0x7c00u) h = ((h & 0x03ffu) == 0) ? h : (unsigned short)(h | 0x0200u);
    else if ((h & 0x7c00u) == 0) h = (unsigned short)(h + 0x007fu + fixup);
    else h = (unsigned short)(h + rnd);
    ((GM unsigned char*)p.out)[(long long)b * p.bs_out + (long long)j * p.ldo + (long long)i] = (unsigned char)(h >> 8);
  }
  if (d1 == total) { st[l] = s[0]; st[l + 16] = s[1]; st[l + 32] = s[2]; st[l + 48] = s[3]; }
}
// T^(2^k) as 128 columns each, k < 64, on the current device (built once per device)
static const u32x4m* dropout_jump_tables() {
  static std::mutex mu; static std::unordered_map<int, const u32x4m*> per_device;
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto it = per_device.find(dev);
  if (it != per_device.end()) return it->second;
  struct S { unsigned int w[4]; };
  auto step = [](S s) { const unsigned int t0 = s.w[1] << 9; s.w[2] ^= s.w[0]; s.w[3] ^= s.w[1]; s.w[1] ^= s.w[2]; s.w[0] ^= s.w[3]; s.w[2] ^= t0; s.w[3] = (s.w[3] << 11) | (s.w[3] >> 21); return s; };
  std::vector<S> tab(64 * 128);
  for (int b = 0; b < 128; ++b) { S e{{0, 0, 0, 0}}; e.w[b / 32] = 1u << (b % 32); tab[b] = step(e); }      // T
  auto apply = [&](const S* cols, S v) { S a{{0, 0, 0, 0}}; for (int b = 0; b < 128; ++b) if ((v.w[b / 32] >> (b % 32)) & 1u) for (int q = 0; q < 4; ++q) a.w[q] ^= cols[b].w[q]; return a; };
  for (int k = 1; k < 64; ++k) for (int b = 0; b < 128; ++b) tab[128 * k + b] = apply(&tab[128 * (k - 1)], tab[128 * (k - 1) + b]);     // T^(2^k) = (T^(2^(k-1)))^2
  void* d = nullptr;
  if (hipMalloc(&d, tab.size() * sizeof(S)) != hipSuccess || hipMemcpy(d, tab.data(), tab.size() * sizeof(S), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  per_device[dev] = (const u32x4m*)d;
  return (const u32x4m*)d;
}

// BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD: out[0] = sum_ij in0(i, j) * in1(i, j) [ref: mateltwise ref :2523-2542] -- the dot product the
// layernorm-backward equations end in.  One workgroup per batch element: 1024 partial sums (element e goes to thread e % 1024), folded
// pairwise in LDS: deterministic, a different (tree) order than the reference's serial sum.
__global__ __launch_bounds__(1024) void mul_reduce_scalar_kernel(MeltwArgs p) {
  __shared__ float part[1024];
  const unsigned int b = blockIdx.x;
  gcptr in0 = (gcptr)p.in0 + (long long)b * p.bs_in0; gcptr in1 = (gcptr)p.in1 + (long long)b * p.bs_in1;
  const int bc0 = bcast_kind(p.operation, p.type, p.flags, 0), bc1 = bcast_kind(p.operation, p.type, p.flags, 1);
  const long long total = (long long)p.m * p.n;
  float acc = 0.0f;
  for (long long e = threadIdx.x; e < total; e += 1024) {
    const long long j = e / p.m, i = e - j * p.m;
    acc += mw_load(in0, bc_index(bc0, i, j, p.ldi), p.in0_type) * mw_load(in1, bc_index(bc1, i, j, p.ldi1), p.in1_type);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) { if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) mw_store((gptr)p.out + (long long)b * p.bs_out, 0, p.out_type, part[0]);
}

// UNARY_REDUCE_TO_SCALAR_OP_ADD: out[0] = sum_ij in(i, j) [ref: mateltwise ref :2097-2116] -- the same scheme (1024 partial sums folded pairwise: deterministic, a
// tree order instead of the reference's serial one); all-f64 descriptors accumulate in double as the reference does.
template <typename T>
__global__ __launch_bounds__(1024) void reduce_scalar_kernel(MeltwArgs p) {
  __shared__ T part[1024];
  const unsigned int b = blockIdx.x;
  gcptr in0 = (gcptr)p.in0 + (long long)b * p.bs_in0;
  const int bc0 = bcast_kind(p.operation, p.type, p.flags, 0);
  const long long total = (long long)p.m * p.n;
  T acc = (T)0;
  for (long long e = threadIdx.x; e < total; e += 1024) {
    const long long j = e / p.m, i = e - j * p.m;
    if constexpr (sizeof(T) == 8) acc += ((GM const double*)in0)[bc_index(bc0, i, j, p.ldi)];
    else acc += mw_load(in0, bc_index(bc0, i, j, p.ldi), p.in0_type);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) { if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) {
    if constexpr (sizeof(T) == 8) ((GM double*)((gptr)p.out + (long long)b * p.bs_out))[0] = part[0];
    else mw_store((gptr)p.out + (long long)b * p.bs_out, 0, p.out_type, part[0]);
  }
}

// UNARY_REDUCE_X_OP_ADD_NCNC_FORMAT [ref: mateltwise ref :2118-2141]: the input is a blocked [N / bn][C / bc][bn][bc] tensor (bc = the shape's m, bn = its n, C = its
// ldi, N = its ldo); out[c] = sum over all N of channel c, added in the reference's order (blocks of bn rows, rows inside a block) -- one thread per channel, so the
// sum is BIT-IDENTICAL to the reference's; consecutive threads read consecutive addresses (ic is the innermost index).
__global__ __launch_bounds__(256) void reduce_ncnc_kernel(MeltwArgs p) {
  const int bc = p.m, bn = p.n, C = p.ldi, N = p.ldo;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= (C / bc) * bc) return;
  const unsigned int b = blockIdx.y;
  gcptr in = (gcptr)p.in0 + (long long)b * p.bs_in0;
  const int iC = c / bc, ic = c - iC * bc;
  float tmp = 0.0f;
  for (int iN = 0; iN < N / bn; ++iN) {
    const long long base = (long long)iN * C * bn + (long long)iC * bn * bc + ic;
    for (int i_n = 0; i_n < bn; ++i_n) tmp += mw_load(in, base + (long long)i_n * bc, p.in0_type);
  }
  mw_store((gptr)p.out + (long long)b * p.bs_out, c, p.out_type, tmp);
}

// second pass of the two-pass column reduction: partial[z][2][m] -> out (chunks combined in order z = 0, 1, ...)
__global__ __launch_bounds__(256) void reduce_combine_kernel(MeltwArgs p, const float* partial, int nchunks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.m) return;
  const bool init_acc = (p.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_INIT_ACC) != 0;
  const int type = p.type;
  const bool want_x = type != LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD;
  const bool want_x2 = type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD || type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD;
  const bool is_add = type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD || want_x2;
  GM const float* px = (GM const float*)partial + i;
  float a = px[0], b = want_x2 ? px[p.m] : 0.0f;
  auto fold = [&](float x, float x2) {
    if (is_add) a += x; else if (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX) a = (a < x) ? x : a; else if (type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN) a = (a > x) ? x : a; else a = fmaxf(fabsf(a), fabsf(x));
    b += x2;
  };
  // sixteen chunks' partial sums in flight per thread, folded in chunk order (the loop used to ask for one pair at a time: with 128 chunks the 16 workgroups of this
  // kernel spent 30 us waiting for 256 dependent round trips each -- why "few, long-running blocks" won the first pass's block-count scan)
  int z = 1;
  for (; z + 15 < nchunks; z += 16) {
    float x[16], x2[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { x[u] = px[(long long)(z + u) * 2 * p.m]; x2[u] = want_x2 ? px[((long long)(z + u) * 2 + 1) * p.m] : 0.0f; }
#pragma unroll
    for (int u = 0; u < 16; ++u) fold(x[u], x2[u]);
  }
  for (; z < nchunks; ++z) fold(px[(long long)z * 2 * p.m], want_x2 ? px[((long long)z * 2 + 1) * p.m] : 0.0f);
  gptr out = (gptr)p.out;
  gptr out2 = (want_x && want_x2) ? out + (long long)p.ldo * mw_size(p.out_type) : out;
  if (is_add && init_acc) { if (want_x) a += mw_load(out, i, p.out_type); if (want_x2) b += mw_load(out2, i, p.out_type); }
  if (want_x) mw_store(out, i, p.out_type, a);
  if (want_x2) mw_store(out2, i, p.out_type, b);
}

// ------------------------------------------------------------------------------------------------
// host-side selection
// ------------------------------------------------------------------------------------------------

// ---- QUANT to a microscaling type: 32 consecutive rows of a column share one E8M0 scale byte ---------------------------------
// [ref: samples/eltwise/eltwise_unary_quantization_to_mxfp4.c:20-105 and _to_mxbf8.c:22-71 (the drivers' gold code for the
//  reference's bf16 -> MXFP4X2 / MXBF8 QUANT TPP)]: amax over the block (NaN sticks), scale exponent = exponent(amax) - emax_elem
// (2 for E2M1, 15 for E5M2) clamped at 0, an inf / NaN block stores scale 0xFF and max-normal elements; otherwise every element
// is divided by 2^(scale - 127) -- an exact rescaling -- and rounded to nearest even into its format.  out.primary: the elements
// (MXFP4X2: two per byte, even row in the low nibble, ldo / 2 bytes per column; MXBF8: ldo bytes per column), out.secondary: the
// scales, ldo / 32 per column.  One thread per block: 64 B (bf16) in, 16 or 32 B + 1 B out.
__device__ __forceinline__ unsigned int e2m1_rne(float a) {        // |value| -> 3-bit code of {0, .5, 1, 1.5, 2, 3, 4, 6}, ties to even codes
  if (a != a || a > 5.0f) return 7u;
  if (a >= 3.5f) return 6u;
  if (a > 2.5f) return 5u;
  if (a >= 1.75f) return 4u;
  if (a > 1.25f) return 3u;
  if (a >= 0.75f) return 2u;
  if (a > 0.25f) return 1u;
  return 0u;
}
template <bool FP4>
__global__ __launch_bounds__(256) void mx_quant_kernel(MeltwArgs p, unsigned int mblk, unsigned int total) {
  const unsigned int t = blockIdx.x * 256u + threadIdx.x;
  if (t >= total) return;
  const unsigned int b = t % mblk, j = (t / mblk) % (unsigned int)p.n, z = t / (mblk * (unsigned int)p.n);
  gcptr in = (gcptr)p.in0 + (long long)z * p.bs_in0;
  float x[32]; float amax = 0.0f;
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    x[e] = mw_load(in, (long long)j * p.ldi + (long long)b * 32 + e, p.in0_type);
    const float a = fabsf(x[e]);
    if (a > amax || a != a) amax = a;
  }
  int se = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  const bool special = se == 0xff;
  se -= FP4 ? 2 : 15;
  if (special) se = 0xff;
  if (se < 0) se = 0;
  ((GM unsigned char*)p.aux_out)[(long long)z * p.bs_aux + (long long)j * (p.ldo / 32) + b] = (unsigned char)se;
  unsigned int w[8];
  if (FP4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned int word = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = x[8 * q + e];
        const unsigned int code = special ? 7u : (((__float_as_uint(v) >> 31) << 3) | e2m1_rne(fabsf(ldexpf(v, 127 - se))));
        word |= code << (4 * e);
      }
      w[q] = word;
    }
    GM unsigned int* o = (GM unsigned int*)((GM char*)p.out + (long long)z * p.bs_out + (long long)j * (p.ldo / 2) + (long long)b * 16);
    if ((((size_t)o) & 3) == 0) { for (int q = 0; q < 4; ++q) o[q] = w[q]; }
    else { GM unsigned char* ob = (GM unsigned char*)o; for (int q = 0; q < 16; ++q) ob[q] = (unsigned char)(w[q >> 2] >> (8 * (q & 3))); }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      unsigned int word = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned int code = special ? 0x7bu : (unsigned int)lowp::f16_to_bf8_rne(lowp::f32_to_f16(ldexpf(x[4 * q + e], 127 - se)));
        word |= code << (8 * e);
      }
      w[q] = word;
    }
    GM unsigned int* o = (GM unsigned int*)((GM char*)p.out + (long long)z * p.bs_out + (long long)j * p.ldo + (long long)b * 32);
    if ((((size_t)o) & 3) == 0) { for (int q = 0; q < 8; ++q) o[q] = w[q]; }
    else { GM unsigned char* ob = (GM unsigned char*)o; for (int q = 0; q < 32; ++q) ob[q] = (unsigned char)(w[q >> 2] >> (8 * (q & 3))); }
  }
}


// ---- QUANT to NVFP4X2: 16 consecutive rows share one E4M3 scale byte, all intermediate arithmetic rounded to bf16 ---------------
// [ref: samples/eltwise/eltwise_unary_quantization_to_nvfp4.c:43-266 (the driver's gold code of the reference's bf16 -> NVFP4X2 QUANT TPP);
//  src/generator_mateltwise_reference_impl.c:1947, :2274-2300]: scale = E4M3(bf16(bf16(amax) * bf16(1/6))), a saturating E4M3 whose top
// code is 0x78; elements = E2M1(bf16(x * bf16(1 / scale))) with the sign of x; a zero scale stores zeros.
__device__ __forceinline__ float nv_bf16(float x) { return mw_bf2f(mw_f2bf(x)); }
__device__ __forceinline__ unsigned int nv_e4m3_of(float val) {
  const unsigned int u = __float_as_uint(val), sign = u >> 31, fe = (u >> 23) & 0xffu, fm = u & 0x7fffffu;
  if (fe == 0xffu && fm != 0u) return (sign << 7) | 0x7fu;
  if (fe == 0xffu || fabsf(val) > 448.0f) return (sign << 7) | 0x78u;
  if (fe == 0u) return sign << 7;                                   // zero and f32 subnormals
  const int ub = (int)fe - 127;
  if (ub > 8) return (sign << 7) | 0x78u;
  if (ub < -9) return sign << 7;
  if (ub >= -6) {                                                    // normal: 23 -> 3 mantissa bits, nearest even
    unsigned int e = (unsigned int)(ub + 7), tm = fm >> 20;
    if (((fm >> 19) & 1u) && ((fm & 0x7ffffu) || (tm & 1u))) ++tm;
    if (tm >= 8u) { tm = 0u; ++e; }
    return e >= 0xfu ? ((sign << 7) | 0x78u) : ((sign << 7) | (e << 3) | tm);
  }
  const int shift = -6 - ub;                                         // subnormal: 1.mmm shifted right, nearest even
  if (shift >= 4) return sign << 7;
  const unsigned int full = 8u | ((fm >> 20) & 7u);
  unsigned int tm = full >> shift;
  const unsigned int rb = (full >> (shift - 1)) & 1u, sticky = ((full & ((1u << (shift - 1)) - 1u)) || (fm & 0xfffffu)) ? 1u : 0u;
  if (rb && (sticky || (tm & 1u))) ++tm;
  return tm >= 8u ? ((sign << 7) | 8u) : ((sign << 7) | (tm & 7u));
}
__device__ __forceinline__ float nv_e4m3_value(unsigned int b) {
  const unsigned int sign = (b >> 7) & 1u, ex = (b >> 3) & 0xfu, mant = b & 7u;
  float v;
  if (ex == 0u) v = (float)mant * (1.0f / 512.0f);
  else if (ex == 0xfu && mant != 0u) return __uint_as_float(0x7fc00000u);
  else v = ldexpf(1.0f + (float)mant * 0.125f, (int)ex - 7);
  return sign ? -v : v;
}
__global__ __launch_bounds__(256) void nvfp4_quant_kernel(MeltwArgs p, unsigned int mblk, unsigned int total) {
  const unsigned int t = blockIdx.x * 256u + threadIdx.x;
  if (t >= total) return;
  const unsigned int b = t % mblk, j = (t / mblk) % (unsigned int)p.n, z = t / (mblk * (unsigned int)p.n);
  gcptr in = (gcptr)p.in0 + (long long)z * p.bs_in0;
  float x[16]; float amax = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    x[e] = mw_load(in, (long long)j * p.ldi + (long long)b * 16 + e, p.in0_type);
    const float a = fabsf(x[e]);
    if (a > amax || a != a) amax = a;
  }
  unsigned int sb = 0u; float sf = 0.0f;
  if (amax != 0.0f) { sb = nv_e4m3_of(nv_bf16(nv_bf16(amax) * __uint_as_float(0x3e2a0000u))); sf = nv_e4m3_value(sb); }
  ((GM unsigned char*)p.aux_out)[(long long)z * p.bs_aux + (long long)j * (p.ldo / 16) + b] = (unsigned char)sb;
  unsigned int w[2] = {0u, 0u};
  if (sf != 0.0f) {
    const float rcp = nv_bf16(1.0f / nv_bf16(sf));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const unsigned int code = ((__float_as_uint(x[e]) >> 31) << 3) | e2m1_rne(fabsf(nv_bf16(x[e] * rcp)));
      w[e >> 3] |= code << (4 * (e & 7));
    }
  }
  GM unsigned int* o = (GM unsigned int*)((GM char*)p.out + (long long)z * p.bs_out + (long long)j * (p.ldo / 2) + (long long)b * 8);
  if ((((size_t)o) & 3) == 0) { o[0] = w[0]; o[1] = w[1]; }
  else { GM unsigned char* ob = (GM unsigned char*)o; for (int q = 0; q < 8; ++q) ob[q] = (unsigned char)(w[q >> 2] >> (8 * (q & 3))); }
}

static int payload_size(int t) { return typesize(t); }


static int xform_mode(int type, int* v) {
  switch (type) {
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2_PAD: *v = 2; return XF_NORM_TO_VNNI;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4_PAD: *v = 4; return XF_NORM_TO_VNNI;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8_PAD: *v = 8; return XF_NORM_TO_VNNI;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI2_TO_VNNI2T: *v = 2; return XF_VNNI_TO_VNNIT;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_VNNI4T: *v = 4; return XF_VNNI_TO_VNNIT;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI8_TO_VNNI8T: *v = 8; return XF_VNNI_TO_VNNIT;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI2T: *v = 2; return XF_NORM_TO_VNNIT;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI4T: *v = 4; return XF_NORM_TO_VNNIT;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_VNNI8T: *v = 8; return XF_NORM_TO_VNNIT;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI2T_TO_NORM: *v = 2; return XF_VNNIT_TO_NORM;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4T_TO_NORM: *v = 4; return XF_VNNIT_TO_NORM;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI8T_TO_NORM: *v = 8; return XF_VNNIT_TO_NORM;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_NORM: *v = 4; return XF_VNNI4_TO_NORM;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_VNNI4_TO_VNNI2: *v = 4; return XF_VNNI4_TO_VNNI2;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADM_MOD2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADM_MOD4: *v = 1; return XF_PAD;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADN_MOD2: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADNM_MOD2: *v = 2; return XF_PAD;
    case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADN_MOD4: case LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_PADNM_MOD4: *v = 4; return XF_PAD;
    default: *v = 0; return 0;
  }
}
static bool is_reduce_type(int t) {
  return t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD ||
         t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX;
}

static bool is_reduce_cols_idx_type(int t) {
  return t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_ADD || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MAX || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MIN;
}

bool meltw_supported(const libxsmm_meltw_descriptor& d) {
  const int t = d.param;
  const bool f64 = d.in0_type == LIBXSMM_DATATYPE_F64 && d.out_type == LIBXSMM_DATATYPE_F64;
  if (d.operation == LIBXSMM_MELTW_OPERATION_UNARY) {
    int v; const int sz = payload_size(d.in0_type);
    if (t == LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_NORMT || xform_mode(t, &v) != 0 ||
        t == LIBXSMM_MELTW_TYPE_UNARY_GATHER || t == LIBXSMM_MELTW_TYPE_UNARY_SCATTER) return sz == 1 || sz == 2 || sz == 4 || sz == 8;
    if (t == LIBXSMM_MELTW_TYPE_UNARY_DROPOUT || t == LIBXSMM_MELTW_TYPE_UNARY_DROPOUT_INV)      // no broadcasts; 16 rows per draw
      return is_tpp_float(d.in0_type) && is_tpp_float(d.out_type) && !(d.flags & (LIBXSMM_MELTW_FLAG_UNARY_BCAST_ROW | LIBXSMM_MELTW_FLAG_UNARY_BCAST_COL | LIBXSMM_MELTW_FLAG_UNARY_BCAST_SCALAR | LIBXSMM_MELTW_FLAG_UNARY_STOCHASTIC_ROUND));
    if (is_reduce_cols_idx_type(t)) return is_tpp_float(d.in0_type) && is_tpp_float(d.out_type);
    if (t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_TO_SCALAR_OP_ADD)                                  // [ref: :2097-2116] f64 only when input, output and compute all are
      return (f64 && d.comp_type == LIBXSMM_DATATYPE_F64) || (is_tpp_float(d.in0_type) && is_tpp_float(d.out_type));
    if (t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD_NCNC_FORMAT)                              // [ref: :2118-2141] m = bc, n = bn, ldi = C, ldo = N
      return is_tpp_float(d.in0_type) && is_tpp_float(d.out_type) && d.m > 0 && d.n > 0 && d.ldi >= d.m && d.ldo >= d.n;
    if (t == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X2 || t == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X3)      // [ref: :2437-2470]
      return d.in0_type == LIBXSMM_DATATYPE_F32 && d.out_type == LIBXSMM_DATATYPE_BF16;
    if (is_reduce_type(t) && (d.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_RECORD_ARGOP))       // recorded for MAX / ABSMAX / MIN over columns [ref: :1376-1424]
      return is_tpp_float(d.in0_type) && is_tpp_float(d.out_type) && !(d.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) &&
             (t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX);
    if (is_reduce_type(t)) return is_tpp_float(d.in0_type) && is_tpp_float(d.out_type);
    if (t == LIBXSMM_MELTW_TYPE_UNARY_UNZIP) return d.in0_type == LIBXSMM_DATATYPE_F32 && (d.out_type == LIBXSMM_DATATYPE_BF16 || d.out_type == LIBXSMM_DATATYPE_U16 || d.out_type == LIBXSMM_DATATYPE_I16);
    const auto is_qint = [](int x) { return x == LIBXSMM_DATATYPE_I8 || x == LIBXSMM_DATATYPE_I16 || x == LIBXSMM_DATATYPE_I32; };
    if (t == LIBXSMM_MELTW_TYPE_UNARY_QUANT && d.out_type == LIBXSMM_DATATYPE_NVFP4X2)      // 16-row blocks, E4M3 scales
      return (d.in0_type == LIBXSMM_DATATYPE_BF16 || d.in0_type == LIBXSMM_DATATYPE_F32) && d.m % 16 == 0 && d.ldo % 16 == 0 && d.ldo >= d.m && d.ldi >= d.m;
    if (t == LIBXSMM_MELTW_TYPE_UNARY_QUANT && (d.out_type == LIBXSMM_DATATYPE_MXFP4X2 || d.out_type == LIBXSMM_DATATYPE_MXBF8))     // block-scaled outputs
      return (d.in0_type == LIBXSMM_DATATYPE_BF16 || d.in0_type == LIBXSMM_DATATYPE_F32) && d.m % 32 == 0 && d.ldo % 32 == 0 && d.ldo >= d.m && d.ldi >= d.m;
    if (t == LIBXSMM_MELTW_TYPE_UNARY_QUANT) return d.in0_type == LIBXSMM_DATATYPE_F32 && is_qint(d.out_type);       // [ref: :2195-2240]
    if (t == LIBXSMM_MELTW_TYPE_UNARY_DEQUANT) return is_qint(d.in0_type) && d.out_type == LIBXSMM_DATATYPE_F32;     // [ref: :2330-2360]
    if (f64) {
      switch (t) {
        case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: case LIBXSMM_MELTW_TYPE_UNARY_XOR: case LIBXSMM_MELTW_TYPE_UNARY_X2: case LIBXSMM_MELTW_TYPE_UNARY_SQRT:
        case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: case LIBXSMM_MELTW_TYPE_UNARY_INC: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: return true;
        default: return false;
      }
    }
    if (!is_tpp_float(d.in0_type) || !is_tpp_float(d.out_type)) return false;
    if (d.flags & LIBXSMM_MELTW_FLAG_UNARY_STOCHASTIC_ROUND) {      // BF8 output, the ops of the reference's generic loop [ref: :311-316, :2469-2497]
      if (d.out_type != LIBXSMM_DATATYPE_BF8 || (d.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT)) return false;
      switch (t) {
        case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: case LIBXSMM_MELTW_TYPE_UNARY_X2: case LIBXSMM_MELTW_TYPE_UNARY_SQRT: case LIBXSMM_MELTW_TYPE_UNARY_TANH:
        case LIBXSMM_MELTW_TYPE_UNARY_TANH_INV: case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID: case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID_INV: case LIBXSMM_MELTW_TYPE_UNARY_GELU:
        case LIBXSMM_MELTW_TYPE_UNARY_GELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: case LIBXSMM_MELTW_TYPE_UNARY_INC: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL:
        case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: case LIBXSMM_MELTW_TYPE_UNARY_EXP: return true;
        default: return false;
      }
    }
    switch (t) {
      case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: case LIBXSMM_MELTW_TYPE_UNARY_XOR: case LIBXSMM_MELTW_TYPE_UNARY_X2: case LIBXSMM_MELTW_TYPE_UNARY_SQRT:
      case LIBXSMM_MELTW_TYPE_UNARY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_RELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_TANH: case LIBXSMM_MELTW_TYPE_UNARY_TANH_INV:
      case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID: case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID_INV: case LIBXSMM_MELTW_TYPE_UNARY_GELU: case LIBXSMM_MELTW_TYPE_UNARY_GELU_INV:
      case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: case LIBXSMM_MELTW_TYPE_UNARY_INC: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT:
      case LIBXSMM_MELTW_TYPE_UNARY_EXP: case LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV:
      case LIBXSMM_MELTW_TYPE_UNARY_ELU: case LIBXSMM_MELTW_TYPE_UNARY_ELU_INV: case LIBXSMM_MELTW_TYPE_UNARY_DUMP: return true;
      default: return false;
    }
  }
  if (d.operation == LIBXSMM_MELTW_OPERATION_BINARY) {
    if (t == LIBXSMM_MELTW_TYPE_BINARY_ZIP) return d.in0_type == LIBXSMM_DATATYPE_BF16 || d.in0_type == LIBXSMM_DATATYPE_U16 || d.in0_type == LIBXSMM_DATATYPE_I16;
    const bool arith = t == LIBXSMM_MELTW_TYPE_BINARY_ADD || t == LIBXSMM_MELTW_TYPE_BINARY_MUL || t == LIBXSMM_MELTW_TYPE_BINARY_SUB || t == LIBXSMM_MELTW_TYPE_BINARY_DIV ||
                       t == LIBXSMM_MELTW_TYPE_BINARY_MULADD || t == LIBXSMM_MELTW_TYPE_BINARY_MAX || t == LIBXSMM_MELTW_TYPE_BINARY_MIN;
    const bool cmp = t >= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GT && t <= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_NE;
    if (t == LIBXSMM_MELTW_TYPE_BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD)
      return !f64 && !(d.flags & LIBXSMM_MELTW_FLAG_BINARY_STOCHASTIC_ROUND) && is_tpp_float(d.in0_type) && is_tpp_float(d.in1_type) && is_tpp_float(d.out_type);
    if (d.flags & LIBXSMM_MELTW_FLAG_BINARY_STOCHASTIC_ROUND)       // (MULADD reads its BF8 output as an input: not through the two-pass scheme)
      return arith && t != LIBXSMM_MELTW_TYPE_BINARY_MULADD && !f64 && d.out_type == LIBXSMM_DATATYPE_BF8 && is_tpp_float(d.in0_type) && is_tpp_float(d.in1_type);
    if (f64) return arith && d.in1_type == LIBXSMM_DATATYPE_F64;
    if (!is_tpp_float(d.in0_type) || !is_tpp_float(d.in1_type)) return false;
    if (cmp) return true;
    return arith && is_tpp_float(d.out_type);
  }
  if (d.operation == LIBXSMM_MELTW_OPERATION_TERNARY) {
    if ((d.flags & LIBXSMM_MELTW_FLAG_TERNARY_STOCHASTIC_ROUND) && (f64 || d.out_type != LIBXSMM_DATATYPE_BF8)) return false;
    if (t == LIBXSMM_MELTW_TYPE_TERNARY_SELECT) return f64 ? d.in1_type == LIBXSMM_DATATYPE_F64 : (is_tpp_float(d.in0_type) && is_tpp_float(d.in1_type) && is_tpp_float(d.out_type));
    if (t == LIBXSMM_MELTW_TYPE_TERNARY_MULADD || t == LIBXSMM_MELTW_TYPE_TERNARY_NMULADD)
      return is_tpp_float(d.in0_type) && is_tpp_float(d.in1_type) && is_tpp_float(d.in2_type) && is_tpp_float(d.out_type);
    return false;
  }
  return false;
}

template <template <int> class K> struct SizeSwitch;

#define LAUNCH_BY_SIZE(KERNEL, SZ, GRID, BLOCK, ST, ...)                                          \
  do { switch (SZ) {                                                                              \
      case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, BLOCK, 0, ST, __VA_ARGS__); break;             \
      case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, BLOCK, 0, ST, __VA_ARGS__); break;             \
      case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, BLOCK, 0, ST, __VA_ARGS__); break;             \
      default: hipLaunchKernelGGL((KERNEL<8>), GRID, BLOCK, 0, ST, __VA_ARGS__); break; } } while (0)

// row gather / scatter through an LDS-staged column: the staged extent (the side that is indexed) is the leading dimension of that side, must be
// 16-byte granular and aligned, fit 64 KiB, and be at least half used (m >= rows / 2) -- otherwise the direct kernel moves fewer bytes.
// The index values must lie inside the staged extent: true by the TPP's contract (an index beyond the leading dimension would alias the next column).
static bool gs_rows_lds_ok(const MeltwArgs& a, int sz) {
  const bool gather = a.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER;
  const long long rows = gather ? a.ldi : a.ldo;
  if (rows <= 0 || (rows * sz) % 16 != 0 || rows * sz > 65536 || 2ll * a.m < rows || a.n <= 0 || a.n > 65535 * 16 || a.nbatch >= 65536) return false;
  const size_t base = gather ? ((size_t)a.in0 | (size_t)a.bs_in0) : ((size_t)a.out | (size_t)a.bs_out);
  return (base & 15) == 0;
}

typedef float f32x4m __attribute__((ext_vector_type(4)));
// MX-typed C of an MX x MX GEMM [ref: gemm ref :661-817]: the f32 result [batch][n][ldc] is quantised in blocks of 32 consecutive rows --
// the reference's bf16-flavoured pipeline (inputs, scale, reciprocal and every scaled value pass through bf16), NOT the QUANT TPP's exact one.
// One thread per block: 128 bytes in, 16 (E2M1 pairs) or 32 (E5M2) bytes + one scale byte out.
__global__ __launch_bounds__(256) void mx_out_quant_kernel(const float* src_, unsigned char* dst_, unsigned char* scf_, int m, int n, int ldc, int fp4,
                                                           unsigned int nbatch, long long bs_dst, long long bs_scf, unsigned int total) {
  const unsigned int t = blockIdx.x * 256u + threadIdx.x;
  if (t >= total) return;
  const unsigned int mblk = (unsigned int)m / 32u, per = mblk * (unsigned int)n;
  const unsigned int b = t / per, r = t - b * per, j = r / mblk, i = (r - j * mblk) * 32u;
  GM const float* in = (GM const float*)src_ + ((long long)b * n + j) * ldc + i;
  float x[32], amax = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) { const f32x4m v = *(GM const f32x4m*)(in + 4 * q); x[4 * q] = nv_bf16(v[0]); x[4 * q + 1] = nv_bf16(v[1]); x[4 * q + 2] = nv_bf16(v[2]); x[4 * q + 3] = nv_bf16(v[3]); }
#pragma unroll
  for (int e = 0; e < 32; ++e) { const float a = fabsf(x[e]); if (a > amax || a != a) amax = a; }
  int se = (amax == 0.0f) ? 0 : (int)((__float_as_uint(amax) >> 23) & 0xffu);
  se -= fp4 ? 2 : 15;
  se = se < 0 ? 0 : (se > 254 ? 254 : se);
  ((GM unsigned char*)scf_)[(long long)b * bs_scf + (long long)j * (ldc / 32) + i / 32u] = (unsigned char)se;
  const float scale = nv_bf16(__uint_as_float(((unsigned int)se << 23) | (se == 0 ? (1u << 22) : 0u)));
  const float rcp = nv_bf16(1.0f / scale);
  if (fp4) {
    GM unsigned char* out = (GM unsigned char*)dst_ + (long long)b * bs_dst + (long long)j * (ldc / 2) + i / 2u;
    unsigned int w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const float a = fabsf(nv_bf16(x[e] * rcp));
      const unsigned int code = (a != a) ? 7u : (a > 5.0f) ? 7u : (a >= 3.5f) ? 6u : (a > 2.5f) ? 5u : (a >= 1.75f) ? 4u : (a > 1.25f) ? 3u : (a >= 0.75f) ? 2u : (a > 0.25f) ? 1u : 0u;
      w[e / 8] |= (((__float_as_uint(x[e]) >> 31) << 3) | code) << (4 * (e % 8));
    }
    u32x4m o; o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; o[3] = w[3];
    *(GM u32x4m*)out = o;
  } else {
    GM unsigned char* out = (GM unsigned char*)dst_ + (long long)b * bs_dst + (long long)j * ldc + i;
    unsigned int w[8];
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      unsigned int c = (unsigned int)lowp::f16_to_bf8_rne(lowp::f32_to_f16(nv_bf16(x[e] * rcp)));
      if ((c & 0x7cu) == 0x7cu) c = (c & 0x80u) | 0x7bu;
      if (e % 4 == 0) w[e / 4] = c; else w[e / 4] |= c << (8 * (e % 4));
    }
    u32x4m o0, o1; o0[0] = w[0]; o0[1] = w[1]; o0[2] = w[2]; o0[3] = w[3]; o1[0] = w[4]; o1[1] = w[5]; o1[2] = w[6]; o1[3] = w[7];
    *(GM u32x4m*)out = o0; *(GM u32x4m*)(out + 16) = o1;
  }
}
int launch_mx_out_quant(const float* src, void* dst, void* scf, int m, int n, int ldc, int fp4, unsigned int nbatch, long long bs_dst, long long bs_scf, void* stream) {
  const unsigned long long total = (unsigned long long)(m / 32) * (unsigned long long)n * nbatch;
  if (total == 0 || total >= (1ull << 32)) return total == 0 ? 0 : (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mx_out_quant_kernel, dim3((unsigned int)((total + 255ull) / 256ull)), dim3(256), 0, (hipStream_t)stream, src, (unsigned char*)dst, (unsigned char*)scf, m, n, ldc, fp4,
                     nbatch, bs_dst, bs_scf, (unsigned int)total);
  return (int)hipGetLastError();
}

static const u32x4m* dropout_jump_tables();
// second pass of a TPP with stochastic rounding: `a.in0` = the f32 results [call][n][m], `a.out` = the BF8 destination, a.aux_in = the state
int launch_stochastic_bf8(const MeltwArgs& a, void* stream) {
  const u32x4m* jt = dropout_jump_tables();
  if (!jt) return (int)hipErrorOutOfMemory;
  const unsigned long long mn = (unsigned long long)a.m * a.n, per_stream = ((mn + 15ull) / 16ull) * a.nbatch;
  unsigned long long segs = std::min<unsigned long long>((per_stream + 7ull) / 8ull, 16384ull);
  if (segs == 0) segs = 1;
  const unsigned long long L = (per_stream + segs - 1ull) / segs;
  segs = (per_stream + L - 1ull) / L;
  if (!a.ws || a.ws_bytes < 256) return (int)hipErrorInvalidValue;
  if (hipMemcpyAsync(a.ws, a.aux_in, 256, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
  hipLaunchKernelGGL(stochastic_bf8_kernel, dim3((unsigned int)((segs * 16ull + 255ull) / 256ull)), dim3(256), 0, (hipStream_t)stream, a, jt, L);
  return (int)hipGetLastError();
}

int launch_meltw(const MeltwArgs& a, void* stream, const char** name) {
  hipStream_t st = (hipStream_t)stream;
  const bool listed = a.operation == LIBXSMM_MELTW_OPERATION_UNARY && is_reduce_cols_idx_type(a.type);      // the shape's n does not matter there (the reference's driver passes 0)
  // the extent along n may come with the call instead of the descriptor: the listed reductions (their column list), REPLICATE_COL_VAR (op.primary) --
  // the reference's drivers dispatch those with n = 0
  const bool n_by_call = listed || (a.operation == LIBXSMM_MELTW_OPERATION_UNARY && a.type == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR && a.scalar_u64 > 0);
  if (a.nbatch == 0 || a.m <= 0 || (a.n <= 0 && !n_by_call)) { if (name) *name = "(empty)"; return 0; }
  const int sz = payload_size(a.in0_type);
  if (listed) {
    hipLaunchKernelGGL(reduce_cols_listed_kernel, dim3((unsigned int)((a.m + 255) / 256), a.nbatch), dim3(256), 0, st, a);
    if (name) *name = "reduce_cols_listed_kernel";
    return (int)hipGetLastError();
  }
  if (ew8_ok(a)) {
    const int g = ew_elems(a);
    const unsigned int m8 = (unsigned int)(a.m / g), total = m8 * (unsigned int)a.n * (unsigned int)a.nbatch;
    const dim3 grid((total + 255u) / 256u);
#define EW_(N_, T_, E_) hipLaunchKernelGGL((meltw_ew8_kernel<N_, T_, E_>), grid, dim3(256), 0, st, a, m8, total)
#define EWN_(T_, E_) do { if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY) EW_(1, T_, E_); else if (a.operation == LIBXSMM_MELTW_OPERATION_BINARY) EW_(2, T_, E_); else EW_(3, T_, E_); } while (0)
    // (non-temporal: f32 copy 4096 x 8192 0.715 -> 0.75; the bf16 tile streams of config #5 lose 1 % with it and stay cacheable -- profiles/r06_tpp_nt.jsonl)
    if (g == 4) { if (a.nt) EWN_(true, 4); else EWN_(false, 4); }
    else EWN_(false, 8);
#undef EWN_
#undef EW_
    if (name) *name = "meltw_ew8_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_BINARY && a.type == LIBXSMM_MELTW_TYPE_BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD) {
    hipLaunchKernelGGL(mul_reduce_scalar_kernel, dim3(a.nbatch), dim3(1024), 0, st, a);
    if (name) *name = "mul_reduce_scalar_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY && a.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_TO_SCALAR_OP_ADD) {
    if (a.in0_type == LIBXSMM_DATATYPE_F64) hipLaunchKernelGGL((reduce_scalar_kernel<double>), dim3(a.nbatch), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((reduce_scalar_kernel<float>), dim3(a.nbatch), dim3(1024), 0, st, a);
    if (name) *name = "reduce_scalar_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY && a.type == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD_NCNC_FORMAT) {
    hipLaunchKernelGGL(reduce_ncnc_kernel, dim3((unsigned int)((a.ldi + 255) / 256), a.nbatch), dim3(256), 0, st, a);
    if (name) *name = "reduce_ncnc_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY && a.type == LIBXSMM_MELTW_TYPE_UNARY_DROPOUT) {
    const u32x4m* jt = dropout_jump_tables();
    if (!jt) return (int)hipErrorOutOfMemory;
    const unsigned long long total = (unsigned long long)a.n * (unsigned long long)((a.m + 15) / 16) * a.nbatch;
    unsigned long long segs = std::min<unsigned long long>((total + 7ull) / 8ull, 16384ull);      // at least 8 draws per thread, at most 256 Ki threads
    if (segs == 0) segs = 1;
    const unsigned long long L = (total + segs - 1ull) / segs;
    segs = (total + L - 1ull) / L;
    if (!a.ws || a.ws_bytes < 256) return (int)hipErrorInvalidValue;
    if (hipMemcpyAsync(a.ws, a.aux_in, 256, hipMemcpyDeviceToDevice, st) != hipSuccess) return (int)hipGetLastError();      // the kernel reads the snapshot, writes aux_in
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned int)((segs * 16ull + 255ull) / 256ull)), dim3(256), 0, st, a, jt, L);
    if (name) *name = "dropout_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY && a.type == LIBXSMM_MELTW_TYPE_UNARY_DROPOUT_INV) {
    const unsigned long long total = (unsigned long long)a.m * a.n * a.nbatch;
    hipLaunchKernelGGL(dropout_inv_kernel, dim3((unsigned int)((total + 255ull) / 256ull)), dim3(256), 0, st, a);
    if (name) *name = "dropout_inv_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY && a.type == LIBXSMM_MELTW_TYPE_UNARY_QUANT && a.out_type == LIBXSMM_DATATYPE_NVFP4X2) {
    const unsigned int mblk = (unsigned int)(a.m / 16), total = mblk * (unsigned int)a.n * (unsigned int)a.nbatch;
    hipLaunchKernelGGL(nvfp4_quant_kernel, dim3((total + 255u) / 256u), dim3(256), 0, st, a, mblk, total);
    if (name) *name = "nvfp4_quant_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY && a.type == LIBXSMM_MELTW_TYPE_UNARY_QUANT && (a.out_type == LIBXSMM_DATATYPE_MXFP4X2 || a.out_type == LIBXSMM_DATATYPE_MXBF8)) {
    const unsigned int mblk = (unsigned int)(a.m / 32), total = mblk * (unsigned int)a.n * (unsigned int)a.nbatch;
    if (a.out_type == LIBXSMM_DATATYPE_MXFP4X2) hipLaunchKernelGGL((mx_quant_kernel<true>), dim3((total + 255u) / 256u), dim3(256), 0, st, a, mblk, total);
    else hipLaunchKernelGGL((mx_quant_kernel<false>), dim3((total + 255u) / 256u), dim3(256), 0, st, a, mblk, total);
    if (name) *name = "mx_quant_kernel";
    return (int)hipGetLastError();
  }
  if (a.operation == LIBXSMM_MELTW_OPERATION_UNARY) {
    int v = 0; const int mode = xform_mode(a.type, &v);
    constexpr bool xvec_off = false;
    const bool base16 = ((((size_t)a.in0 | (size_t)a.out | (size_t)a.bs_in0 | (size_t)a.bs_out) & 15) == 0);
    const int vec = 16 / (sz > 0 ? sz : 1);
    if (a.type == LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_NORMT && !xvec_off && base16 && a.m % vec == 0 && a.n % vec == 0 && a.ldi % vec == 0 && a.ldo % vec == 0 &&
        (long long)((a.m + 63) / 64) * ((a.n + 63) / 64) * a.nbatch < (1ll << 31)) {
      const unsigned int tm = (unsigned int)((a.m + 63) / 64), tn = (unsigned int)((a.n + 63) / 64);
      LAUNCH_BY_SIZE(transpose_vec_kernel, sz, dim3(tm * tn * (unsigned int)a.nbatch), dim3(256), st, a, tm, tn);
      if (name) *name = "transpose_vec_kernel";
    } else if (mode == XF_NORM_TO_VNNI && v == 2 && sz == 2 && !xvec_off && base16 && a.m % 4 == 0 && a.ldi % 4 == 0 && a.ldo % 4 == 0 &&
               (long long)(a.ldo / 4) * ((a.n + 1) / 2) * a.nbatch < (1ll << 32) - 256) {
      // four positions per thread (8-byte loads, one 16-byte store); the eight-position form stays for what is not 8-byte granular on the input side -- nothing: ldi % 4 == 0
      // with a 16-byte aligned base makes every row start 8-byte aligned
      const unsigned int o4 = (unsigned int)(a.ldo / 4), total = o4 * (unsigned int)((a.n + 1) / 2) * (unsigned int)a.nbatch;
      hipLaunchKernelGGL((vnni2_vec_kernel<4>), dim3((total + 255u) / 256u), dim3(256), 0, st, a, o4, total);
      if (name) *name = "vnni2_vec_kernel";
    } else if (mode == XF_NORM_TO_VNNI && v == 2 && sz == 2 && !xvec_off && (((size_t)a.out | (size_t)a.bs_out) % 4) == 0 && (long long)a.ldo * ((a.n + 1) / 2) < (1ll << 31) && a.nbatch < 65536) {
      constexpr bool quad_off = false;
      if (!quad_off && a.ldo >= 16 && (((size_t)a.in0 | (size_t)a.bs_in0) % 2) == 0) {          // four positions per thread (rows of at least a few threads)
        const unsigned int q4 = ((unsigned int)a.ldo + 3u) / 4u, per_q = q4 * (unsigned int)((a.n + 1) / 2);
        hipLaunchKernelGGL(vnni2_quad_kernel, dim3((per_q + 255u) / 256u, a.nbatch), dim3(256), 0, st, a, q4, per_q);
        if (name) *name = "vnni2_quad_kernel";
      } else {
      const unsigned int per_batch = (unsigned int)a.ldo * (unsigned int)((a.n + 1) / 2);
      hipLaunchKernelGGL(vnni2_pair_kernel, dim3((per_batch + 255u) / 256u, a.nbatch), dim3(256), 0, st, a, per_batch);
      if (name) *name = "vnni2_pair_kernel";
      }
    } else if (mode == XF_NORM_TO_VNNI && v == 4 && sz == 1 && !xvec_off && base16 && a.m % 4 == 0 && a.ldi % 4 == 0 && a.ldo % 4 == 0 &&
               (long long)(a.ldo / 4) * ((a.n + 3) / 4) * a.nbatch < (1ll << 32) - 256) {
      const unsigned int o4 = (unsigned int)(a.ldo / 4), total = o4 * (unsigned int)((a.n + 3) / 4) * (unsigned int)a.nbatch;
      hipLaunchKernelGGL(vnni4_vec_kernel, dim3((total + 255u) / 256u), dim3(256), 0, st, a, o4, total);
      if (name) *name = "vnni4_vec_kernel";
    } else if (a.type == LIBXSMM_MELTW_TYPE_UNARY_TRANSFORM_NORM_TO_NORMT) {
      const long long tiles = (long long)((a.m + 31) / 32) * ((a.n + 31) / 32) * a.nbatch;
      LAUNCH_BY_SIZE(transpose_kernel, sz, dim3((unsigned int)tiles), dim3(256), st, a);
      if (name) *name = "transpose_kernel";
    } else if (mode != 0) {
      long long total; int pad_n = a.n;
      if (mode == XF_NORM_TO_VNNI) total = (long long)a.ldo * (((a.n + v - 1) / v) * v);
      else if (mode == XF_PAD) { pad_n = ((a.n + v - 1) / v) * v; total = (long long)a.ldo * pad_n; }
      else total = (long long)a.m * a.n;
      LAUNCH_BY_SIZE(xform_kernel, sz, dim3((unsigned int)((total + 255) / 256), a.nbatch), dim3(256), st, a, mode, v, 0, pad_n);
      if (name) *name = "xform_kernel";
    } else if (a.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER || a.type == LIBXSMM_MELTW_TYPE_UNARY_SCATTER) {
      const long long total = (long long)a.m * a.n;
      const long long colbytes = (long long)a.m * sz;
      if ((a.flags & LIBXSMM_MELTW_FLAG_UNARY_GS_COLS) && !xvec_off && colbytes % 16 == 0 && ((long long)a.ldi * sz) % 16 == 0 && ((long long)a.ldo * sz) % 16 == 0 &&
          ((((size_t)a.in0 | (size_t)a.out | (size_t)a.bs_in0 | (size_t)a.bs_out) & 15) == 0) && (colbytes / 16) * a.n < (1ll << 32) - 256 && a.nbatch < 65536) {
        const unsigned int vpc = (unsigned int)(colbytes / 16), tot = vpc * (unsigned int)a.n;
        hipLaunchKernelGGL(gather_cols_vec_kernel, dim3((tot + 255u) / 256u, a.nbatch), dim3(256), 0, st, a, sz, vpc, tot);
        if (name) *name = "gather_cols_vec_kernel";
      } else if ((a.flags & LIBXSMM_MELTW_FLAG_UNARY_GS_ROWS) && !xvec_off && gs_rows_lds_ok(a, sz)) {
        const int rows = a.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER ? a.ldi : a.ldo;
        const size_t lds_bytes = (size_t)rows * sz;
        constexpr int nc_env = 2;       // columns per workgroup of the gather (1: the older form).  4096 x 8192 / 2048 x 16384 f32: 1 -> 0.550 / 0.601, 2 -> 0.563 / 0.628, 4 (64 KiB of LDS, two workgroups per CU) -> 0.448 / 0.578
        if (a.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER && nc_env > 1 && (sz == 4 || sz == 2) && a.n >= 64 && ((long long)a.ldi * sz) % 16 == 0) {
          const int nc = (nc_env >= 4 && lds_bytes * 4 <= 65536) ? 4 : ((lds_bytes * 2 <= 65536) ? 2 : 1);
          if (nc > 1) {
            const dim3 g((unsigned int)((a.n + nc - 1) / nc), a.nbatch);
            if (sz == 4) { if (nc == 4) hipLaunchKernelGGL((gs_rows_lds_multi_kernel<4, 4>), g, dim3(256), lds_bytes * 4, st, a, rows); else hipLaunchKernelGGL((gs_rows_lds_multi_kernel<4, 2>), g, dim3(256), lds_bytes * 2, st, a, rows); }
            else { if (nc == 4) hipLaunchKernelGGL((gs_rows_lds_multi_kernel<2, 4>), g, dim3(256), lds_bytes * 4, st, a, rows); else hipLaunchKernelGGL((gs_rows_lds_multi_kernel<2, 2>), g, dim3(256), lds_bytes * 2, st, a, rows); }
            if (name) *name = "gs_rows_lds_multi_kernel";
            return (int)hipGetLastError();
          }
        }
        switch (sz) {
          case 1: hipLaunchKernelGGL((gs_rows_lds_kernel<1>), dim3((unsigned int)a.n, a.nbatch), dim3(256), lds_bytes, st, a, rows); break;
          case 2: hipLaunchKernelGGL((gs_rows_lds_kernel<2>), dim3((unsigned int)a.n, a.nbatch), dim3(256), lds_bytes, st, a, rows); break;
          case 4: hipLaunchKernelGGL((gs_rows_lds_kernel<4>), dim3((unsigned int)a.n, a.nbatch), dim3(256), lds_bytes, st, a, rows); break;
          default: hipLaunchKernelGGL((gs_rows_lds_kernel<8>), dim3((unsigned int)a.n, a.nbatch), dim3(256), lds_bytes, st, a, rows); break;
        }
        if (name) *name = "gs_rows_lds_kernel";
      } else if (!(a.flags & (LIBXSMM_MELTW_FLAG_UNARY_GS_COLS | LIBXSMM_MELTW_FLAG_UNARY_GS_ROWS | LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_8BYTES)) && !xvec_off && (sz == 4 || sz == 2) &&
                 a.m % 4 == 0 && (long long)a.m * a.n / 4 < (1ll << 31) &&
                 ((size_t)(a.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER ? a.aux_in : (const void*)a.aux_out) % 16) == 0 &&
                 (a.type == LIBXSMM_MELTW_TYPE_UNARY_GATHER ? (((size_t)a.out | (size_t)a.bs_out) % (4 * sz) == 0 && ((long long)a.ldo * sz) % (4 * sz) == 0)
                                                             : (((size_t)a.in0 | (size_t)a.bs_in0) % (4 * sz) == 0 && ((long long)a.ldi * sz) % (4 * sz) == 0))) {
        const unsigned int m4 = (unsigned int)(a.m / 4), tot4 = m4 * (unsigned int)a.n;
        if (sz == 4) hipLaunchKernelGGL((gs_offs_vec4_kernel<4>), dim3((tot4 + 255u) / 256u, a.nbatch), dim3(256), 0, st, a, m4, tot4);
        else hipLaunchKernelGGL((gs_offs_vec4_kernel<2>), dim3((tot4 + 255u) / 256u, a.nbatch), dim3(256), 0, st, a, m4, tot4);
        if (name) *name = "gs_offs_vec4_kernel";
      } else {
        LAUNCH_BY_SIZE(gather_scatter_kernel, sz, dim3((unsigned int)((total + 255) / 256), a.nbatch), dim3(256), st, a);
        if (name) *name = "gather_scatter_kernel";
      }
    } else if (is_reduce_cols_idx_type(a.type) || (is_reduce_type(a.type) && (a.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_RECORD_ARGOP))) {
      hipLaunchKernelGGL(reduce_cols_listed_kernel, dim3((unsigned int)((a.m + 255) / 256), a.nbatch), dim3(256), 0, st, a);
      if (name) *name = "reduce_cols_listed_kernel";
    } else if (is_reduce_type(a.type)) {
      const bool rows = (a.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) != 0;
      const bool bf = a.in0_type == LIBXSMM_DATATYPE_BF16;
      constexpr bool rvec_off = false;
      const size_t al = bf ? 8 : 16;
      if (!rvec_off && is_float_type(a.in0_type) && a.m % 4 == 0 && a.ldi % 4 == 0 && (((size_t)a.in0 | (size_t)a.bs_in0) % al) == 0 && a.nbatch < 65536) {
        int G = 1; while (G < 64 && G < a.m / 4) G <<= 1;
        const int slices = (!rows && a.n >= 256) ? 16 : 1;
        const unsigned int gx = rows ? (unsigned int)((a.n + 4 * (64 / G) - 1) / (4 * (64 / G))) : (unsigned int)((a.m / 4 + 15) / 16);
        // one big matrix over its columns: too few row groups to fill the chip -> split the columns over blockIdx.z (two passes)
        int nchunks = 1;
        // measured on one 4096 x 8192 f32 matrix (round 3): 2048 blocks 38.9 us, 1024 34.7, 512 32.1, 384 31.6, 256 36.9, 128 60.4 -- few, long-running blocks win;
        // 64 or 32 row groups per block (whole 1 KiB / 512-byte row segments per wave, 4 / 8 column slices) were SLOWER at every block count (42 - 62 us);
        // round 4: 256 row groups per block (one contiguous 4 KiB run of every column, eight columns in flight per thread, 32 - 512 column chunks): 54 - 66 us
        // (8192 x 8192: 52 -> 66 - 91 us; 1024 x 65536: 60 -> 138 - 392 us) -- the short pieces of many columns at once are what this memory system wants here
        if (!rows && a.ws && a.nbatch == 1 && gx < 512) {
          // round 6, with the second pass no longer waiting for one chunk at a time: 512 blocks 33.1 us, 1024 34.9, 2048 34.2, 4096 35.5; whole 1 KiB row segments per wave
          // (64 row groups x 4 slices) 38 us at 512 and at 2048 blocks (profiles/r06_reduce_cols_scan.jsonl) -- still the 16 x 16 form at 512 blocks
          nchunks = (int)std::min<long long>(128, std::min<long long>(a.n / 64, 512 / (gx ? gx : 1)));
          if ((size_t)nchunks * 2 * (size_t)a.m * sizeof(float) > a.ws_bytes) nchunks = 1;
        }
        if (rows && a.m / 4 <= G && a.n >= 64) {     // short columns: four per lane group and trip
          const unsigned int gx4 = (unsigned int)((a.n + 16 * (64 / G) - 1) / (16 * (64 / G)));
          if (bf) hipLaunchKernelGGL((reduce_vec_kernel<true, 16, 4>), dim3(gx4, a.nbatch), dim3(256), 0, st, a, G, slices, 0, (float*)nullptr);
          else hipLaunchKernelGGL((reduce_vec_kernel<false, 16, 4>), dim3(gx4, a.nbatch), dim3(256), 0, st, a, G, slices, 0, (float*)nullptr);
          if (name) *name = "reduce_vec_kernel";
        } else if (nchunks > 1) {
          const int chunk = (a.n + nchunks - 1) / nchunks;
          nchunks = (a.n + chunk - 1) / chunk;
          if (bf) hipLaunchKernelGGL((reduce_vec_kernel<true>), dim3(gx, 1, nchunks), dim3(256), 0, st, a, G, slices, chunk, (float*)a.ws);
          else hipLaunchKernelGGL((reduce_vec_kernel<false>), dim3(gx, 1, nchunks), dim3(256), 0, st, a, G, slices, chunk, (float*)a.ws);
          hipLaunchKernelGGL(reduce_combine_kernel, dim3((unsigned int)((a.m + 255) / 256)), dim3(256), 0, st, a, (const float*)a.ws, nchunks);
          if (name) *name = "reduce_vec_kernel+combine";
        } else {
          if (bf) hipLaunchKernelGGL((reduce_vec_kernel<true>), dim3(gx, a.nbatch), dim3(256), 0, st, a, G, slices, 0, (float*)nullptr);
          else hipLaunchKernelGGL((reduce_vec_kernel<false>), dim3(gx, a.nbatch), dim3(256), 0, st, a, G, slices, 0, (float*)nullptr);
          if (name) *name = "reduce_vec_kernel";
        }
      } else {
        const unsigned int gx = rows ? (unsigned int)((a.n + 3) / 4) : (unsigned int)((a.m + 255) / 256);
        hipLaunchKernelGGL(reduce_kernel, dim3(gx, a.nbatch), dim3(256), 0, st, a);
        if (name) *name = "reduce_kernel";
      }
    } else {
      const int bc = bcast_kind(a.operation, a.type, a.flags, 0);
      const bool simple = bc == BC_NONE && a.in0_type == a.out_type && is_float_type(a.in0_type) && (a.m % 4 == 0) && (a.ldi % 4 == 0) && (a.ldo % 4 == 0) &&
        !(a.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) && a.type != LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR &&
        a.type != LIBXSMM_MELTW_TYPE_UNARY_RELU_INV && a.type != LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV && a.type != LIBXSMM_MELTW_TYPE_UNARY_ELU_INV &&
        a.type != LIBXSMM_MELTW_TYPE_UNARY_UNZIP && a.type != LIBXSMM_MELTW_TYPE_UNARY_DUMP;
      const int esz = (a.in0_type == LIBXSMM_DATATYPE_F32) ? 4 : 2;
      const bool aligned = (((size_t)a.in0 | (size_t)a.out | (size_t)a.bs_in0 | (size_t)a.bs_out) % (size_t)(4 * esz)) == 0;
      if (simple && aligned) {
        const long long total = (long long)(a.m / 4) * a.n * a.nbatch;
        if (esz == 4) hipLaunchKernelGGL((meltw_unary_vec4_kernel<f32x4, false>), dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((meltw_unary_vec4_kernel<u16x4, true>), dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, st, a);
        if (name) *name = "meltw_unary_vec4_kernel";
      } else {
        const int n_eff = (a.type == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR) ? (int)a.scalar_u64 : a.n;
        const long long blocks = (long long)((a.m + 63) / 64) * ((n_eff + 3) / 4) * a.nbatch;
        if (blocks > 0) hipLaunchKernelGGL(meltw_unary_kernel, dim3((unsigned int)blocks), dim3(64, 4), 0, st, a);
        if (name) *name = "meltw_unary_kernel";
      }
    }
  } else {
    const long long blocks = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * a.nbatch;
    if (a.operation == LIBXSMM_MELTW_OPERATION_BINARY) { hipLaunchKernelGGL(meltw_binary_kernel, dim3((unsigned int)blocks), dim3(64, 4), 0, st, a); if (name) *name = "meltw_binary_kernel"; }
    else { hipLaunchKernelGGL(meltw_ternary_kernel, dim3((unsigned int)blocks), dim3(64, 4), 0, st, a); if (name) *name = "meltw_ternary_kernel"; }
  }
  return (int)hipGetLastError();
}

}  // namespace xamd
