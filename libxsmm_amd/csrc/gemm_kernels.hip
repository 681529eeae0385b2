// gemm_kernels.hip -- dense GEMM / BRGEMM kernels for gfx950 (MI355X, CDNA4).
//
// Semantics restated from the reference [ref: src/generator_gemm_reference_impl.c:1359-1426 (f32),
// :2127-2170 / :2367-2419 (bf16), :294-372 (fused bias / ReLU / sigmoid), :180-197 (BR addressing)]:
//   C[m x n] = act( bcast_col(D) + beta*C + sum_{r<br} A_r * B_r ),  alpha == 1, beta in {0,1}
// all matrices column-major ("m fastest").  One *wavefront* owns one output tile for the whole
// (batch-reduce, K) reduction; the batch axis of the batched launchers is simply more tiles.
//
// Kernel families
//   gemm_mfma_f32   v_mfma_f32_32x32x2_f32.  The product is formed TRANSPOSED (D' = B^T A^T): the
//                   MFMA result then has lanes running along GEMM-i, the contiguous dimension of
//                   C, so C is read/written with fully coalesced 128-byte rows and the A operand
//                   (m contiguous) is loaded straight into its fragment layout with coalesced
//                   dword loads -- no LDS round trip.  The k-contiguous operand (B, or A under
//                   TRANS_A) is fetched with 16-byte loads and put into natural k order with
//                   v_permlane32_swap, so the accumulation is the k-ordered fmaf chain.
//   gemm_mfma_f32_t16  v_mfma_f32_16x16x4_f32 for m,n multiples of 16 (the 16^3 headline shape).
//   gemm_mfma_bf16  v_mfma_f32_32x32x16_bf16, A in VNNI-2 layout, fp32 accumulate, one RNE at store.
//   gemm_generic    one thread per C element, no FMA contraction, every dtype/flag combination the
//                   library accepts (f64, flat/transposed bf16, VNNI_B, VNNI_C ...): the semantic
//                   backstop, bit-identical to the reference's serial loops.
// All four are HBM-bound for the small shapes this library exists for (32^3 f32: 5.3 flop/byte),
// so they are organised as streaming kernels: many independent waves, each with its whole tile
// (8-24 KiB) of loads in flight, no barriers, no LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <utility>
#include "internal.hpp"
#include "lowp.hpp"
#include "gemm_device.hpp"
#include "gemm_tile.hpp"
#include "gemm_8bit.hpp"

// a*b+c below means two roundings unless fma()/MFMA is spelled out: parity with the reference's C
// loops (built without FMA contraction) depends on it.
#pragma clang fp contract(off)

namespace xamd {




// MXFP4 A: base of the E8M0 scales of batch-reduce element r [ref: gemm ref :200-222] -- one byte per (32-deep k-block, row):
// pointer list / byte offset of A * 2 / 32 / byte stride of A * 2 / 32
__device__ __forceinline__ bool is_mx_type(int t) { return t == LIBXSMM_DATATYPE_MXFP4X2 || t == LIBXSMM_DATATYPE_MXBF8 || t == LIBXSMM_DATATYPE_MXHF8 || t == LIBXSMM_DATATYPE_MXBF6 || t == LIBXSMM_DATATYPE_MXHF6; }
__device__ __forceinline__ bool is_fp6_type(int t) { return t == LIBXSMM_DATATYPE_MXBF6 || t == LIBXSMM_DATATYPE_MXHF6; }
// E2M3 / E3M2 (MXHF6 / MXBF6) -> f32, exact [ref: gemm ref :70-92 via E4M3]
__device__ __forceinline__ float fp6_to_f32(unsigned int v, bool e3m2) {
  const unsigned int e = e3m2 ? ((v >> 2) & 7u) : ((v >> 3) & 3u), m = e3m2 ? (v & 3u) : (v & 7u);
  float mag;
  if (e3m2) mag = (e == 0u) ? (float)m * 0.0625f : (1.0f + (float)m * 0.25f) * (float)(1u << e) * 0.125f;
  else mag = (e == 0u) ? (float)m * 0.125f : (1.0f + (float)m * 0.125f) * (float)(1u << e) * 0.5f;
  return ((v >> 5) & 1u) ? -mag : mag;
}
// scales of A (of_b = false) or B: one byte per 32 elements, so a byte distance D of the operand is D * (elements per byte) / 32 here
__device__ __forceinline__ gcptr mx_scale_base(const GemmArgs& p, unsigned int bidx, unsigned long long r, bool of_b) {
  if (p.batch_inner) {                                                                // the scales step with their operand
    if (p.map2d_shift) {
      const unsigned int sh = p.map2d_shift, wg = bidx >> 2, wgs_shift = 2u * sh - 2u, k = wg >> 3;
      const unsigned int S = (wg & 7u) + 8u * (k >> wgs_shift), pl = ((k & ((1u << wgs_shift) - 1u)) << 2) | (bidx & 3u), nsi = p.batch_inner >> sh;
      bidx = of_b ? ((S / nsi) << sh) + (pl >> sh) : ((S % nsi) << sh) + (pl & ((1u << sh) - 1u));
    } else bidx = of_b ? bidx / p.batch_inner : bidx % p.batch_inner;
  }
  gcptr base = of_b ? (gcptr)p.b_scf + (long long)bidx * p.bs_bscf : (gcptr)p.a_scf + (long long)bidx * p.bs_scf;
  const long long epb = ((of_b ? p.b_type : p.a_type) == LIBXSMM_DATATYPE_MXFP4X2) ? 2 : 1;
  if (p.br_mode == 1) return list_entry((const void*)(size_t)base, r);
  if (p.br_mode == 2) return base + ((long long)uniform_u64((unsigned long long)((GM const long long*)(of_b ? p.offs_b : p.offs_a))[r]) * epb) / 32;
  if (p.br_mode == 3 && is_fp6_type(of_b ? p.b_type : p.a_type)) return base + (((of_b ? p.br_stride_b : p.br_stride_a) * 4 / 3) / 32) * (long long)r;     // four elements in three bytes
  if (p.br_mode == 3) return base + (((of_b ? p.br_stride_b : p.br_stride_a) * epb) / 32) * (long long)r;
  return base;
}
__device__ __forceinline__ float e2m1_to_f32(unsigned int c) {     // [ref: gemm ref :60-64]
  const unsigned int m = c & 7u;
  const unsigned int bits = (m == 0u) ? 0u : (m == 1u) ? 0x3f000000u : (((m >> 1) + 126u) << 23) | ((m & 1u) << 22);
  return __uint_as_float(bits | ((c & 8u) << 28));
}

// activation, ReLU bitmask, output conversion / VNNI-C of one element (block = 64 x 4: a wave is 64 rows of one column)
__device__ __forceinline__ void generic_epilogue(const GemmArgs& p, const BatchPtrs& q, int i, int j, bool valid, float acc) {
  const float y = act_apply(p.act, acc);
  if (p.act == 2 && q.mask) {   // bit i%8 of byte i/8 + j*(mask_ld/8) [ref: mateltwise ref :150-157, :2142]
    const unsigned long long ballot = __ballot(valid && !(acc <= 0.0f));
    const unsigned long long vmask = __ballot(valid);
    const int lane = threadIdx.x;   // blockDim.x == 64: lane within the wave
    if ((lane & 7) == 0 && valid) {
      const long long mask_ld = ((p.ldc + 15) / 16) * 16;
      GM unsigned char* byte = q.mask + i / 8 + (long long)j * (mask_ld / 8);
      const unsigned char vm = (unsigned char)((vmask >> lane) & 0xffu);
      const unsigned char nb = (unsigned char)((ballot >> lane) & 0xffu);
      *byte = (unsigned char)((*byte & ~vm) | (nb & vm));
    }
  }
  if (!valid) return;
  if (p.vnni_c && (p.c_type == LIBXSMM_DATATYPE_BF16 || p.c_type == LIBXSMM_DATATYPE_F16)) {
    // NORM -> VNNI2 of the result [ref: gemm ref :2802-2815]; the pad column of an odd n is zero-filled
    GM unsigned short* c = (GM unsigned short*)q.c;
    c[(long long)(j / 2) * p.ldc * 2 + (long long)i * 2 + (j % 2)] = p.c_type == LIBXSMM_DATATYPE_F16 ? __builtin_bit_cast(unsigned short, (_Float16)y) : f32_to_bf16_rne(y);
    if ((p.n & 1) && j == p.n - 1) c[(long long)(j / 2) * p.ldc * 2 + (long long)i * 2 + 1] = 0;
  } else if (p.vnni_c && (p.c_type == LIBXSMM_DATATYPE_BF8 || p.c_type == LIBXSMM_DATATYPE_HF8)) {
    // 8-bit results: NORM -> VNNI4 [ref: gemm ref :2806, mateltwise ref :737-759]; the pad columns up to a multiple of four are zero-filled
    GM unsigned char* c = (GM unsigned char*)q.c;
    c[(long long)(j / 4) * p.ldc * 4 + (long long)i * 4 + (j % 4)] = p.c_type == LIBXSMM_DATATYPE_BF8 ? lowp::f16_to_bf8_rne(lowp::f32_to_f16(y)) : lowp::f16_to_hf8_rne(lowp::f32_to_f16(y));
    if (j == p.n - 1) for (int jj = p.n; (jj & 3) != 0; ++jj) c[(long long)(jj / 4) * p.ldc * 4 + (long long)i * 4 + (jj % 4)] = 0;
  } else if (p.c_type == LIBXSMM_DATATYPE_F32) {
    ((GM float*)q.c)[(long long)j * p.ldc + i] = y;
  } else if (p.c_type == LIBXSMM_DATATYPE_F16) {        // (the reduce pass of a k-sliced GEMM with IEEE-half C: round 3; fused IEEE-half GEMMs: round 6)
    ((GM _Float16*)q.c)[(long long)j * p.ldc + i] = (_Float16)y;
  } else if (p.c_type == LIBXSMM_DATATYPE_BF8 || p.c_type == LIBXSMM_DATATYPE_HF8) {     // C in the operands' 8-bit type [ref: gemm ref :2511-2619, mateltwise ref :310-319]: one two-step RNE at the end
    ((GM unsigned char*)q.c)[(long long)j * p.ldc + i] = p.c_type == LIBXSMM_DATATYPE_BF8 ? lowp::f16_to_bf8_rne(lowp::f32_to_f16(y)) : lowp::f16_to_hf8_rne(lowp::f32_to_f16(y));
  } else {
    ((GM unsigned short*)q.c)[(long long)j * p.ldc + i] = f32_to_bf16_rne(y);
  }
}

// ------------------------------------------------------------------------------------------------
// generic kernel: one thread per C element
// ------------------------------------------------------------------------------------------------
// Under `#pragma clang fp contract(off)` a product followed by a sum is two correctly rounded
// operations (the HIP __fmul_rn/__fadd_rn helpers are plain operators compiled with the header's
// own contraction setting and DO get fused, so they are not used).


// block = (64, 4): x walks i (so a wave is 64 consecutive rows of one column, which makes the
// ReLU bitmask a ballot), y walks j.
#if !defined(XAMD_GEMM_SHARD)      // a plain (non-template) kernel: emitted by the main translation unit only
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs p) {
  const int tiles_i = (p.m + 63) / 64, tiles_j = (p.n + 3) / 4;
  const long long per_gemm = (long long)tiles_i * tiles_j;
  const long long blk = blockIdx.x;
  const unsigned int bidx = (unsigned int)(blk / per_gemm);
  const int t = (int)(blk % per_gemm);
  const int i = (t % tiles_i) * 64 + threadIdx.x;
  const int j = (t / tiles_i) * 4 + threadIdx.y;
  const bool valid = (i < p.m) && (j < p.n);
  const BatchPtrs q = batch_ptrs(p, bidx);
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const bool ta = (p.flags & LIBXSMM_GEMM_FLAG_TRANS_A) != 0, tb = (p.flags & LIBXSMM_GEMM_FLAG_TRANS_B) != 0;
  const bool va = (p.flags & LIBXSMM_GEMM_FLAG_VNNI_A) != 0, vb = (p.flags & LIBXSMM_GEMM_FLAG_VNNI_B) != 0;

  if (p.a_type == LIBXSMM_DATATYPE_F64) {
    if (!valid) return;
    GM double* c = (GM double*)q.c + (long long)j * p.ldc + i;
    double acc = beta0 ? 0.0 : *c;
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      GM const double* a = (GM const double*)ar; GM const double* b = (GM const double*)br;
      for (int s = 0; s < p.k; ++s) {
        const double av = ta ? a[(long long)i * p.lda + s] : a[(long long)s * p.lda + i];
        const double bv = tb ? b[(long long)s * p.ldb + j] : b[(long long)j * p.ldb + s];
        acc = add_rn(acc, mul_rn(av, bv));
      }
    }
    *c = acc;
    return;
  }

  if (p.a_type == LIBXSMM_DATATYPE_F16) {
    // IEEE half GEMM, f32 accumulation [ref: gemm ref :2025-2124]: k ascending (also inside a VNNI-2 pair), beta * C added AFTER the sum,
    // an f32 C rounded to f16 on the way in
    // (round 6) fused column bias / activation [ref: gemm ref :294-372]: the start value bias (+ C), formed in f32, takes C's place in that rule
    const int kb = va ? 2 : 1;
    float acc = 0.0f;
    if (valid) {
      for (unsigned long long r = 0; r < p.br_count; ++r) {
        gcptr ar, br; br_base(p, q, r, ar, br);
        for (int s = 0; s < p.k; ++s) {
          const long long ai = (long long)(s / kb) * ((long long)p.lda * kb) + (long long)i * kb + (s % kb);
          const long long bi = tb ? (long long)s * p.ldb + j : (long long)j * p.ldb + s;
          const float av = (float)__builtin_bit_cast(_Float16, ((GM const unsigned short*)ar)[ai]), bv = (float)__builtin_bit_cast(_Float16, ((GM const unsigned short*)br)[bi]);
          acc = add_rn(acc, mul_rn(av, bv));
          if (p.comp_f16) acc = (float)(_Float16)acc;             // comp_type F16 [ref: gemm ref :2042,:2059-2062]
        }
      }
      if (!beta0 || p.colbias) {
        float start = beta0 ? 0.0f : load_c_f32(q.c, (long long)j * p.ldc + i, p.c_type);
        if (p.colbias) { const float bias = load_c_f32(q.d, i, p.c_type); start = beta0 ? bias : add_rn(bias, start); }
        acc = add_rn(acc, (float)(_Float16)start);
      }
    }
    generic_epilogue(p, q, i, j, valid, acc);
    return;
  }

  if ((p.flags & LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT) && (p.a_type == LIBXSMM_DATATYPE_I4X2 || p.a_type == LIBXSMM_DATATYPE_U4X2 || p.a_type == LIBXSMM_DATATYPE_MXFP4X2)) {
    // interleaved 4-bit weights x 8-bit activations [ref: gemm ref :1009-1088, :1272-1330]; layouts and orders as in oracle_gemm.c contract_i4_intlv
    if (!valid) return;
    const bool mx = p.a_type == LIBXSMM_DATATYPE_MXFP4X2;
    int iacc = 0; float facc = 0.0f;
    if (!mx && !beta0) iacc = ((GM const int*)q.c)[(long long)j * p.ldc + i];
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      const long long brs_a = p.br_mode == 3 ? p.br_stride_a : 0, brs_b = p.br_mode == 3 ? p.br_stride_b : 0;
      if (!mx) {
        const int zpt = ((GM const unsigned char*)p.a_scf)[(long long)bidx * p.bs_scf + ((brs_a * 2) / p.k) * (long long)r + i];
        for (int o = 0; o < p.k / 8; ++o) {
          const unsigned int aw = *(GM const unsigned int*)(ar + ((long long)o * p.lda + i) * 4);
          const unsigned int b0 = *(GM const unsigned int*)(br + (long long)j * p.ldb + 8 * o), b1 = *(GM const unsigned int*)(br + (long long)j * p.ldb + 8 * o + 4);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int lo = (int)(signed char)((int)((aw >> (8 * t)) & 15u) - zpt), hi = (int)(signed char)((int)((aw >> (8 * t + 4)) & 15u) - zpt);
            iacc += lo * (int)((b0 >> (8 * t)) & 255u) + hi * (int)((b1 >> (8 * t)) & 255u);
          }
        }
      } else {
        GM const unsigned char* sa = (GM const unsigned char*)p.a_scf + (long long)bidx * p.bs_scf + ((brs_a * 2) / 32) * (long long)r;
        GM const float* sb = (GM const float*)(p.b_scf + (long long)bidx * p.bs_bscf) + (brs_b / 32) * (long long)r + (long long)j * (p.ldb / 32);
        for (int s = 0; s < p.k / 32; ++s) {
          int tmp = 0;
          for (int o = 4 * s; o < 4 * s + 4; ++o) {
            const unsigned int aw = *(GM const unsigned int*)(ar + ((long long)o * p.lda + i) * 4);
            const unsigned int b0 = *(GM const unsigned int*)(br + (long long)j * p.ldb + 8 * o), b1 = *(GM const unsigned int*)(br + (long long)j * p.ldb + 8 * o + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const unsigned int cl = (aw >> (8 * t)) & 15u, ch = (aw >> (8 * t + 4)) & 15u;
              const int ml = (int)((0x7f55402a20150b00ull >> (8 * (cl & 7u))) & 255ull), mh = (int)((0x7f55402a20150b00ull >> (8 * (ch & 7u))) & 255ull);   // {0,11,21,32,42,64,85,127}
              tmp += ((cl & 8u) ? -ml : ml) * (int)(signed char)(b0 >> (8 * t)) + ((ch & 8u) ? -mh : mh) * (int)(signed char)(b1 >> (8 * t));
            }
          }
          facc = add_rn(facc, mul_rn(mul_rn((float)tmp, __uint_as_float((unsigned int)sa[(long long)s * p.lda + i] << 23)), sb[s]));
        }
      }
    }
    if (!mx) ((GM int*)q.c)[(long long)j * p.ldc + i] = iacc;
    else if (p.c_type == LIBXSMM_DATATYPE_F32) { GM float* c = (GM float*)q.c + (long long)j * p.ldc + i; *c = add_rn(beta0 ? 0.0f : *c, facc); }
    else { GM unsigned short* c = (GM unsigned short*)q.c + (long long)j * p.ldc + i; *c = f32_to_bf16_rne(add_rn(beta0 ? 0.0f : bf16_to_f32(*c), facc)); }
    return;
  }

  if (p.a_type == LIBXSMM_DATATYPE_I1X8 || p.a_type == LIBXSMM_DATATYPE_I2X4) {
    // 1-bit (+-1) and 2-bit (0, +1, -1) weights x 8-bit activations -> i32 [ref: gemm ref :1100-1300]; layouts as in oracle_gemm.c contract_lowbit
    if (!valid) return;
    const bool ub = p.b_type == LIBXSMM_DATATYPE_U8, one_bit = p.a_type == LIBXSMM_DATATYPE_I1X8;
    const int mq = p.m / 4;
    GM int* c = (GM int*)q.c + (long long)j * p.ldc + i;
    int acc = beta0 ? 0 : *c;
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      GM const unsigned char* a = (GM const unsigned char*)ar;
      for (int s = 0; s < p.k / 4; ++s) {
        const unsigned int bw = *(GM const unsigned int*)(br + (long long)j * p.ldb + 4 * s);      // four k of the column (k % 4 == 0, ldb % 4 == 0, base 4-byte aligned: launch checks)
        if (one_bit) {
          const unsigned int nib = ((unsigned int)a[((long long)s * p.lda) / 2 + i / 2] >> (4 * (i & 1))) & 15u;
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) { const int bv = ub ? (int)((bw >> (8 * k2)) & 255u) : (int)(signed char)(bw >> (8 * k2)); acc += ((nib >> k2) & 1u) ? -bv : bv; }
        } else {
          const unsigned int aw = *(GM const unsigned int*)(ar + (long long)s * p.lda + 4 * (i % mq));
          const int sh = 2 * (i / mq);
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) {
            const unsigned int code = (aw >> (8 * k2 + sh)) & 3u;
            const int bv = ub ? (int)((bw >> (8 * k2)) & 255u) : (int)(signed char)(bw >> (8 * k2));
            acc += (code == 0u) ? 0 : (code == 1u) ? bv : -bv;
          }
        }
      }
    }
    *c = acc;
    return;
  }

  if (p.a_type == LIBXSMM_DATATYPE_I16) {       // 16-bit integers -> i32, A optionally VNNI-2 [ref: gemm ref :1427-1450]
    if (!valid) return;
    const int kb = va ? 2 : 1;
    GM int* c = (GM int*)q.c + (long long)j * p.ldc + i;
    int acc = beta0 ? 0 : *c;
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      for (int s = 0; s < p.k; ++s)
        acc += (int)((GM const short*)ar)[(long long)(s / kb) * ((long long)p.lda * kb) + (long long)i * kb + (s % kb)] * (int)((GM const short*)br)[(long long)j * p.ldb + s];
    }
    *c = acc;
    return;
  }
  if (p.a_type == LIBXSMM_DATATYPE_I8 && p.b_type == LIBXSMM_DATATYPE_BF16) {
    // i8 weights with one f32 scale per row (a.tertiary) x bf16 activations [ref: gemm ref :1684-1730]: the scaled weight is rounded to bf16, the
    // products are summed from 0 in k order, beta * C comes last
    if (!valid) return;
    const float scf = ((GM const float*)(p.a_scf + (long long)bidx * p.bs_scf))[i];
    float acc = 0.0f;
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      for (int s = 0; s < p.k; ++s) {
        const float a_use = bf16_to_f32(f32_to_bf16_rne(mul_rn((float)(int)((GM const signed char*)ar)[(long long)s * p.lda + i], scf)));
        acc = add_rn(acc, mul_rn(a_use, bf16_to_f32(((GM const unsigned short*)br)[(long long)j * p.ldb + s])));
      }
    }
    if (p.c_type == LIBXSMM_DATATYPE_BF16) { GM unsigned short* c = (GM unsigned short*)q.c + (long long)j * p.ldc + i; if (!beta0) acc = add_rn(acc, bf16_to_f32(*c)); *c = f32_to_bf16_rne(acc); }
    else { GM float* c = (GM float*)q.c + (long long)j * p.ldc + i; if (!beta0) acc = add_rn(acc, *c); *c = acc; }
    return;
  }

  if (p.a_type == LIBXSMM_DATATYPE_I8 || p.a_type == LIBXSMM_DATATYPE_U8) {
    // 8-bit integer GEMM, i32 accumulation [ref: gemm ref :1452-1683]; A VNNI-4 (always for f32 output), B flat
    if (!valid) return;
    const bool ua = p.a_type == LIBXSMM_DATATYPE_U8, ub = p.b_type == LIBXSMM_DATATYPE_U8, c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
    const int kb4 = (c_f32 || va) ? 4 : 1;
    int acc = 0;
    if (!c_f32 && !beta0) acc = ((GM const int*)q.c)[(long long)j * p.ldc + i];
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      for (int s = 0; s < p.k; ++s) {
        const long long ai = (long long)(s / kb4) * ((long long)p.lda * kb4) + (long long)i * kb4 + (s % kb4), bi = (long long)j * p.ldb + s;
        const int av = ua ? (int)((GM const unsigned char*)ar)[ai] : (int)((GM const signed char*)ar)[ai];
        const int bv = ub ? (int)((GM const unsigned char*)br)[bi] : (int)((GM const signed char*)br)[bi];
        acc += av * bv;
      }
    }
    if (c_f32) {
      GM float* c = (GM float*)q.c + (long long)j * p.ldc + i;
      float f = mul_rn((float)acc, p.scf);
      if (!beta0) f = add_rn(f, *c);
      *c = f;
    } else ((GM int*)q.c)[(long long)j * p.ldc + i] = acc;
    return;
  }

  if (is_mx_type(p.a_type) && p.b_type == p.a_type) {
    // MX x MX [ref: gemm ref :2620-2665 (fp8), :2731-2785 (fp4)]: A and B as a dword per (row, k-group), E8M0 scales per (32 k, row) in
    // a/b.tertiary; partial sums per k-group (fp8: 4 products, high k first; fp4: 8 products ascending) times scale_a times scale_b
    if (!valid) return;
    const bool fp4 = p.a_type == LIBXSMM_DATATYPE_MXFP4X2, hf8 = p.a_type == LIBXSMM_DATATYPE_MXHF8;
    float acc = 0.0f;
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      GM const unsigned char* sa = (GM const unsigned char*)mx_scale_base(p, bidx, r, false);
      GM const unsigned char* sb = (GM const unsigned char*)mx_scale_base(p, bidx, r, true);
      if (fp4) {
        for (int s = 0; s < p.k / 32; ++s) {
          const float sca = __uint_as_float((unsigned int)sa[(long long)s * p.lda + i] << 23), scb = __uint_as_float((unsigned int)sb[(long long)s * p.ldb + j] << 23);
          for (int g = 0; g < 4; ++g) {
            float tmp = 0.0f;
            const unsigned int wa = *(GM const unsigned int*)(ar + ((long long)(s * 4 + g) * p.lda + i) * 4), wb = *(GM const unsigned int*)(br + ((long long)(s * 4 + g) * p.ldb + j) * 4);
            for (int k2 = 0; k2 < 8; ++k2) tmp = add_rn(tmp, mul_rn(e2m1_to_f32((wa >> (4 * k2)) & 15u), e2m1_to_f32((wb >> (4 * k2)) & 15u)));
            acc = add_rn(acc, mul_rn(mul_rn(tmp, sca), scb));
          }
        }
      } else if (is_fp6_type(p.a_type)) {      // [ref: gemm ref :2680-2727]: [k/4][ld][3 bytes], four 6-bit values per row and k-group, high k first
        const bool e3m2 = p.a_type == LIBXSMM_DATATYPE_MXBF6;
        for (int s = 0; s < p.k / 4; ++s) {
          const float sca = __uint_as_float((unsigned int)sa[(long long)(s / 8) * p.lda + i] << 23), scb = __uint_as_float((unsigned int)sb[(long long)(s / 8) * p.ldb + j] << 23);
          GM const unsigned char* pa = (GM const unsigned char*)ar + ((long long)s * p.lda + i) * 3; GM const unsigned char* pb = (GM const unsigned char*)br + ((long long)s * p.ldb + j) * 3;
          const unsigned int wa = (unsigned int)pa[0] | ((unsigned int)pa[1] << 8) | ((unsigned int)pa[2] << 16), wb = (unsigned int)pb[0] | ((unsigned int)pb[1] << 8) | ((unsigned int)pb[2] << 16);
          float tmp = 0.0f;
          for (int k2 = 3; k2 >= 0; --k2) tmp = add_rn(tmp, mul_rn(fp6_to_f32((wa >> (6 * k2)) & 63u, e3m2), fp6_to_f32((wb >> (6 * k2)) & 63u, e3m2)));
          acc = add_rn(acc, mul_rn(mul_rn(tmp, sca), scb));
        }
      } else {
        for (int s = 0; s < p.k / 4; ++s) {
          const float sca = __uint_as_float((unsigned int)sa[(long long)(s / 8) * p.lda + i] << 23), scb = __uint_as_float((unsigned int)sb[(long long)(s / 8) * p.ldb + j] << 23);
          const unsigned int wa = *(GM const unsigned int*)(ar + ((long long)s * p.lda + i) * 4), wb = *(GM const unsigned int*)(br + ((long long)s * p.ldb + j) * 4);
          float tmp = 0.0f;
          for (int k2 = 3; k2 >= 0; --k2) {
            const unsigned char xa = (unsigned char)(wa >> (8 * k2)), xb = (unsigned char)(wb >> (8 * k2));
            tmp = add_rn(tmp, mul_rn(hf8 ? hf8_to_f32(xa) : bf8_to_f32(xa), hf8 ? hf8_to_f32(xb) : bf8_to_f32(xb)));
          }
          acc = add_rn(acc, mul_rn(mul_rn(tmp, sca), scb));
        }
      }
    }
    GM float* c = (GM float*)q.c + (long long)j * p.ldc + i;
    *c = add_rn(beta0 ? 0.0f : *c, acc);
    return;
  }

  if (p.a_type == LIBXSMM_DATATYPE_MXFP4X2) {
    // MXFP4 weights x bf16/f32 activations [ref: gemm ref :949-1008]: packed E2M1 pairs [k/2][lda] (low nibble = even k), value * scale
    // is exact, products summed serially from 0 (unfused), C = (beta ? C : 0) + sum with one RNE for bf16 output
    if (!valid) return;
    float acc = 0.0f;
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      gcptr sr = mx_scale_base(p, bidx, r, false);
      for (int s = 0; s < p.k / 32; ++s) {
        const float scf = __uint_as_float((unsigned int)((GM const unsigned char*)sr)[(long long)s * p.lda + i] << 23);
        for (int k2 = 0; k2 < 32; k2 += 2) {
          const unsigned int pk = ((GM const unsigned char*)ar)[(long long)(s * 32 + k2) * p.lda / 2 + i];
          const long long bi = (long long)j * p.ldb + s * 32 + k2;
          acc = add_rn(acc, mul_rn(mul_rn(e2m1_to_f32(pk & 15u), scf), load_c_f32(br, bi, p.b_type)));
          acc = add_rn(acc, mul_rn(mul_rn(e2m1_to_f32(pk >> 4), scf), load_c_f32(br, bi + 1, p.b_type)));
        }
      }
    }
    const long long ci = (long long)j * p.ldc + i;
    const float base = beta0 ? 0.0f : load_c_f32(q.c, ci, p.c_type);
    const float y = add_rn(base, acc);
    if (p.c_type == LIBXSMM_DATATYPE_F32) ((GM float*)q.c)[ci] = y; else ((GM unsigned short*)q.c)[ci] = f32_to_bf16_rne(y);
    return;
  }

  float acc = 0.0f;
  if (valid) {
    // 8-bit floats x themselves: VNNI-4, k ascending; 8-bit float weights x bf16 run the bf16 loop (pairs, high k first) [ref: gemm ref :2171-2366]
    const bool fp8 = (p.a_type == LIBXSMM_DATATYPE_BF8 || p.a_type == LIBXSMM_DATATYPE_HF8) && p.b_type == p.a_type;
    const bool fp8w = (p.a_type == LIBXSMM_DATATYPE_BF8 || p.a_type == LIBXSMM_DATATYPE_HF8) && p.b_type == LIBXSMM_DATATYPE_BF16;
    const int kb = fp8 ? (va ? 4 : 1) : (((p.a_type == LIBXSMM_DATATYPE_BF16 || fp8w) && va) ? 2 : 1);
    if (!beta0) acc = (p.c_type == LIBXSMM_DATATYPE_BF8 || p.c_type == LIBXSMM_DATATYPE_HF8) ? load_as_f32(q.c, (long long)j * p.ldc + i, p.c_type) : load_c_f32(q.c, (long long)j * p.ldc + i, p.c_type);
    if (p.colbias) {
      const float bias = (p.c_type == LIBXSMM_DATATYPE_BF8 || p.c_type == LIBXSMM_DATATYPE_HF8) ? load_as_f32(q.d, i, p.c_type) : load_c_f32(q.d, i, p.c_type);
      acc = beta0 ? bias : add_rn(bias, acc);
    }
    for (unsigned long long r = 0; r < p.br_count; ++r) {
      gcptr ar, br; br_base(p, q, r, ar, br);
      for (int s = 0; s < p.k / kb; ++s) {
        for (int q2 = 0; q2 < kb; ++q2) {               // bf16 VNNI pair: high k first [ref: gemm ref :2144]; fp8 quad: ascending [:2436]
          const int k2 = fp8 ? q2 : kb - 1 - q2;
          const int kk = s * kb + k2;
          const long long ai = ta ? ((long long)i * p.lda + kk)
                                  : ((long long)(kk / kb) * ((long long)p.lda * kb) + (long long)i * kb + (kk % kb));
          const long long bi = (tb && vb) ? ((long long)j * kb + (long long)(kk / kb) * ((long long)p.ldb * kb) + (kk % kb))
                              : tb ? ((long long)kk * p.ldb + j) : ((long long)j * p.ldb + kk);
          acc = add_rn(acc, mul_rn(load_as_f32(ar, ai, p.a_type), load_as_f32(br, bi, p.b_type)));
        }
      }
    }
  }
  generic_epilogue(p, q, i, j, valid, acc);
}
#endif

// One long batch-reduce chain split over the chip: `nsplit` partial C tiles (f32, [split][n][m]) were produced by the
// tile kernels; this pass adds them up in split order on top of beta*C (+ bias) and applies the epilogue of the
// original descriptor.  [the reference runs the chain serially, gemm ref :490-530; partial sums change the rounding
// order only]
// block = (64, 16): x walks i, the 16 y-slices share the splits of one column j (fixed, deterministic order: slice y
// sums splits y, y+16, ... with four independent chains; the slices are then added in y order through LDS)
#if !defined(XAMD_GEMM_SHARD)      // a plain (non-template) kernel: emitted by the main translation unit only
__global__ __launch_bounds__(1024) void brsplit_reduce_kernel(GemmArgs p, const float* partial, int nsplit) {
  __shared__ float part[16][64];
  const int tiles_i = (p.m + 63) / 64;
  const int t = (int)blockIdx.x;
  const int i = (t % tiles_i) * 64 + threadIdx.x;
  const int j = t / tiles_i;
  const bool valid = (i < p.m) && (j < p.n);
  float sum = 0.0f;
  if (valid) {
    GM const float* w = (GM const float*)partial + (long long)j * p.m + i;
    const long long stride = (long long)p.m * p.n;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    int s = threadIdx.y;
    for (; s + 48 < nsplit; s += 64) { c0 += w[s * stride]; c1 += w[(s + 16) * stride]; c2 += w[(s + 32) * stride]; c3 += w[(s + 48) * stride]; }
    for (; s < nsplit; s += 16) c0 += w[s * stride];
    sum = (c0 + c1) + (c2 + c3);
    // (Round 4, rocprofv3 kernel trace of variant B at br = 4096: this kernel lasts 4.8 us on 256 slabs of 4 KiB; a form with sixteen slab loads in flight per
    // thread measured the same there and slower on few large slabs -- 4.8 us is what ANY dependent kernel of a few workgroups costs behind a kernel
    // boundary here: a 16-workgroup launch of 16^3 problems lasts 3.3 - 4.2 us.)
  }
  part[threadIdx.y][threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.y != 0) return;
  const BatchPtrs q = batch_ptrs(p, 0);
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  float acc = 0.0f;
  if (valid) {
    if (!beta0) acc = load_c_f32(q.c, (long long)j * p.ldc + i, p.c_type);
    if (p.colbias) { const float bias = load_c_f32(q.d, i, p.c_type); acc = beta0 ? bias : add_rn(bias, acc); }
#pragma unroll
    for (int y = 0; y < 16; ++y) acc = add_rn(acc, part[y][threadIdx.x]);
  }
  generic_epilogue(p, q, i, j, valid, acc);
}
#endif

// ------------------------------------------------------------------------------------------------
// MFMA building blocks.  "X" operand: free index contiguous in memory, element (f,k) at base[f + k*ld]
// (A normal, B under TRANS_B).  "Y" operand: k contiguous, element (f,k) at base[k + f*ld] (B normal,
// A under TRANS_A).  Both deliver w[s] = op(f = lane&31, k = k0 + 2s + (lane>>5)), s = 0..15, the
// operand layout of v_mfma_f32_32x32x2_f32 in natural k order.
// ------------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ __forceinline__ void load_x_f32(float (&w)[16], GM const float* base, long long ld, int f, bool fvalid, int k0, int K, int h) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = k0 + 2 * s + h;
    w[s] = (EXACT || (fvalid && k < K)) ? base[f + (long long)k * ld] : 0.0f;
  }
}
template <bool EXACT>
__device__ __forceinline__ void load_y_f32(float (&w)[16], GM const float* base, long long ld, int f, bool fvalid, int k0, int K, int h) {
  GM const float* col = base + (long long)f * ld + k0 + 16 * h;   // this lane's 16 consecutive k
  float v[16];
  // wave-uniform alignment test: 16-byte loads need ld % 4 == 0 and a 16-byte aligned base
  const bool vec = ((((unsigned long long)(size_t)base) & 15ull) == 0ull) && ((ld & 3) == 0) && ((k0 & 3) == 0);
  if (EXACT && vec) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = *(GM const f32x4*)(col + 4 * q);
      v[4 * q + 0] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (EXACT || (fvalid && (k0 + 16 * h + e) < K)) ? col[e] : 0.0f;
  }
  // lanes 0-31 hold k0..k0+15, lanes 32-63 hold k0+16..k0+31; one half-wave exchange per register
  // pair interleaves them: afterwards lane half h owns k0 + 2s + h.
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
    w[s] = __uint_as_float(r[0]);       // [lower.v[2s]   | lower.v[2s+1]]   -> k0 + 2s + h
    w[s + 8] = __uint_as_float(r[1]);   // [upper.v[2s]   | upper.v[2s+1]]   -> k0 + 16 + 2s + h
  }
}

// Coalesced variant for exact tiles (measured with tools/gemm_probe.hip: +15% at batch 4096, equal to a
// plain copy kernel at large batches): a 32x32 f32 operand tile is fetched with four fully coalesced
// 16-byte-per-lane loads (whole 128-byte rows), parked in a wave-private 4 KiB LDS image and read back in
// MFMA fragment order.  No barrier: the image is written and read by the same wave.
//   KCONTIG == false (free index contiguous): image [k][f] linear, fragment = 16 conflict-free ds_read_b32.
//   KCONTIG == true  (k contiguous): image [f][8 chunks of 4 k], chunk index XOR-swizzled with (f>>1)&7 so that
//   the per-column ds_read_b128 is conflict free; v_permlane32_swap then interleaves the two k halves.
template <bool KCONTIG>
__device__ __forceinline__ void tile_gload(f32x4 (&g)[4], GM const float* base, long long ld, int f0, int k0, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = lane + 64 * q, hi = t >> 3, lo = (t & 7) * 4;
    GM const float* src = KCONTIG ? base + (long long)(f0 + hi) * ld + k0 + lo : base + (long long)(k0 + hi) * ld + f0 + lo;
    g[q] = *(GM const f32x4*)src;
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void tile_to_frag(float (&w)[16], const f32x4 (&g)[4], float* lds, int lane) {
  const int li = lane & 31, h = lane >> 5;
  if (!KCONTIG) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ((f32x4*)lds)[lane + 64 * q] = g[q];
#pragma unroll
    for (int s = 0; s < 16; ++s) w[s] = lds[(2 * s + h) * 32 + li];
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = lane + 64 * q, f = t >> 3, c = (t & 7) ^ ((f >> 1) & 7);
      ((f32x4*)lds)[f * 8 + c] = g[q];
    }
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 x = ((const f32x4*)lds)[li * 8 + ((4 * h + q) ^ ((li >> 1) & 7))];
      v[4 * q + 0] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
      w[s] = __uint_as_float(r[0]); w[s + 8] = __uint_as_float(r[1]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// f32, 32x32 MFMA tiles, MT x NT tiles per wave
// ------------------------------------------------------------------------------------------------
// MODE 0: arbitrary m/n/k (masked loads/stores); 1: exact tiles, operands fetched straight into fragment
// layout (any alignment); 2: exact tiles with 16-byte aligned operands, staged through LDS (fastest).
enum { GM_MASKED = 0, GM_EXACT = 1, GM_STAGED = 2 };
template <int MT, int NT, bool TA, bool TB, int MODE>
__global__ __launch_bounds__(256) void gemm_mfma_f32_kernel(GemmArgs p) {
  constexpr bool EXACT = MODE != GM_MASKED;
  constexpr bool STAGED = MODE == GM_STAGED;
  // wave-private staging images for the coalesced path: (MT + NT) tiles of 4 KiB per wave
  constexpr int kLdsFloats = STAGED ? (MT + NT) * 1024 : 4;
  __shared__ __attribute__((aligned(16))) float lds_all[4][kLdsFloats];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  float* lds = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h;
      tc[mt][nt].ivalid = tc[mt][nt].i < p.m;
      tile_init<EXACT, true>(acc[mt][nt], p, q, tc[mt][nt]);
    }
  const int kchunks = (p.k + 31) / 32;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    GM const float* A = (GM const float*)ar; GM const float* B = (GM const float*)br;
    for (int kc = 0; kc < kchunks; ++kc) {
      const int k0 = kc * 32;
      float af[MT][16], bf[NT][16];
      if constexpr (STAGED) {
        // all global loads of the chunk are issued before the first LDS access
        f32x4 ga[MT][4], gb[NT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) tile_gload<TA>(ga[mt], A, p.lda, job.i0 + 32 * mt, k0, lane);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) tile_gload<!TB>(gb[nt], B, p.ldb, job.j0 + 32 * nt, k0, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) tile_to_frag<TA>(af[mt], ga[mt], lds + 1024 * mt, lane);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) tile_to_frag<!TB>(bf[nt], gb[nt], lds + 1024 * (MT + nt), lane);
      } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int i = job.i0 + 32 * mt + li;
        if (TA) load_y_f32<EXACT>(af[mt], A, p.lda, i, i < p.m, k0, p.k, h);
        else load_x_f32<EXACT>(af[mt], A, p.lda, i, i < p.m, k0, p.k, h);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int j = job.j0 + 32 * nt + li;
        if (TB) load_x_f32<EXACT>(bf[nt], B, p.ldb, j, j < p.n, k0, p.k, h);
        else load_y_f32<EXACT>(bf[nt], B, p.ldb, j, j < p.n, k0, p.k, h);
      }
      }
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[nt][s], af[mt][s], acc[mt][nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) tile_store<EXACT, true>(acc[mt][nt], p, q, tc[mt][nt]);
}

// ------------------------------------------------------------------------------------------------
// f32 streaming kernel: the hot path for exact 32x32 tiles with 16-byte aligned operands (the batched
// small-GEMM case).  Same algorithm as gemm_mfma_f32_kernel<1,1,.,.,GM_STAGED>, written for minimum
// instruction count because one launch is a single round of waves that all run their prologue and
// epilogue at the same time (issue-bound, not latency-bound: 4 waves per SIMD x instructions per wave):
//   * all tile bases are wave-uniform SGPR pointers, lanes only carry a 32-bit byte offset
//     (global_load/store saddr + voffset form, no 64-bit VALU address math),
//   * one flattened loop over (batch-reduce element, 32-deep K chunk),
//   * the epilogue variants (beta / bias / activation / bitmask) sit behind wave-uniform branches.
// Measured with tools/gemm_probe.hip; PMC: 86 VALU + 5 SALU instructions per wave for the probe kernel.
// ------------------------------------------------------------------------------------------------
// BF32 (round 3): f32 storage whose operands the reference rounds to bf16 before multiplying [ref: gemm ref :1366, :1384-1389] -- the loaded vectors are
// rounded in registers (the reference's RNE: software, so denormals and NaNs agree too); products of two bf16 values are exact in f32, so the MFMA's
// k-ordered chain equals the reference's loop bit for bit.
__device__ __forceinline__ f32x4 round_to_bf16(f32x4 v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __uint_as_float((unsigned int)f32_to_bf16_rne(v[e]) << 16);
  return v;
}
template <bool TA, bool TB, bool BF32 = false>
__global__ __launch_bounds__(256) void gemm_f32_stream_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds_all[4][2048];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = logical_block(p) * 4u + wave;
  const unsigned int per_gemm = (unsigned int)(p.tiles_m * p.tiles_n);
  if (wid >= per_gemm * p.nbatch) return;
  unsigned int bidx = wid, i0 = 0, j0 = 0;
  if (per_gemm != 1) {
    bidx = wid / per_gemm;
    const unsigned int t = wid - bidx * per_gemm, tn = t / (unsigned int)p.tiles_m;
    i0 = (t - tn * (unsigned int)p.tiles_m) * 32u; j0 = tn * 32u;
  }
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  float* lds = lds_all[wave];
  const BatchPtrs q = batch_ptrs(p, bidx);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb, ldc = (unsigned int)p.ldc;
  // per-lane byte offsets inside a 32x32 operand tile for the four 16-byte loads (rows 8q + lane/8)
  const unsigned int offA = ((lane >> 3) * lda + (lane & 7u) * 4u) * 4u;
  const unsigned int offB = ((lane >> 3) * ldb + (lane & 7u) * 4u) * 4u;
  const unsigned long long stepA = 32ull * lda, stepB = 32ull * ldb;          // bytes per 8 rows
  // origin of this wave's operand tiles inside A_r / B_r (bytes), per K chunk add kstepX
  const unsigned long long orgA = TA ? 4ull * i0 * lda : 4ull * i0, kstepA = TA ? 128ull : 128ull * lda;
  const unsigned long long orgB = TB ? 4ull * j0 : 4ull * j0 * ldb, kstepB = TB ? 128ull * ldb : 128ull;
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  gptr ctile = q.c + 4ull * ((unsigned long long)j0 * ldc + i0);
  const unsigned int offC = (4u * h * ldc + li) * 4u;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  if (!beta0 || p.colbias) {
    const float bias = p.colbias ? ((GM const float*)q.d)[i0 + li] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float start = beta0 ? 0.0f : *(GM const float*)(ctile + (unsigned long long)(((r & 3) + 8 * (r >> 2)) * ldc) * 4ull + offC);
      acc[r] = p.colbias ? (beta0 ? bias : bias + start) : start;
    }
  }
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  unsigned long long r = 0; unsigned int kc = 0;
  gcptr ar, br;
  if (p.br_count != 0) br_base(p, q, 0, ar, br);
  const unsigned long long total = p.br_count * kchunks;
  // Software pipeline over the (batch-reduce element, K chunk) sequence: the global loads of chunk t+1 are issued as soon as chunk t
  // has been parked in LDS, i.e. BEFORE the 16 MFMAs of chunk t (1024 cycles of matrix pipe) -- with operands that other waves
  // have already pulled into L2 (2-D batches, long chains) the wave then never waits for memory.  One chunk (br = 1, k = 32:
  // the streaming headline) runs the same instruction sequence as before.
  f32x4 ga[4], gb[4];
  if (total != 0) {
    gcptr au = ar + orgA, bu = br + orgB;                                   // wave-uniform
#pragma unroll
    for (int x = 0; x < 4; ++x) ga[x] = *(GM const f32x4*)(au + x * stepA + offA);
#pragma unroll
    for (int x = 0; x < 4; ++x) gb[x] = *(GM const f32x4*)(bu + x * stepB + offB);
  }
  for (unsigned long long t = 0; t < total; ++t) {
    float af[16], bf[16];
    if constexpr (BF32) {
#pragma unroll
      for (int x = 0; x < 4; ++x) { ga[x] = round_to_bf16(ga[x]); gb[x] = round_to_bf16(gb[x]); }
    }
    tile_to_frag<TA>(af, ga, lds, (int)lane);
    tile_to_frag<!TB>(bf, gb, lds + 1024, (int)lane);
    if (++kc == kchunks) { kc = 0; if (++r < p.br_count) br_base(p, q, r, ar, br); }
    if (t + 1 < total) {
      gcptr au = ar + orgA + kc * kstepA, bu = br + orgB + kc * kstepB;
#pragma unroll
      for (int x = 0; x < 4; ++x) ga[x] = *(GM const f32x4*)(au + x * stepA + offA);
#pragma unroll
      for (int x = 0; x < 4; ++x) gb[x] = *(GM const f32x4*)(bu + x * stepB + offB);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
  }
  if (p.act == 0) {
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2)
      st_stream((GM float*)(ctile + (unsigned long long)(((r2 & 3) + 8 * (r2 >> 2)) * ldc) * 4ull + offC), acc[r2]);
  } else {
    const unsigned int mask_row = (((ldc + 15u) / 16u) * 16u) / 8u;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
      const unsigned int jr = (r2 & 3) + 8 * (r2 >> 2);
      const float x = acc[r2];
      st_stream((GM float*)(ctile + (unsigned long long)(jr * ldc) * 4ull + offC), act_apply(p.act, x));
      if (p.act == 2 && q.mask) {
        const unsigned long long pos = __ballot(!(x <= 0.0f));
        if ((lane & 7u) == 0u) q.mask[(i0 + li) / 8u + (unsigned long long)(j0 + jr + 4u * h) * mask_row] = (unsigned char)((pos >> lane) & 0xffu);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same algorithm for the case the headline benchmark is: a 1-D batch of independent 32x32x(32 br kchunks) problems with 16-byte
// aligned strided operands, beta = 0 and no fused epilogue.  A launch of 4096 such problems is ONE round of waves that lasts ~10 us, so the
// time every wave spends before its first load is issued is on the critical path of the whole launch.  The general kernel above reads a
// 280-byte argument block in several dependent scalar loads, divides by the tile count and walks the batch / batch-reduce / epilogue
// options; this one takes an 88-byte block (one scalar load round trip), has no option left to test, and in its single-chunk form (br = 1,
// k = 32) has no loop either.
// NTL: non-temporal operand loads.  Measured (tools/headline_probe.hip, profiles/r02_copy_floor.csv): with operands coming from HBM they
// save 0.5 us of a 9.9 us launch (they do not displace the Infinity Cache's contents), but operands that ARE resident in the 256 MiB
// Infinity Cache -- the normal case for a 48 MiB batch produced by the previous kernel -- are then not kept there: 8.7 instead of 5.8 us.
// The launcher therefore asks for them only when one launch moves more than the Infinity Cache holds (they cannot be resident then).
// ------------------------------------------------------------------------------------------------
struct LeanF32Args {
  const char* a; const char* b; char* c;
  long long bs_a, bs_b, bs_c, brs_a, brs_b;          // batch strides, batch-reduce strides (bytes)
  unsigned int nbatch, nchunks, kchunks, lda, ldb, ldc;   // nchunks = br_count * kchunks
};
// 16-byte operand load through a wave-uniform buffer resource with gfx950 cache-policy bits (aux: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX> __device__ __forceinline__ f32x4 ld16_pol(__amdgpu_buffer_rsrc_t r, unsigned int voffset) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, 0, AUX));
}
// POL 0: operands loaded `sc0 sc1`, C leaves as whole 16-byte pieces through the wave's LDS image, non-temporal.  Measured on the headline
//        footprint (tools/policy_probe.hip, profiles/r02_cache_policy.txt): against plain loads + dword nt stores 9.95 vs 10.88 us with the
//        operands in HBM AND 6.25 vs 6.7 us with the operands resident in the Infinity Cache -- no trade-off, so it is the default.
// POL 1: operands loaded `nt`, C as dword nt stores: 9.65 us from HBM but 8.4 us on resident operands (an nt read is not kept in the
//        Infinity Cache): only for launches that move more than the Infinity Cache holds, whose operands cannot be resident anyway.
// POL 2: plain loads, dword nt stores (C not 16-byte aligned).
// POL 3 (round 4): POL 1's nt loads with POL 0's 16-byte stores through the LDS image -- tools/headline_probe: a copy of this footprint with nt loads and 16-byte nt
//        stores takes 8.99 us where the POL 1 kernel takes 9.99 (dword stores: four times the store instructions).
template <bool TA, bool TB, bool SINGLE, int POL>
__global__ __launch_bounds__(256) void gemm_f32_stream_kernel_lean(LeanF32Args p) {
  constexpr int AUX = POL == 0 ? 17 : ((POL == 1 || POL == 3) ? 2 : 0);
  __shared__ __attribute__((aligned(16))) float lds_all[4][2048];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int bidx = blockIdx.x * 4u + wave;
  if (bidx >= p.nbatch) return;
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  float* lds = lds_all[wave];
  const unsigned int lda = p.lda, ldb = p.ldb, ldc = p.ldc;
  gcptr ar = (gcptr)p.a + (long long)bidx * p.bs_a, br = (gcptr)p.b + (long long)bidx * p.bs_b;
  const unsigned int offA = ((lane >> 3) * lda + (lane & 7u) * 4u) * 4u;
  const unsigned int offB = ((lane >> 3) * ldb + (lane & 7u) * 4u) * 4u;
  const unsigned int stepA = 32u * lda, stepB = 32u * ldb;          // bytes per 8 rows
  f32x4 ga[4], gb[4];
  {
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar), rb = wave_rsrc(br);
#pragma unroll
    for (int x = 0; x < 4; ++x) ga[x] = ld16_pol<AUX>(ra, x * stepA + offA);
#pragma unroll
    for (int x = 0; x < 4; ++x) gb[x] = ld16_pol<AUX>(rb, x * stepB + offB);
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  if (SINGLE) {
    float af[16], bf[16];
    tile_to_frag<TA>(af, ga, lds, (int)lane);
    tile_to_frag<!TB>(bf, gb, lds + 1024, (int)lane);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
  } else {
    const unsigned long long kstepA = TA ? 128ull : 128ull * lda, kstepB = TB ? 128ull * ldb : 128ull;
    unsigned int kc = 0;
    for (unsigned int t = 0; t < p.nchunks; ++t) {
      float af[16], bf[16];
      tile_to_frag<TA>(af, ga, lds, (int)lane);
      tile_to_frag<!TB>(bf, gb, lds + 1024, (int)lane);
      if (++kc == p.kchunks) { kc = 0; ar += p.brs_a; br += p.brs_b; }
      if (t + 1 < p.nchunks) {        // chunk t+1 is in flight while the matrix core works on chunk t
        const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + kc * kstepA), rb = wave_rsrc(br + kc * kstepB);
#pragma unroll
        for (int x = 0; x < 4; ++x) ga[x] = ld16_pol<AUX>(ra, x * stepA + offA);
#pragma unroll
        for (int x = 0; x < 4; ++x) gb[x] = ld16_pol<AUX>(rb, x * stepB + offB);
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
    }
  }
  gptr ctile = (gptr)p.c + (long long)bidx * p.bs_c;
  if (POL == 0 || POL == 3) {
    // C tile -> column-major LDS image (lanes along i: conflict free) -> whole 128-byte columns, 16 bytes per lane
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) lds[li + (unsigned int)jl_of(r2, (int)h) * 32u] = acc[r2];
    const __amdgpu_buffer_rsrc_t rc = wave_rsrc((gcptr)ctile);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const unsigned int L = lane + 64u * x;
      __builtin_amdgcn_raw_buffer_store_b128(((const u32x4*)lds)[L], rc, (int)(((L >> 3) * ldc + (L & 7u) * 4u) * 4u), 0, 2);
    }
  } else {
    const unsigned int offC = (4u * h * ldc + li) * 4u;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2)
      st_stream((GM float*)(ctile + (unsigned long long)(((r2 & 3) + 8 * (r2 >> 2)) * ldc) * 4ull + offC), acc[r2]);
  }
}

// ------------------------------------------------------------------------------------------------
// ONE long STRIDE batch-reduce chain (SURVEY 8(d) config #2 variant B: a single BRGEMM with br = 4096): the chain is cut into slices, a workgroup
// of EIGHT waves takes 8 * chunk consecutive blocks of one 32 x 32 output tile (wave w: blocks w * chunk .. + chunk - 1 of the slice, the loop of
// gemm_f32_stream_kernel), adds the eight accumulators up through LDS in wave order and writes ONE partial tile; brsplit_reduce_kernel adds the
// partial tiles in slice order.  Round 2 ran the chain as 1024 waves of 4 blocks with one partial tile per wave (0.21 of the HBM roofline: a
// quarter of the waves the chip wants, 4 MiB of partial sums written and read back); here br = 4096 is 4096 waves and 512 partial tiles.
// f32, NN, whole 32 x 32 tiles, k % 32 == 0.  partial: [slice][n][m] f32.
// Round 5, built, verified and NOT adopted (the review's "variant B in a single launch"; the patch is profiles/r05_variant_b_single_launch_not_adopted.patch, the numbers
// profiles/r05_variant_b.jsonl): the sum over the slices in THIS launch -- arrival counters, two deterministic levels (the last workgroup of 16 slices adds the group up, the
// last group adds the groups up and applies the epilogue), nobody waits.  With device-scope fences (__threadfence: write-back + invalidate of the XCD's whole L2 per
// workgroup) br = 4096 went from 12.7 to 80 us; with every partial-tile access an agent-scope relaxed atomic (sc1: coherent line by line, no cache maintenance) 27 us; with
// sixteen such loads in flight 17.3 us (br = 1024: 11.9 against 8.7, br = 65 536: 103 against 101).  Data that crosses the eight L2s inside a launch travels through memory
// twice per level -- as long as a kernel boundary (4.8 us) -- so the second launch stays.
// ------------------------------------------------------------------------------------------------
#if !defined(XAMD_GEMM_SHARD)      // a plain (non-template) kernel: emitted by the main translation unit only
__global__ __launch_bounds__(512) void gemm_f32_brchain_kernel(GemmArgs p, float* partial, unsigned int chunk, unsigned int nslices) {
  __shared__ __attribute__((aligned(16))) float lds_all[8][2048];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int per_gemm = (unsigned int)(p.tiles_m * p.tiles_n);
  const unsigned int slice = blockIdx.x / per_gemm, t = blockIdx.x - slice * per_gemm, tn = t / (unsigned int)p.tiles_m;
  const unsigned int i0 = (t - tn * (unsigned int)p.tiles_m) * 32u, j0 = tn * 32u;
  float* lds = lds_all[wave];
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int offA = ((lane >> 3) * lda + (lane & 7u) * 4u) * 4u, offB = ((lane >> 3) * ldb + (lane & 7u) * 4u) * 4u;
  const unsigned long long stepA = 32ull * lda, stepB = 32ull * ldb, orgA = 4ull * i0, kstepA = 128ull * lda, orgB = 4ull * j0 * ldb, kstepB = 128ull;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  const unsigned long long first = ((unsigned long long)slice * 8ull + wave) * chunk;
  const unsigned long long last = first + chunk < p.br_count ? first + chunk : p.br_count;           // (first >= br_count: nothing to do, the wave contributes zeros)
  if (first < last) {
    gcptr ar = (gcptr)p.a + p.br_stride_a * (long long)first, br = (gcptr)p.b + p.br_stride_b * (long long)first;
    const unsigned long long total = (last - first) * kchunks;
    unsigned int kc = 0;
    f32x4 ga[4], gb[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) ga[x] = *(GM const f32x4*)(ar + orgA + x * stepA + offA);
#pragma unroll
    for (int x = 0; x < 4; ++x) gb[x] = *(GM const f32x4*)(br + orgB + x * stepB + offB);
    for (unsigned long long u = 0; u < total; ++u) {
      float af[16], bf[16];
      tile_to_frag<false>(af, ga, lds, (int)lane);
      tile_to_frag<true>(bf, gb, lds + 1024, (int)lane);
      if (++kc == kchunks) { kc = 0; ar += p.br_stride_a; br += p.br_stride_b; }
      if (u + 1 < total) {
        gcptr au = ar + orgA + kc * kstepA, bu = br + orgB + kc * kstepB;
#pragma unroll
        for (int x = 0; x < 4; ++x) ga[x] = *(GM const f32x4*)(au + x * stepA + offA);
#pragma unroll
        for (int x = 0; x < 4; ++x) gb[x] = *(GM const f32x4*)(bu + x * stepB + offB);
      }
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s2], af[s2], acc, 0, 0, 0);
    }
  }
  // the eight accumulators -> one partial tile: register r of lane l goes to word r * 64 + l of the wave's image (the operand images are done with)
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) lds[r * 64 + (int)lane] = acc[r];
  __syncthreads();
  float* tile = partial + (unsigned long long)slice * (unsigned long long)p.m * (unsigned long long)p.n;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const unsigned int r = 2u * wave + (unsigned int)rr;
    float sum = lds_all[0][r * 64 + lane];
#pragma unroll
    for (int w2 = 1; w2 < 8; ++w2) sum += lds_all[w2][r * 64 + lane];              // wave order: deterministic
    const unsigned int j = j0 + (r & 3u) + 8u * (r >> 2) + 4u * h;
    tile[(unsigned long long)j * (unsigned int)p.m + i0 + li] = sum;
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// f32 streaming kernel, LDS-DMA form (MT x NT tiles of 32x32 per wave).  Same arithmetic as gemm_f32_stream_kernel;
// the operand tiles of a 32-deep K chunk travel global -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no
// ds_write), in the same two LDS images (linear [k][f] for the operand whose free index is contiguous, XOR-swizzled
// [f][8 x 16 B] for the k-contiguous one -- the swizzle is applied to the SOURCE address because the DMA destination
// is lane-linear).  Per chunk: wait for the DMA, read the fragments into registers, immediately start the DMA of the
// NEXT chunk into the same image, then run the MFMAs: the matrix core hides the fetch latency without a second
// buffer.  That is what lifts the 64x64 tile (two chunks per problem, 64 MFMAs per chunk) off its latency floor.
// ------------------------------------------------------------------------------------------------
template <bool KCONTIG>
__device__ __forceinline__ void frag_read(float (&w)[16], const float* lds, int lane) {
  const int li = lane & 31, h = lane >> 5;
  if (!KCONTIG) {
#pragma unroll
    for (int s = 0; s < 16; ++s) w[s] = lds[(2 * s + h) * 32 + li];
  } else {
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 x = ((const f32x4*)lds)[li * 8 + ((4 * h + q) ^ ((li >> 1) & 7))];
      v[4 * q + 0] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
      w[s] = __uint_as_float(r[0]); w[s + 8] = __uint_as_float(r[1]);
    }
  }
}
template <int MT, int NT, bool TA, bool TB, int AUX = 0>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds_all[4][(MT + NT) * 1024];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  float* lds = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
  static_for<MT * NT>([&](auto idx) {
    constexpr int mt = idx.value / NT, nt = idx.value % NT;
    tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h; tc[mt][nt].ivalid = true;
    tile_init<true, true>(acc[mt][nt], p, q, tc[mt][nt]);
  });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  // per-lane byte offset of the 16-byte piece that lands in LDS slot (lane + 64x) of a tile image
  unsigned int offA[4], offB[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const unsigned int L = (unsigned int)lane + 64u * x, hi = L >> 3, lo = L & 7u;
    offA[x] = TA ? (hi * lda + ((lo ^ ((hi >> 1) & 7u)) * 4u)) * 4u : (hi * lda + lo * 4u) * 4u;       // TA: k contiguous (swizzled image)
    offB[x] = TB ? (hi * ldb + lo * 4u) * 4u : (hi * ldb + ((lo ^ ((hi >> 1) & 7u)) * 4u)) * 4u;
  }
  // origin of tile (mt / nt) and step per K chunk, bytes
  const unsigned long long orgA = TA ? 4ull * job.i0 * lda : 4ull * job.i0, tileA = TA ? 128ull * lda : 128ull, kstepA = TA ? 128ull : 128ull * lda;
  const unsigned long long orgB = TB ? 4ull * job.j0 : 4ull * job.j0 * ldb, tileB = TB ? 128ull : 128ull * ldb, kstepB = TB ? 128ull * ldb : 128ull;
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  const unsigned long long total = p.br_count * kchunks;
  unsigned long long r = 0; unsigned int kc = 0;
  gcptr ar, br;
  if (p.br_count != 0) br_base(p, q, 0, ar, br);
  auto issue = [&](gcptr a0, gcptr b0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int x = 0; x < 4; ++x)
        __builtin_amdgcn_global_load_lds((GM const void*)(a0 + mt * tileA + offA[x]), (lds_vptr)((char*)lds + 4096 * mt + 1024 * x), 16, 0, AUX);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int x = 0; x < 4; ++x)
        __builtin_amdgcn_global_load_lds((GM const void*)(b0 + nt * tileB + offB[x]), (lds_vptr)((char*)lds + 4096 * (MT + nt) + 1024 * x), 16, 0, AUX);
  };
  if (total != 0) issue(ar + orgA, br + orgB);
  for (unsigned long long t = 0; t < total; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float af[MT][16], bf[NT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) frag_read<TA>(af[mt], lds + 1024 * mt, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) frag_read<!TB>(bf[nt], lds + 1024 * (MT + nt), lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (++kc == kchunks) { kc = 0; if (++r < p.br_count) br_base(p, q, r, ar, br); }
    if (t + 1 < total) issue(ar + orgA + kc * kstepA, br + orgB + kc * kstepB);
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[nt][s], af[mt][s], acc[mt][nt], 0, 0, 0);
  }
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<true, true>(acc[mt][nt], p, q, tc[mt][nt]); });
}

// ------------------------------------------------------------------------------------------------
// f32 blocked kernel: the operand-reuse regime (libxsmm_hip_gemm_batch_strided_2d -- a blocked GEMM out of (BR)GEMM tiles).
// With one tile per wave and wave-private operands every wave pulls its own 8 KiB per 16 matrix instructions out of L2: measured
// (profiles/r02_wave_breakdown.json) 41-52 % matrix-core busy at 8 TB/s of L2 traffic, the waves stalled on instruction issue.  Here a
// WORKGROUP owns a 128 x 128 macro tile of C (4 x 4 problems of 32^3, or 2 x 2 of 64^3) and its four waves a 64 x 64 quarter each:
//   * per 32-deep K step the workgroup brings in 4 A blocks + 4 B blocks (32 KiB) ONCE, by LDS-DMA (global_load_lds_dwordx4: no
//     staging VGPRs, no ds_write), wave w fetching A block w and B block w, into one half of a 64 KiB double buffer;
//   * every wave reads the two A and two B fragments it needs (each block is used by two waves), then issues its share of the NEXT
//     step's DMA into the other half and runs 64 MFMAs (4096 matrix-pipe cycles) on four independent accumulators;
//   * one workgroup barrier per step: a buffer half is only rewritten after every wave has passed the barrier that follows its last read.
// L2 -> CU traffic per matrix instruction drops 4x against the one-tile-per-wave kernels; images and fragment order are the same as in
// gemm_f32_dma_kernel, so results stay bitwise the k-ordered fma chain.  NN layout, STRIDE or no batch-reduce, any epilogue.
// MB = 32-blocks per problem edge (1: 32^3 problems, 2: 64^3 problems; K any multiple of 32).
// ------------------------------------------------------------------------------------------------
template <int MB>
__global__ __launch_bounds__(256) void gemm_f32_blocked_kernel(GemmArgs p) {
  constexpr int PPW = 4 / MB;                       // problems per workgroup edge
  __shared__ __attribute__((aligned(16))) float lds_all[2][8][1024];
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  // macro tile of this workgroup: contiguous bands of macro columns per XCD (hardware workgroup g runs on XCD g % 8)
  const unsigned int ni = p.batch_inner, nj = p.nbatch / p.batch_inner, MI = ni / PPW;
  unsigned int g = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) g = (g & 7u) * (gridDim.x >> 3) + (g >> 3);
  const unsigned int mj = g / MI, mi = g - mj * MI;
  (void)nj;
  // --- DMA duty of this wave: A block w (32 rows x 32 k) and B block w (32 k x 32 columns) of the macro tile, every step
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int pa = mi * PPW + w / MB, pb = mj * PPW + w / MB;                   // problem that block w belongs to
  gcptr a_blk = (gcptr)p.a + (long long)pa * p.bs_a + 128ull * (w % MB);               // + 32 rows per sub-block
  gcptr b_blk = (gcptr)p.b + (long long)pb * p.bs_b + 128ull * ldb * (w % MB);         // + 32 columns per sub-block
  unsigned int offA[4], offB[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const unsigned int L = lane + 64u * x, hi = L >> 3, lo = L & 7u;
    offA[x] = (hi * lda + lo * 4u) * 4u;                                         // image [k][i] linear
    offB[x] = (hi * ldb + ((lo ^ ((hi >> 1) & 7u)) * 4u)) * 4u;                    // image [j][8 x 16 B], chunk XOR-swizzled
  }
  const unsigned long long kstepA = 128ull * lda, kstepB = 128ull;
  const long long brs_a = p.br_mode == 3 ? p.br_stride_a : 0, brs_b = p.br_mode == 3 ? p.br_stride_b : 0;
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  const unsigned long long total = p.br_count * kchunks;
  auto issue = [&](gcptr a0, gcptr b0, int buf) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(a0 + offA[x]), (lds_vptr)((char*)&lds_all[buf][w][0] + 1024 * x), 16, 0, 0);
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(b0 + offB[x]), (lds_vptr)((char*)&lds_all[buf][4 + w][0] + 1024 * x), 16, 0, 0);
  };
  // --- compute duty: the 64 x 64 quarter (wi, wj) of the macro tile
  const unsigned int wi = w & 1u, wj = w >> 1;
  f32x16 acc[2][2];
  TileCtx tc[2][2];
  BatchPtrs q[2][2];
  static_for<4>([&](auto idx) {
    constexpr int mt = idx.value / 2, nt = idx.value % 2;
    const unsigned int bi = mi * PPW + (2 * wi + mt) / MB, bj = mj * PPW + (2 * wj + nt) / MB;
    q[mt][nt].a = nullptr; q[mt][nt].b = nullptr;
    q[mt][nt].c = (gptr)p.c + (long long)bi * p.bs_c + (long long)bj * p.bs_c2;
    q[mt][nt].d = p.d ? (gcptr)p.d + (long long)bi * p.bs_d : nullptr;
    q[mt][nt].mask = p.relu_mask ? (GM unsigned char*)p.relu_mask + (long long)bi * p.bs_mask + (long long)bj * p.bs_mask2 : nullptr;
    tc[mt][nt].i = (int)(32u * ((2 * wi + mt) % MB) + li); tc[mt][nt].j0 = (int)(32u * ((2 * wj + nt) % MB)); tc[mt][nt].h = (int)h; tc[mt][nt].ivalid = true;
    tile_init<true, true>(acc[mt][nt], p, q[mt][nt], tc[mt][nt]);
  });
  unsigned long long r = 0; unsigned int kc = 0;
  if (total != 0) issue(a_blk, b_blk, 0);
  for (unsigned long long t = 0; t < total; ++t) {
    const int buf = (int)(t & 1ull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA is ordered by the issuing wave's vmcnt only
    __syncthreads();                                   // ... and by a barrier for the other waves: step t's eight blocks are in LDS
    float af[2][16], bf[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) frag_read<false>(af[mt], &lds_all[buf][2 * wi + mt][0], (int)lane);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) frag_read<true>(bf[nt], &lds_all[buf][4 + 2 * wj + nt][0], (int)lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (++kc == kchunks) { kc = 0; ++r; }
    if (t + 1 < total) issue(a_blk + (long long)r * brs_a + kc * kstepA, b_blk + (long long)r * brs_b + kc * kstepB, buf ^ 1);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[nt][s2], af[mt][s2], acc[mt][nt], 0, 0, 0);
  }
  static_for<4>([&](auto idx) { constexpr int mt = idx.value / 2, nt = idx.value % 2; tile_store<true, true>(acc[mt][nt], p, q[mt][nt], tc[mt][nt]); });
}

// ------------------------------------------------------------------------------------------------
// The blocked kernel for 16 x 16 x K tiles (K a multiple of 16, plain epilogue): same workgroup structure as gemm_f32_blocked_kernel -- a
// 128 x 128 macro tile (8 x 8 problems), eight 32 x 32 x 32 operand blocks per step by LDS-DMA into a double buffer, wave = 64 x 64 on the
// 32x32x2 MFMA -- only the addressing differs: a 32 x 32 operand block is 2 x 2 SUB-blocks, (problem 2b / 2b+1) x (16-deep sub-step 2t / 2t+1
// of the flattened (batch-reduce element, 16-deep K chunk) sequence), and a 32 x 32 accumulator tile is 2 x 2 problems of C.
//   A image [32 k][32 i]: DMA instruction x covers k = 8x .. 8x+7 -> sub-step x >> 1 is uniform per instruction; lanes pick the problem.
//   B image [32 j][8 x 16 B] swizzled: a lane's 16-byte piece belongs to sub-step (source chunk) >> 2 -> selected per lane.
// One tile per wave (gemm_p16_kernel / t16) reaches 37 % of the f32 matrix peak on this regime; this form is the one that is matrix-bound.
// ------------------------------------------------------------------------------------------------
#if !defined(XAMD_GEMM_SHARD)      // a plain (non-template) kernel: emitted by the main translation unit only
__global__ __launch_bounds__(256) void gemm_f32_blocked16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds_all[2][8][1024];
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int ni = p.batch_inner, MI = ni / 8u;
  unsigned int g = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) g = (g & 7u) * (gridDim.x >> 3) + (g >> 3);
  const unsigned int mj = g / MI, mi = g - mj * MI;
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const long long brs_a = p.br_mode == 3 ? p.br_stride_a : 0, brs_b = p.br_mode == 3 ? p.br_stride_b : 0;
  const unsigned int kch = (unsigned int)p.k >> 4;                         // 16-deep sub-steps per batch-reduce element
  const unsigned int total = ((unsigned int)p.br_count * kch) >> 1;        // 32-deep steps (launch_gemm guarantees an even sub-step count)
  // --- DMA duty: A block w = problems pa0, pa0 + 1 (rows 0..15 / 16..31 of the block); B block w = problems pb0, pb0 + 1 (columns)
  const unsigned int pa0 = mi * 8u + 2u * w, pb0 = mj * 8u + 2u * w;
  gcptr a_base = (gcptr)p.a + (long long)pa0 * p.bs_a, b_base = (gcptr)p.b + (long long)pb0 * p.bs_b;
  long long offA[4], offB[4]; unsigned int subB[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const unsigned int L = lane + 64u * x, hi = L >> 3, lo = L & 7u;
    // A piece: image row k = hi (sub-step hi >> 4 == x >> 1), 16-byte column group lo: problem lo >> 2, rows 4 (lo & 3) ..
    offA[x] = (long long)(lo >> 2) * p.bs_a + (long long)(((hi & 15u) * lda + (lo & 3u) * 4u) * 4u);
    // B piece: image column j = hi (problem hi >> 4, local column hi & 15), slot lo holds source chunk cs = lo ^ swizzle: k = 4 cs .. 4 cs + 3
    const unsigned int cs = lo ^ ((hi >> 1) & 7u);
    subB[x] = cs >> 2;
    offB[x] = (long long)(hi >> 4) * p.bs_b + (long long)(((hi & 15u) * ldb + (cs & 3u) * 4u) * 4u);
  }
  auto issue = [&](unsigned int t, int buf) {
    // the two 16-deep sub-steps of step t: flattened index 2t and 2t + 1 -> (batch-reduce element, K chunk)
    long long sa[2], sb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned int q = 2u * t + u, r = q / kch, kc = q - r * kch;
      sa[u] = (long long)r * brs_a + (long long)kc * 64ll * lda;           // 16 k columns of A = 16 * lda floats
      sb[u] = (long long)r * brs_b + (long long)kc * 64ll;                 // 16 k rows of B = 64 bytes down every column
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(a_base + sa[x >> 1] + offA[x]), (lds_vptr)((char*)&lds_all[buf][w][0] + 1024 * x), 16, 0, 0);
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(b_base + (subB[x] ? sb[1] : sb[0]) + offB[x]), (lds_vptr)((char*)&lds_all[buf][4 + w][0] + 1024 * x), 16, 0, 0);
  };
  const unsigned int wi = w & 1u, wj = w >> 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
  if (total != 0) issue(0, 0);
  for (unsigned int t = 0; t < total; ++t) {
    const int buf = (int)(t & 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float af[2][16], bf[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) frag_read<false>(af[mt], &lds_all[buf][2 * wi + mt][0], (int)lane);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) frag_read<true>(bf[nt], &lds_all[buf][4 + 2 * wj + nt][0], (int)lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 1 < total) issue(t + 1, buf ^ 1);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[nt][s2], af[mt][s2], acc[mt][nt], 0, 0, 0);
  }
  // C: accumulator tile (mt, nt) covers problems (bi0 + (li >> 4), bj0 + (j >> 4)); lane = row li, register r = column jl_of(r, h)
  const unsigned int ldc = (unsigned int)p.ldc;
  static_for<4>([&](auto idx) {
    constexpr int mt = idx.value / 2, nt = idx.value % 2;
    const unsigned int bi = mi * 8u + 2u * (2u * wi + mt) + (li >> 4), bj0 = mj * 8u + 2u * (2u * wj + nt);
    gptr crow = (gptr)p.c + (long long)bi * p.bs_c + (long long)((li & 15u) * 4u);
    static_for<16>([&](auto rc) {
      constexpr int r = rc.value;
      const unsigned int j = (unsigned int)jl_of(r, (int)h);                 // 0..31 inside the tile
      st_stream((GM float*)(crow + (long long)(bj0 + (j >> 4)) * p.bs_c2 + (long long)((j & 15u) * ldc) * 4ll), acc[mt][nt][r]);
    });
  });
}
#endif

// ------------------------------------------------------------------------------------------------
// f32 64 x 64 x K problems, one problem per WORKGROUP (streaming batches).  gemm_f32_dma_kernel<2,2> gives a wave the whole 64 x 64 tile:
// 16 KiB of wave-private LDS, 8 waves per CU, 128 dependent-latency-bound MFMAs per problem behind every load round trip -- 0.51 of the
// HBM roofline at batch 4096.  Here the four waves of a workgroup share the problem: per 32-deep K step the workgroup brings in two A
// blocks and two B blocks (16 KiB) once by LDS-DMA, wave w fetching block w, and wave (wi, wj) multiplies A block wi with B block wj
// (16 MFMAs): a quarter of the matrix work per wave, 20 waves per CU, and BOTH steps of a k = 64 problem in flight before the first
// MFMA (two LDS images; longer chains refill an image two steps ahead, behind a second barrier).
// NN, exact, 16-byte aligned operands, plain or STRIDE batch-reduce; any batch form and any epilogue (batch_ptrs / tile_init / tile_store).
// ------------------------------------------------------------------------------------------------
template <int AUX>
__global__ __launch_bounds__(256) void gemm_f32_wg64_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds_all[2][4][1024];
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int bidx = logical_block(p);
  const BatchPtrs q = batch_ptrs(p, bidx);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  // DMA duty of wave w: blocks 0, 1 = A rows 0..31 / 32..63 (image [k][i] linear), blocks 2, 3 = B columns 0..31 / 32..63 (image [j][8 x 16 B] swizzled)
  const bool isA = w < 2u;
  unsigned int off[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const unsigned int L = lane + 64u * x, hi = L >> 3, lo = L & 7u;
    off[x] = isA ? (hi * lda + lo * 4u) * 4u : (hi * ldb + ((lo ^ ((hi >> 1) & 7u)) * 4u)) * 4u;
  }
  const unsigned long long sub = isA ? 128ull * w : 128ull * ldb * (w - 2u), kstep = isA ? 128ull * lda : 128ull;
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  const unsigned long long total = p.br_count * kchunks;
  gcptr ar, br;
  auto issue = [&](unsigned long long t) {
    const unsigned int r = (unsigned int)t / kchunks, kc = (unsigned int)t - r * kchunks;          // 32-bit: launch_gemm checks the step count
    br_base(p, q, r, ar, br);
    gcptr src = (isA ? ar : br) + sub + kc * kstep;
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(src + off[x]), (lds_vptr)((char*)&lds_all[t & 1ull][w][0] + 1024 * x), 16, 0, AUX);
  };
  const unsigned int wi = w & 1u, wj = w >> 1;
  f32x16 acc;
  TileCtx tc; tc.i = (int)(32u * wi + li); tc.j0 = (int)(32u * wj); tc.h = (int)h; tc.ivalid = true;
  tile_init<true, true>(acc, p, q, tc);
  if (total > 0) issue(0);
  if (total > 1) issue(1);
  for (unsigned long long t = 0; t < total; ++t) {
    if (t + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // step t landed, step t + 1 may fly
    wg_barrier();
    float af[16], bf[16];
    frag_read<false>(af, &lds_all[t & 1ull][wi][0], (int)lane);
    frag_read<true>(bf, &lds_all[t & 1ull][2 + wj][0], (int)lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 2 < total) { wg_barrier(); issue(t + 2); }       // every wave has read image t & 1: refill it two steps ahead
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s2], af[s2], acc, 0, 0, 0);
  }
  tile_store<true, true>(acc, p, q, tc);
}

// ------------------------------------------------------------------------------------------------
// f32 "blob" kernel for the odd small shapes LIBXSMM is known for (23x23x23 ...): one masked 32x32 tile, m, n, k, lda, ldb
// <= 32, no transposes.  A_r (k*lda floats) and B_r (n*ldb floats) are contiguous blobs whatever the shape, so they are
// brought in as FLAT dword streams with LDS-DMA (global_load_lds_dword: 256 contiguous bytes per instruction, no VGPRs, lanes
// past the end switched off) instead of per-row / per-column masked loads; the MFMA fragments are then picked out of the
// LDS images with shape-aware indexing (out-of-range operands read as zero).  Natural k order: bitwise the k-ordered
// fmaf chain like the other f32 MFMA kernels.
// ------------------------------------------------------------------------------------------------
#if !defined(XAMD_GEMM_SHARD)      // a plain (non-template) kernel: emitted by the main translation unit only
__global__ __launch_bounds__(256) void gemm_f32_blob_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds_all[4][2048];
  const WaveJob job = wave_job(p, 32, 32);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  float* la = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  float* lb = la + 1024;
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  TileCtx tc; tc.i = li; tc.j0 = 0; tc.h = h; tc.ivalid = li < p.m;
  f32x16 acc;
  tile_init<false, true>(acc, p, q, tc);
  const int na = p.k * p.lda, nb = p.n * p.ldb;           // dwords per blob (<= 1024 each)
  const bool iv = li < p.m, jv = li < p.n;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
#pragma unroll 1
    for (int d0 = 0; d0 < na; d0 += 64)
      if (d0 + lane < na) __builtin_amdgcn_global_load_lds((GM const void*)(ar + 4ll * (d0 + lane)), (lds_vptr)(la + d0), 4, 0, 0);
#pragma unroll 1
    for (int d0 = 0; d0 < nb; d0 += 64)
      if (d0 + lane < nb) __builtin_amdgcn_global_load_lds((GM const void*)(br + 4ll * (d0 + lane)), (lds_vptr)(lb + d0), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float af[16], bf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = 2 * s + h;
      af[s] = (iv && k < p.k) ? la[li + k * p.lda] : 0.0f;
      bf[s] = (jv && k < p.k) ? lb[k + li * p.ldb] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  tile_store<false, true>(acc, p, q, tc);
}
#endif

// ------------------------------------------------------------------------------------------------
// f32 "ragged" kernel: the odd small shapes LIBXSMM exists for (23^3 is BASELINE config #1; 13^3, 40^3, 50^3, 72^3 ...), NN, plain
// epilogue (beta 0 or 1), any leading dimensions, every batch form.  ONE WORKGROUP OF W WAVES PER PROBLEM (W = 1 up to 32 x 32, else 4);
// a problem is a sequence of CHUNKS (batch-reduce block, k-range of at most kc), usually one.
// Such problems are a few KB each: what bounds the kernel is the number of instructions per byte, then the bytes in flight.
//   * NO PER-ELEMENT ARITHMETIC.  A thread's place in a block is fixed once: for A (and C) thread t takes row i = t % m of column t / m
//     of every ROUND of cpr = 64 W / m columns; for B it takes k = t % kc of column t / kc.  A round is then one buffer load "lane
//     constant + scalar round offset" (columns a leading dimension apart cost nothing) and one LDS write "running lane pointer"; the
//     resource ends with the block, so rounds that run past it (k tail, last columns) and idle lanes read 0 and their stores are
//     dropped by the bounds check: no predicates.  m = 23: 46 of 64 lanes busy, 184 contiguous bytes per instruction.
//   * The landing zone is the REGISTER FILE (R2 dwords per thread and operand): all loads of a chunk are in flight at once, and in a
//     problem of several chunks the next one is requested before the current one is multiplied and waits in registers.
//   * The product is formed on v_mfma_f32_16x16x4_f32 tiles, transposed (rows = j, columns = i) so that a lane holds C(i, 4 consecutive
//     j): ceil(m/16) x ceil(n/16) tiles instead of padding to 32 / 64.  A wave owns up to NP vertical tile PAIRS (two i-tiles of one
//     j-tile share the B fragment).  Lanes of rows i >= m / columns j >= n read a clamped address: what they compute never leaves the
//     registers (an output depends on its own row and column only); only the k tail is selected to zero.
//     k is consumed in natural order (MFMA s takes k = 4 s + lane group): bitwise the k-ordered fmaf chain.
//   * C leaves through an LDS image [j][16 ceil(m/16)] and the same round scheme (buffer stores).
// History, measured on 23^3 x 131072 (fraction of the 8 TB/s HBM peak; the bare access pattern copies at 0.74, tools/width_probe.hip):
// flat dword LDS-DMA with per-element div/mod 0.57 -> the same in a persistent, double-buffered loop 0.57 (the counters said VALU-bound:
// 50 % VALU + 27 % MFMA busy, waves waiting for an issue slot half of the time) -> lane-constant rounds, persistent 0.58 (more resident
// waves did not help, fewer hurt) -> lane-constant rounds, one short-lived workgroup per problem 0.73.
// Barriers (W = 4) are raw s_barrier: the fence of __syncthreads() would make the compiler drain loads in flight in front of it.
// ------------------------------------------------------------------------------------------------
struct RaggedCfg {
  unsigned int kc, kchunks;             // k-chunk depth held in LDS, chunks per K
  unsigned int cpr_a, cpr_b;            // columns per round: 64 W / m (A, C), 64 W / kc (B)
  unsigned int pb, pc;                  // LDS pitches: of a B column (k contiguous), of a C column (i contiguous); A columns are m apart
  unsigned int off_b, off_ci, off_co, spare;   // LDS dword offsets: B image, incoming C image (beta = 1), outgoing C image, 64 W spare dwords
  unsigned int tpi, npairs;             // tile pairs along i, tile pairs in total
};
// waves per SIMD the register allocator is held to (registers, not LDS, bound the bytes in flight): the stage is (2 or 3) R2 registers
constexpr int ragged_waves_per_simd(int W, int NP, int R2, bool BETA1, bool SINGLE) {
  if (SINGLE) return 1;                                            // the stage is dead once placed: the allocator needs no help
  const int regs = (BETA1 ? 3 : 2) * R2 + 8 * NP + 44 + ((W == 1 && NP == 2) ? 8 : 0) + (BETA1 ? 8 : 0);     // stage + accumulators + the rest, as measured: no scratch
  return regs <= 64 ? 8 : regs <= 72 ? 7 : regs <= 80 ? 6 : regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : 2;
}
// SINGLE: the problem is one chunk (K fits, one batch-reduce block): straight-line -- fetch, place, multiply, write.  Measured on the bare
// access pattern (tools/width_probe.hip): one short-lived wave per 23^2 block streams at 0.74 of the HBM peak, resident waves looping
// over the same blocks at 0.64-0.67 -- the hardware dispatcher sweeps memory in order and refills a CU the moment a wave retires.
// !SINGLE: the loop over the problem's chunks (written for a stride of problems per workgroup; launched with one problem each).
template <int W, int NP, int R2, bool BETA1, bool SINGLE>
__global__ __launch_bounds__(64 * W, ragged_waves_per_simd(W, NP, R2, BETA1, SINGLE)) void gemm_f32_ragged_kernel(GemmArgs p, RaggedCfg c) {
  extern __shared__ __attribute__((aligned(16))) float rg_lds[];
  const unsigned int tid = threadIdx.x, lane = tid & 63u;
  const unsigned int w = (W == 1) ? 0u : (unsigned int)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const unsigned int x = lane & 15u, g = lane >> 4;
  const unsigned int m = (unsigned int)p.m, n = (unsigned int)p.n, K = (unsigned int)p.k;
  const unsigned int nprob = p.nbatch, stride = gridDim.x;
  const unsigned long long nbr = p.br_count;
  const unsigned int lda4 = 4u * (unsigned int)p.lda, ldb4 = 4u * (unsigned int)p.ldb, ldc4 = 4u * (unsigned int)p.ldc;
  auto block_rsrc = [&](gcptr base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)uniform_u64((unsigned long long)(size_t)base), (short)0, (int)bytes, 0x00020000);
  };
  constexpr unsigned int kIdle = 0x7ffffff0u;                      // lane offset behind every block (round offsets stay below 2^31)
  // ---- a thread's place in a round, fixed for the life of the kernel
  const unsigned int sub_a = tid / m, i_a = tid - sub_a * m, sub_b = tid / c.kc, k_b = tid - sub_b * c.kc;
  const bool act_a = sub_a < c.cpr_a, act_b = sub_b < c.cpr_b;
  const unsigned int vo_a = act_a ? sub_a * lda4 + 4u * i_a : kIdle, vo_b = act_b ? sub_b * ldb4 + 4u * k_b : kIdle, vo_c = act_a ? sub_a * ldc4 + 4u * i_a : kIdle;
  // LDS (dword indices): where the thread's element of round 0 goes, and how far the next round is (idle lanes stay on a spare dword)
  const unsigned int spare = c.spare + tid;                        // idle lanes: a dword of their own behind the images
  const unsigned int ls_a = act_a ? sub_a * m + i_a : spare, st_a = act_a ? c.cpr_a * m : 0u;
  const unsigned int ls_b = act_b ? c.off_b + sub_b * c.pb + k_b : spare, st_b = act_b ? c.cpr_b * c.pb : 0u;
  const unsigned int ls_ci = act_a ? c.off_ci + sub_a * c.pc + i_a : spare, ls_co = act_a ? c.off_co + sub_a * c.pc + i_a : spare, st_c = act_a ? c.cpr_a * c.pc : 0u;
  if (nbr == 0) {                                                  // no blocks: C = beta * C
    if (!BETA1)
      for (unsigned int prob = blockIdx.x; prob < nprob; prob += stride) {
        const __amdgpu_buffer_rsrc_t RC = block_rsrc((gcptr)batch_ptrs(p, prob).c, (n - 1u) * ldc4 + 4u * m);
        for (unsigned int j0 = 0; j0 < n; j0 += c.cpr_a) __builtin_amdgcn_raw_buffer_store_b32(0u, RC, (int)vo_c, (int)(j0 * ldc4), 0);
      }
    return;
  }
  unsigned int prob = blockIdx.x;
  if (prob >= nprob) return;
  // (tile pair PI of this wave) -> lane coordinates; the pair index is wave-uniform
  auto pair_of = [&](int PI, unsigned int& tj, unsigned int& tp) -> bool {
    const unsigned int pair = w + (unsigned int)PI * W;
    if (pair >= c.npairs) return false;
    tj = (c.tpi == 1u) ? pair : pair / c.tpi; tp = pair - tj * c.tpi;
    return true;
  };
  // ---- the staged chunk: R2 rounds of A, R2 of B, (beta = 1) R2 of C, one dword per thread and round
  constexpr int RS = (BETA1 ? 3 : 2) * R2;
  unsigned int stage[RS];
  unsigned int st_kcur = 0; bool st_withc = false;
  auto fetch = [&](const BatchPtrs& q, unsigned long long r, unsigned int kci) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    const unsigned int kc0 = kci * c.kc, kcur = min(c.kc, K - kc0);
    const __amdgpu_buffer_rsrc_t RA = block_rsrc(ar + (unsigned long long)kc0 * lda4, (kcur - 1u) * lda4 + 4u * m);
    const __amdgpu_buffer_rsrc_t RB = block_rsrc(br + 4ull * kc0, (n - 1u) * ldb4 + 4u * kcur);
    static_for<R2>([&](auto IT) { stage[IT] = __builtin_amdgcn_raw_buffer_load_b32(RA, (int)vo_a, (int)((unsigned int)IT * c.cpr_a * lda4), 0); });
    static_for<R2>([&](auto IT) { stage[R2 + IT] = __builtin_amdgcn_raw_buffer_load_b32(RB, (int)vo_b, (int)((unsigned int)IT * c.cpr_b * ldb4), 0); });
    const bool withc = BETA1 && r == 0 && kci == 0;
    if constexpr (BETA1) {
      if (withc) {
        const __amdgpu_buffer_rsrc_t RC = block_rsrc((gcptr)q.c, (n - 1u) * ldc4 + 4u * m);
        static_for<R2>([&](auto IT) { stage[2 * R2 + IT] = __builtin_amdgcn_raw_buffer_load_b32(RC, (int)vo_c, (int)((unsigned int)IT * c.cpr_a * ldc4), 0); });
      }
    }
    st_kcur = kcur; st_withc = withc;
  };
  const unsigned int rounds_c = (n + c.cpr_a - 1u) / c.cpr_a;
  auto spill = [&]() {                                             // registers -> LDS images
    unsigned int* lds = (unsigned int*)rg_lds;
    unsigned int pa_ = ls_a, pb_ = ls_b;
    asm volatile("" : "+v"(pa_), "+v"(pb_));                       // the R2 addresses per image are loop invariant: recompute (one add each), do not park them in registers
    // every round is placed, also those behind an operand's last one (they hold zeros: a skip per round measured 2-5 % slower than the
    // LDS it saves is worth; the images are sized for R2 rounds)
    static_for<R2>([&](auto IT) { lds[pa_] = stage[IT]; pa_ += st_a; });
    static_for<R2>([&](auto IT) { lds[pb_] = stage[R2 + IT]; pb_ += st_b; });
    if constexpr (BETA1) {
      if (st_withc) { unsigned int pc_ = ls_ci; asm volatile("" : "+v"(pc_)); static_for<R2>([&](auto IT) { lds[pc_] = stage[2 * R2 + IT]; pc_ += st_c; }); }
    }
  };
  f32x4 acc[NP][2];
  float* const la = rg_lds; float* const lb = rg_lds + c.off_b; float* const lo = rg_lds + c.off_co;
  auto init_acc = [&]() {                                          // a new problem: accumulators start from 0 or from C
    const float* lc = rg_lds + c.off_ci;
    static_for<NP>([&](auto PI) {
      unsigned int tj, tp;
      acc[PI][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[PI][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (BETA1 && pair_of(PI, tj, tp)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned int i = 32u * tp + 16u * h + x;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const unsigned int j = 16u * tj + 4u * g + e; acc[PI][h][e] = (i < m && j < n) ? lc[j * c.pc + i] : 0.0f; }
        }
      }
    });
  };
  auto multiply = [&](unsigned int kcur) {                         // acc += A image x B image, kcur deep
    const unsigned int full = kcur >> 2;
    static_for<NP>([&](auto PI) {
      unsigned int tj, tp;
      if (pair_of(PI, tj, tp)) {
        const unsigned int i0 = 32u * tp + x, j = 16u * tj + x;
        const bool two = 32u * tp + 16u < m;                       // wave-uniform: the pair's second tile exists
        // rows i >= m and columns j >= n compute from a clamped (valid) address: their results stay in registers nobody stores
        const float* pa = la + g * m + (i0 < m ? i0 : 0u);
        const float* pa1 = la + g * m + (i0 + 16u < m ? i0 + 16u : 0u);
        const float* pbq = lb + (j < n ? j : 0u) * c.pb + g;
        const unsigned int m4 = 4u * m;
        f32x4 c0 = acc[PI][0], c1 = acc[PI][1];
        unsigned int s = 0;
        if (two) {
          for (; s + 2u <= full; s += 2u) {                        // two groups of four k per trip: six fragment reads, then four MFMAs
            const float a0 = pa[0], a1 = pa1[0], a2 = pa[m4], a3 = pa1[m4], b = pbq[0], b2 = pbq[4];
            pa += 2u * m4; pa1 += 2u * m4; pbq += 8;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a1, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, a2, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, a3, c1, 0, 0, 0);
          }
          if (s < full) {
            const float a0 = pa[0], a1 = pa1[0], b = pbq[0];
            pa += m4; pa1 += m4; pbq += 4;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a1, c1, 0, 0, 0);
          }
        } else {
          for (; s + 2u <= full; s += 2u) {
            const float a0 = pa[0], a2 = pa[m4], b = pbq[0], b2 = pbq[4];
            pa += 2u * m4; pbq += 8;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, a2, c0, 0, 0, 0);
          }
          if (s < full) {
            const float a0 = pa[0], b = pbq[0];
            pa += m4; pa1 += m4; pbq += 4;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, c0, 0, 0, 0);
          }
        }
        if (kcur & 3u) {                                           // the last, partial group of four k: both operands zero behind the depth
          const bool kv = 4u * full + g < kcur;
          const float a0 = kv ? pa[0] : 0.0f, a1 = (two && kv) ? pa1[0] : 0.0f, b = kv ? pbq[0] : 0.0f;
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, c0, 0, 0, 0);
          if (two) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a1, c1, 0, 0, 0);
        }
        acc[PI][0] = c0; acc[PI][1] = c1;
      }
    });
  };
  auto write_image = [&]() {                                       // registers -> C image [j][pc]
    static_for<NP>([&](auto PI) {
      unsigned int tj, tp;
      if (pair_of(PI, tj, tp)) {
        const unsigned int j0 = 16u * tj + 4u * g;
        if (j0 < n) {                                              // whole groups of four columns behind n are skipped; rows behind m land in the pitch
          float* dst = lo + j0 * c.pc + 32u * tp + x;
#pragma unroll
          for (int e = 0; e < 4; ++e) { dst[e * c.pc] = acc[PI][0][e]; if (32u * tp + 16u < m) dst[e * c.pc + 16u] = acc[PI][1][e]; }
        }
      }
    });
  };
  auto flush = [&](unsigned int which) {                           // C image -> memory, a round of columns per store
    const __amdgpu_buffer_rsrc_t RC = block_rsrc((gcptr)batch_ptrs(p, which).c, (n - 1u) * ldc4 + 4u * m);
    const unsigned int* src = (const unsigned int*)rg_lds;
    unsigned int pc_ = ls_co, so = 0;
    asm volatile("" : "+v"(pc_));
#pragma unroll 2
    for (unsigned int r = 0; r < rounds_c; ++r) {
      __builtin_amdgcn_raw_buffer_store_b32(src[pc_], RC, (int)vo_c, (int)so, 2);
      pc_ += st_c; so += c.cpr_a * ldc4;
    }
  };
  if constexpr (SINGLE) {                                          // one problem, one chunk: straight through (the C image takes the operands' place)
    fetch(batch_ptrs(p, prob), 0, 0);
    spill();
    if (W > 1) wg_barrier();
    init_acc();
    multiply(K);
    if (W > 1) wg_barrier();
    write_image();
    if (W > 1) wg_barrier();
    flush(prob);
    return;
  }
  // One loop trip = "place the staged chunk, request the one after it, multiply the placed one".  The first trip has nothing staged
  // yet and only requests (one fetch site, one spill site: the kernel is register-bound, code duplicated by inlining costs occupancy).
  // C of a finished problem is written one trip LATE, right before the next request: the wait for a staged chunk (vmcnt, in order) then
  // never has younger stores in front of it -- written right after the product, their acknowledgements would be waited for every trip.
  unsigned long long cur_r = 0, st_r = 0; unsigned int cur_k = 0, st_k = 0, st_prob = prob, out_prob = 0;
  bool placed = false, staged = false, more = true, out_pending = false;
  for (;;) {
    unsigned int kcur = 0;
    placed = staged;
    if (staged) {
      spill();                                                     // the chunk that waited in registers becomes the current one
      kcur = st_kcur; prob = st_prob; cur_r = st_r; cur_k = st_k;
      if (W > 1) wg_barrier();
      // the chunk after it
      if (++st_k == c.kchunks) { st_k = 0; if (++st_r == nbr) { st_r = 0; st_prob += stride; } }
      more = st_prob < nprob;
    }
    if (out_pending) { flush(out_prob); out_pending = false; }
    if (more) { const BatchPtrs qn = batch_ptrs(p, st_prob); fetch(qn, st_r, st_k); }
    staged = more;
    if (!placed) continue;
    if (cur_r == 0 && cur_k == 0) init_acc();
    multiply(kcur);
    if (cur_k + 1u == c.kchunks && cur_r + 1ull == nbr) { write_image(); out_pending = true; out_prob = prob; }     // complete: into its own LDS region
    if (!staged) break;
    if (W > 1) wg_barrier();                                       // everybody is done with the images before the next chunk replaces them
  }
  if (out_pending) { if (W > 1) wg_barrier(); flush(out_prob); }
}

// ------------------------------------------------------------------------------------------------
// f32, 16x16 tiles (v_mfma_f32_16x16x4_f32), NN layout, exact multiples of 16 only.
// lane = (x = lane&15, g = lane>>4).  Transposed product: operand 1 = B(k, j=x), operand 2 = A(i=x, k);
// result lane x = i, register q -> j = 4g + q.  k is consumed as 4g + s (not natural order).
// ------------------------------------------------------------------------------------------------
#if !defined(XAMD_GEMM_SHARD)      // a plain (non-template) kernel: emitted by the main translation unit only
__global__ __launch_bounds__(256) void gemm_mfma_f32_t16_kernel(GemmArgs p) {
  const WaveJob job = wave_job(p, 16, 16);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, x = lane & 15, g = lane >> 4;
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const int i = job.i0 + x;
  f32x4 acc;
  GM float* C = (GM float*)q.c;
  const float bias = p.colbias ? ((GM const float*)q.d)[i] : 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = job.j0 + 4 * g + e;
    const float start = beta0 ? 0.0f : C[(long long)j * p.ldc + i];
    acc[e] = p.colbias ? (beta0 ? bias : bias + start) : start;
  }
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    GM const float* A = (GM const float*)ar; GM const float* B = (GM const float*)br;
    for (int k0 = 0; k0 < p.k; k0 += 16) {
      float af[4]; f32x4 bv;
#pragma unroll
      for (int s = 0; s < 4; ++s) af[s] = A[i + (long long)(k0 + 4 * g + s) * p.lda];
      GM const float* bcol = B + (long long)(job.j0 + x) * p.ldb + k0 + 4 * g;
      if (((((unsigned long long)(size_t)B) & 15ull) == 0ull) && ((p.ldb & 3) == 0)) bv = *(GM const f32x4*)bcol;
      else { bv[0] = bcol[0]; bv[1] = bcol[1]; bv[2] = bcol[2]; bv[3] = bcol[3]; }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[s], af[s], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = job.j0 + 4 * g + e;
    const float xv = acc[e];
    if (p.act == 2 && q.mask) {
      const unsigned long long pos = __ballot(!(xv <= 0.0f));
      if ((lane & 7) == 0) q.mask[i / 8 + (long long)j * ((((p.ldc + 15) / 16) * 16) / 8)] = (unsigned char)((pos >> lane) & 0xffu);
    }
    C[(long long)j * p.ldc + i] = act_apply(p.act, xv);
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// 16 x 16 problems (m = n = 16, k a multiple of 16), plain epilogue, PPW problems per wave.  A 16^3 problem is 1.5 KiB (bf16) or 3 KiB
// (f32): with one problem per wave a launch is bound by the per-wave fixed cost (launch slot, address set-up, one load round trip for
// five loads).  Here a wave owns PPW consecutive batch elements, issues ALL their operand loads before the first MFMA and keeps one
// accumulator per problem.  Operand roles: A rows -> MFMA rows, B columns -> MFMA columns, so a lane ends up with four consecutive i of
// one C column j and stores them as ONE 8- or 16-byte access.
//   f32 : v_mfma_f32_16x16x4_f32, lane (x = lane & 15, g = lane >> 4) supplies A(i = x, k = k0 + 4 g + s) and B(k0 + 4 g + s, j = x);
//         k is consumed group-interleaved inside a 16-deep chunk (as in gemm_mfma_f32_t16_kernel), a valid summation order.
//   bf16: v_mfma_f32_16x16x16_bf16, A in VNNI-2 (two dwords = k 4 g .. 4 g + 3 of row x), B flat (8 contiguous bytes of column x).
// ------------------------------------------------------------------------------------------------
typedef short bf16x4 __attribute__((ext_vector_type(4)));

template <int PPW, bool BF16>
__global__ __launch_bounds__(256) void gemm_p16_kernel(GemmArgs p) {
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int first = (logical_block(p) * 4u + wave) * PPW;
  if (first >= p.nbatch) return;
  const unsigned int lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
  f32x4 acc[PPW];
  BatchPtrs q[PPW];
#pragma unroll
  for (int pp = 0; pp < PPW; ++pp) { acc[pp] = (f32x4)0.0f; q[pp] = batch_ptrs(p, first + pp < p.nbatch ? first + pp : first); }
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const int kchunks = p.k >> 4;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar[PPW], br[PPW];
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) br_base(p, q[pp], r, ar[pp], br[pp]);
    for (int kc = 0; kc < kchunks; ++kc) {
      if (BF16) {
        u32x2 av[PPW], bv[PPW];
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
          GM const unsigned int* A2 = (GM const unsigned int*)ar[pp] + (unsigned long long)(8u * kc + 2u * g) * lda + x;      // dword (k-pair, row)
          av[pp][0] = A2[0]; av[pp][1] = A2[lda];
          bv[pp] = *(GM const u32x2*)((GM const unsigned short*)br[pp] + (unsigned long long)x * ldb + 16u * kc + 4u * g);
        }
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
          acc[pp] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4, av[pp]), __builtin_bit_cast(bf16x4, bv[pp]), acc[pp], 0, 0, 0);
      } else {
        float af[PPW][4]; f32x4 bv[PPW];
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
          GM const float* A = (GM const float*)ar[pp] + (unsigned long long)(16u * kc + 4u * g) * lda + x;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) af[pp][s2] = A[(unsigned long long)s2 * lda];
          bv[pp] = *(GM const f32x4*)((GM const float*)br[pp] + (unsigned long long)x * ldb + 16u * kc + 4u * g);
        }
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) acc[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pp][s2], bv[pp][s2], acc[pp], 0, 0, 0);
      }
    }
  }
  const bool c_f32 = !BF16 || p.c_type == LIBXSMM_DATATYPE_F32;
#pragma unroll
  for (int pp = 0; pp < PPW; ++pp) {
    if (first + pp < p.nbatch) {
      if (c_f32) st_stream((GM f32x4*)((GM float*)q[pp].c + (unsigned long long)x * (unsigned int)p.ldc + 4u * g), acc[pp]);
      else { u32x2 v; v[0] = bf16_pk_exact(acc[pp][0], acc[pp][1]); v[1] = bf16_pk_exact(acc[pp][2], acc[pp][3]);
             st_stream((GM u32x2*)((GM unsigned short*)q[pp].c + (unsigned long long)x * (unsigned int)p.ldc + 4u * g), v); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 x bf16 -> fp32 accumulate, 32x32x16 MFMA; A in VNNI-2 layout [K/2][lda][2], B flat [n][ldb].
// Per K-chunk of 32 (two MFMA steps) lane (f = lane&31, h = lane>>5) owns k = k0 + 16h + 8s + e:
//   A: four dwords (k-pairs) per step, lanes contiguous along i  -> coalesced 128-byte rows
//   B: sixteen contiguous bytes per step (32 per chunk) at column j
// ------------------------------------------------------------------------------------------------
// F16 = true: the same kernel on IEEE halves (v_mfma_f32_32x32x16_f16) with the reference's F16 rule for the start value: beta * C (+ the fused column bias) is rounded
// to f16 on the way in [ref: gemm ref :2025-2124, :296-317]; the reference adds it after the sum, the matrix core before -- a difference of the f32 summation order only
// (round 6: any epilogue; until then the accumulators started at 0 and beta * C was added element by element behind the loop).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// BND (with BL; prepared in round 4, measured and adopted in round 5 for plain results -- ragged16_bounded): the
// operand descriptors carry the exact extent of the block, so a request beyond it (a row of the last k pair beyond m, a k pair beyond k in the last column) is dropped
// by the address unit instead of being clamped in registers -- every lane offset is loop invariant (two registers for B, eight for A, the rest scalar), which is what
// the fourth wave per SIMD of the 64 x 64 form needs (see decision 32: these kernels are short of waves).  Rows beyond m and columns beyond n read whatever lies
// inside the block: they only feed results nobody stores.
template <int MT, int NT, bool EXACT, bool F16 = false, bool BL = false, bool BND = false>      // BL (ragged shapes, B on dwords: launch_gemm): B through LDS
__global__ __launch_bounds__(256, BND ? 4 : 1) void gemm_mfma_bf16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds_img[4][BL ? (BND ? (NT + MT) * 2048 : NT * 2048) : 16];       // ragged shapes: the wave's B chunk ([32 NT columns][32 k] halves, 16-byte slots swizzled); BND: A's too
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h;
      tc[mt][nt].ivalid = tc[mt][nt].i < p.m;
      if (BND) {                                                // (BND: beta = 0, no bias -- launch_gemm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
      }
      else tile_init<EXACT, false, false, false, F16>(acc[mt][nt], p, q, tc[mt][nt]);       // (halves: the start value rounded to f16, see tile_init)
    }
  const int kchunks = (p.k + 31) / 32;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    GM const unsigned int* A2 = (GM const unsigned int*)ar;    // one dword = (k even, k odd) of one row
    GM const unsigned short* B = (GM const unsigned short*)br;
    const bool bvec = ((((unsigned long long)(size_t)B) & 15ull) == 0ull) && ((p.ldb & 7) == 0);
    // ragged shapes (round 4): every load of a chunk is issued UNCONDITIONALLY at its neighbour's address -- a lane beyond m / n reads the last real row / column, a k pair
    // beyond k the last real pair: the same lines its neighbours request -- and the padding is a select afterwards; written with a condition per load the compiler
    // waited for each load before it issued the next (40^3: 0.15 of the HBM roofline).  B as dwords (k pairs) when its columns start on dwords, else as halves.
    const bool bdw = !EXACT && ((((unsigned long long)(size_t)B) & 3ull) == 0ull) && ((p.ldb & 1) == 0);
    if constexpr (BL && BND) {
      // BOTH operand panels of a chunk by LDS-DMA, a dword per lane: nothing in flight occupies a register.  A: instruction x fetches k pair x of the chunk for the wave's
      // 32 MT rows (lanes beyond them idle) into [16 k pairs][32 MT rows] dwords, read back with ds_read_b32 down a column (lanes along the row: conflict free).
      char* image = lds_img[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
      char* image_a = image + NT * 2048;
      const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
      const int kpl = (p.k >> 1) - 1;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)uniform_u64((unsigned long long)(size_t)ar), (short)0, (int)(((unsigned int)kpl * lda + (unsigned int)p.m) * 4u), 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)uniform_u64((unsigned long long)(size_t)br), (short)0, (int)(((unsigned int)(p.n - 1) * ldb + (unsigned int)p.k) * 2u), 0x00020000);
      const unsigned int d = (unsigned int)lane & 15u, fb = (unsigned int)lane >> 4;
      // LDS slot lane + 64 x holds column f = fb + 4 x, 16-byte piece pc = (d >> 2) ^ ((f >> 1) & 3) of its 64 bytes; (f >> 1) & 3 = (fb >> 1) + 2 (x & 1): two lane offsets
      unsigned int voffB[2];
      voffB[0] = fb * ldb * 2u + (((d >> 2) ^ (fb >> 1)) * 16u) + (d & 3u) * 4u;
      voffB[1] = fb * ldb * 2u + (((d >> 2) ^ (fb >> 1) ^ 2u) * 16u) + (d & 3u) * 4u;
      // A: MT = 2: lane = row of the wave's 64, one k pair per instruction; MT = 1: lane = (k pair parity, row of 32), two k pairs per instruction
      const unsigned int voffA = MT == 2 ? (unsigned int)lane * 4u : ((unsigned int)h * lda + (unsigned int)li) * 4u;
      for (int kc = 0; kc < kchunks; ++kc) {
        // (the scalar offsets are running sums over steps the compiler cannot see through: hoisted out of the loop, the 16 + 16 of them overflowed the scalar file)
        unsigned int so_b = (unsigned int)job.j0 * ldb * 2u + 64u * (unsigned int)kc, step_b = ldb * 8u;
        unsigned int so_a = 16u * (unsigned int)kc * lda * 4u + 4u * (unsigned int)job.i0, step_a = (MT == 2 ? 1u : 2u) * lda * 4u;
        asm volatile("" : "+s"(step_b), "+s"(step_a));
#pragma unroll
        for (int x = 0; x < NT * 8; ++x) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(image + 256 * x), 4, (int)voffB[x & 1], (int)so_b, 0, 0); so_b += step_b; }
#pragma unroll
        for (int x = 0; x < MT * 8; ++x) {            // LDS slot lane + 64 x: k pair x (MT = 2) / 2 x + h (MT = 1) of the chunk, row lane (li)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)(image_a + 256 * x), 4, (int)voffA, (int)so_a, 0, 0); so_a += step_a; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool tail = 16 * kc + 15 > kpl;                       // the chunk that holds k pairs beyond k (wave-uniform): zero on BOTH sides
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          u32x4 af[MT], bfr[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { const int f = 32 * nt + li; bfr[nt] = *(const u32x4*)(image + f * 64 + (((2 * h + s) ^ ((f >> 1) & 3)) * 16)); }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) af[mt][e] = *(const unsigned int*)(image_a + ((8 * h + 4 * s + e) * (32 * MT) + 32 * mt + li) * 4);
          if (tail) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool kok = kc * 16 + 8 * h + 4 * s + e <= kpl;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) af[mt][e] = kok ? af[mt][e] : 0u;
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) bfr[nt][e] = kok ? bfr[nt][e] : 0u;
            }
          }
          static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
            acc[mt][nt] = mfma_16bit<F16>(bfr[nt], af[mt], acc[mt][nt]); });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      continue;
    }
    if constexpr (BL) {
      // B through LDS: a lane's MFMA operand is eight k of ONE column, i.e. lanes a whole column apart in memory -- fetched into registers every load instruction
      // touched 32+ cache lines for 4 bytes each.  Here the chunk's B panel is brought in by LDS-DMA a dword per lane (16 lanes = the 64 bytes of one column's chunk,
      // 4 columns per instruction, no registers, any ldb), the 16-byte slots XOR-swizzled on the source side as in gemm_bf16_stream_kernel, and read back as one
      // ds_read_b128 per operand.  A (rows contiguous) stays in registers, buffer-addressed (32-bit offsets: the 64-bit addresses of 32 loads held the kernel at 2 waves).
      char* image = lds_img[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
      const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar), rb = wave_rsrc(br);
      const int kpl = (p.k >> 1) - 1;                           // last real k pair (k is even with a VNNI-2 A)
      const unsigned int d = (unsigned int)lane & 15u, fb = (unsigned int)lane >> 4;
      unsigned int icol[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { const int i = job.i0 + 32 * mt + li; icol[mt] = 4u * (unsigned int)(i < p.m ? i : p.m - 1); }
      // (measured and not adopted: TWO chunks requested together -- second image, second set of A registers, a counted wait -- as gemm_bf16_stream_kernel does: the
      //  registers and LDS cost a wave per SIMD and every shape lost, 40^3 0.49 -> 0.40, 24^3 0.60 -> 0.48; the masked 8-bit kernel likewise, u8 x i8 0.52 -> 0.49:
      //  what these kernels are short of is waves, not requests per wave: profiles/r04b_m8_lds.jsonl, tag "two".  Nor can the fourth wave of the 64 x 64 form be
      //  had by a register bound: 128 registers spill 13 of them, and the scratch set-up alone costs 0.49 -> 0.34, tag "lb4"; 32 x 32 tiles for 40^3 problems --
      //  four light waves per problem, operands read twice -- 0.44 against 0.55 on the 8-bit kernel, tag "lb4_tile1")
      for (int kc = 0; kc < kchunks; ++kc) {
        u32x4 af[MT][2], bfr[NT][2];
#pragma unroll
        for (int x = 0; x < NT * 8; ++x) {
          const unsigned int f = fb + 4u * x, pc = (d >> 2) ^ ((f >> 1) & 3u);          // LDS slot lane + 64 x: column f, 16-byte piece pc of its 64 bytes
          const int kp = kc * 16 + (int)(pc * 4u + (d & 3u)), jc = job.j0 + (int)f;
          const unsigned int voff = (unsigned int)(jc < p.n ? jc : p.n - 1) * (unsigned int)p.ldb * 2u + 4u * (unsigned int)(kp < kpl ? kp : kpl);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(image + 256 * x), 4, (int)voff, 0, 0, 0);
        }
        bool kok[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kp = kc * 16 + 8 * h + 4 * s + e;
            kok[s][e] = kp <= kpl;
            const unsigned int krow = (unsigned int)(kok[s][e] ? kp : kpl) * (unsigned int)p.lda * 4u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt][s][e] = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(krow + icol[mt]), 0, 0);
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int f = 32 * nt + li;
            bfr[nt][s] = *(const u32x4*)(image + f * 64 + (((2 * h + s) ^ ((f >> 1) & 3)) * 16));
          }
        // k beyond the problem: zero on BOTH sides (the clamped loads brought real numbers); rows beyond m / columns beyond n only feed results nobody stores
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt][s][e] = kok[s][e] ? af[mt][s][e] : 0u;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt][s][e] = kok[s][e] ? bfr[nt][s][e] : 0u;
          }
#pragma unroll
        for (int s = 0; s < 2; ++s)
          static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
            acc[mt][nt] = mfma_16bit<F16>(bfr[nt][s], af[mt][s], acc[mt][nt]); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // LDS reads retired before the image is refilled
      }
      continue;
    }
    for (int kc = 0; kc < kchunks; ++kc) {
      const int k0 = kc * 32;
      u32x4 af[MT][2], bfr[NT][2];
      if (!EXACT && bdw) {
        const int kpl = (p.k >> 1) - 1;                         // last real k pair (k is even with a VNNI-2 A)
        bool kok[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kp = (k0 + 16 * h + 8 * s) / 2 + e;
            kok[s][e] = kp <= kpl;
            const long long kps = kok[s][e] ? kp : kpl;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { const int i = job.i0 + 32 * mt + li; af[mt][s][e] = A2[kps * p.lda + (i < p.m ? i : p.m - 1)]; }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { const int j = job.j0 + 32 * nt + li; bfr[nt][s][e] = ((GM const unsigned int*)(B + (long long)(j < p.n ? j : p.n - 1) * p.ldb))[kps]; }
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt][s][e] = (kok[s][e] && job.i0 + 32 * mt + li < p.m) ? af[mt][s][e] : 0u;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt][s][e] = (kok[s][e] && job.j0 + 32 * nt + li < p.n) ? bfr[nt][s][e] : 0u;
          }
      } else
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kb = k0 + 16 * h + 8 * s;                    // first k of this lane's 8
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int i = job.i0 + 32 * mt + li;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kp = kb / 2 + e;                          // k-pair index
            af[mt][s][e] = (EXACT || (i < p.m && 2 * kp < p.k)) ? A2[(long long)kp * p.lda + i] : 0u;
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = job.j0 + 32 * nt + li;
          GM const unsigned short* col = B + (long long)j * p.ldb + kb;
          if (EXACT && bvec) {
            bfr[nt][s] = *(GM const u32x4*)col;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned int lo = (EXACT || (j < p.n && kb + 2 * e < p.k)) ? col[2 * e] : 0u;
              const unsigned int hi = (EXACT || (j < p.n && kb + 2 * e + 1 < p.k)) ? col[2 * e + 1] : 0u;
              bfr[nt][s][e] = lo | (hi << 16);
            }
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bfr[nt][s]), __builtin_bit_cast(f16x8, af[mt][s]), acc[mt][nt], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[nt][s]), __builtin_bit_cast(bf16x8, af[mt][s]), acc[mt][nt], 0, 0, 0);
    }
  }
  if constexpr (BND) {
    const unsigned int esz = p.c_type == LIBXSMM_DATATYPE_F32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)uniform_u64((unsigned long long)(size_t)q.c), (short)0, (int)(((unsigned int)(p.n - 1) * (unsigned int)p.ldc + (unsigned int)p.m) * esz), 0x00020000);
    const bool c_dword = (uniform_u64((unsigned long long)(size_t)q.c) & 3ull) == 0ull;
    static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store_buf(acc[mt][nt], p, rc, c_dword, tc[mt][nt]); });
    return;
  }
  // (ragged tiles: plain stores -- their columns are pieces of cache lines, which non-temporal stores would send to memory one by one)
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<EXACT, false, EXACT>(acc[mt][nt], p, q, tc[mt][nt]); });
}

// ------------------------------------------------------------------------------------------------
// 8-bit WEIGHTS x bf16 activations on the bf16 matrix cores (round 4): BF8 / HF8 weights (A in VNNI-2 byte pairs or flat) [ref: gemm ref :2171-2366] and int8 weights with
// one f32 scale per row (flat A, the scaled weight rounded to bf16) [ref: gemm ref :1684-1730].  These ran on the one-element-per-thread kernel.  The weights become the
// bf16 values the reference multiplies with -- an E5M2 / E4M3 number IS a bf16 number (v_cvt_pk_f32_bf8 / _fp8, the upper halves of the two results), the scaled int8
// weight is rounded with v_cvt_pk_bf16_f32 -- in registers, on the way into the masked bf16 kernel's operand layout (any shape, any leading dimension, any batch-reduce
// form); sums in the matrix core's order, epilogues as the bf16 kernels.
// KIND 0 / 1: BF8 / HF8 in VNNI-2 pairs, 2 / 3: BF8 / HF8 flat, 4: scaled int8 flat.
// ------------------------------------------------------------------------------------------------
// EXACT: whole 32 x 32 x 32 tiles and 16-byte aligned B columns (the host checks): no masks, a step's B operand is ONE 16-byte load.
template <int MT, int NT, int KIND, bool EXACT, bool BL = false>          // BL (ragged shapes, B on dwords, k even: launch_gemm): B through LDS as gemm_mfma_bf16_kernel
__global__ __launch_bounds__(256) void gemm_w8_bf16_kernel(GemmArgs p) {
  constexpr bool PAIRS = KIND < 2;
  __shared__ __attribute__((aligned(16))) char lds_img[4][BL ? NT * 2048 : 16];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
  float scf[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int i = job.i0 + 32 * mt + li;
    scf[mt] = (KIND == 4 && (EXACT || i < p.m)) ? ((GM const float*)(p.a_scf + (long long)job.bidx * p.bs_scf))[i] : 1.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      tc[mt][nt].i = i; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h; tc[mt][nt].ivalid = EXACT || i < p.m;
      tile_init<EXACT, false>(acc[mt][nt], p, q, tc[mt][nt]);
    }
  }
  const int kchunks = (p.k + 31) / 32;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    GM const unsigned char* A8 = (GM const unsigned char*)ar;
    GM const unsigned short* B = (GM const unsigned short*)br;
    for (int kc = 0; kc < kchunks; ++kc) {
      const int k0 = kc * 32;
      u32x4 af[MT][2], bfr[NT][2];
      unsigned int raw[MT][2][4][2];                           // every load of the chunk is issued before the first conversion (written as one loop the compiler waited after each)
      if constexpr (!EXACT) {
        // ragged shapes: as gemm_mfma_bf16_kernel -- every load unconditional at the neighbour's address (last real row / column / k), the padding a select afterwards
        const int kl = p.k - 1;
        unsigned int blo[BL ? 1 : NT][2][4], bhi[BL ? 1 : NT][2][4];
        char* image = lds_img[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
        if constexpr (BL) {
          const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br);
          const unsigned int d = (unsigned int)lane & 15u, fb = (unsigned int)lane >> 4;
          const int kpl = (p.k >> 1) - 1;
#pragma unroll
          for (int x = 0; x < NT * 8; ++x) {
            const unsigned int f = fb + 4u * x, pc = (d >> 2) ^ ((f >> 1) & 3u);          // LDS slot lane + 64 x: column f, 16-byte piece pc of its 64 bytes
            const int kp = kc * 16 + (int)(pc * 4u + (d & 3u)), jc = job.j0 + (int)f;
            const unsigned int voff = (unsigned int)(jc < p.n ? jc : p.n - 1) * (unsigned int)p.ldb * 2u + 4u * (unsigned int)(kp < kpl ? kp : kpl);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(image + 256 * x), 4, (int)voff, 0, 0, 0);
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ke = k0 + 16 * h + 8 * s + 2 * e;
            const long long k_lo = ke < kl ? ke : kl, k_hi = ke + 1 < kl ? ke + 1 : kl;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const int i = job.i0 + 32 * mt + li, ic = i < p.m ? i : p.m - 1;
              if constexpr (PAIRS) { raw[mt][s][e][0] = *(GM const unsigned short*)(A8 + (((ke < kl ? ke : kl - 1) >> 1) * (long long)p.lda + ic) * 2); raw[mt][s][e][1] = 0u; }
              else { raw[mt][s][e][0] = A8[k_lo * p.lda + ic]; raw[mt][s][e][1] = A8[k_hi * p.lda + ic]; }
            }
            if constexpr (!BL) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int j = job.j0 + 32 * nt + li;
              GM const unsigned short* col = B + (long long)(j < p.n ? j : p.n - 1) * p.ldb;
              blo[nt][s][e] = col[k_lo]; bhi[nt][s][e] = col[k_hi];
            }
            }
          }
        if constexpr (BL) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) { const int f = 32 * nt + li; bfr[nt][s] = *(const u32x4*)(image + f * 64 + (((2 * h + s) ^ ((f >> 1) & 3)) * 16)); }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // (the image is refilled by the next chunk's requests)
        } else asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ke = k0 + 16 * h + 8 * s + 2 * e;
            const bool k0ok = ke < p.k, k1ok = ke + 1 < p.k;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const bool iok = job.i0 + 32 * mt + li < p.m;
              raw[mt][s][e][0] = (iok && k0ok) ? raw[mt][s][e][0] : 0u; raw[mt][s][e][1] = (iok && k1ok) ? raw[mt][s][e][1] : 0u;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const bool jok = job.j0 + 32 * nt + li < p.n;
              if constexpr (BL) bfr[nt][s][e] = k0ok ? bfr[nt][s][e] : 0u;          // (k even: a pair is whole)
              else bfr[nt][s][e] = ((jok && k0ok) ? blo[nt][s][e] : 0u) | (((jok && k1ok) ? bhi[nt][s][e] : 0u) << 16);
            }
          }
      } else {
      char* ximage = lds_img[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
      if constexpr (EXACT && BL) {
        // (prepared in round 4, adopted in round 5) whole tiles: B as gemm_bf16_stream_kernel fetches it -- 16-byte LDS-DMA requests, four
        // lanes = the 64 bytes of one column's chunk, slots swizzled on the source side -- instead of 16 bytes per lane from 64 different columns
        const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br + 2ull * (unsigned long long)job.j0 * (unsigned long long)p.ldb);
#pragma unroll
        for (int x = 0; x < NT * 2; ++x) {
          const unsigned int L = (unsigned int)lane + 64u * x, f = L >> 2, pc = (L & 3u) ^ ((f >> 1) & 3u);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(ximage + 1024 * x), 16, (int)((f * (unsigned int)p.ldb) * 2u + pc * 16u), 64 * kc, 0, 0);
        }
        asm volatile("" ::: "memory");                          // (B's requests first: moved behind A's loads the compiler put a counted wait after each of them)
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kb = k0 + 16 * h + 8 * s;                    // first k of this lane's 8
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int i = job.i0 + 32 * mt + li;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ke = kb + 2 * e;                          // even k of the pair
            raw[mt][s][e][0] = raw[mt][s][e][1] = 0u;
            if (EXACT || (i < p.m && ke < p.k)) {
              if constexpr (PAIRS) raw[mt][s][e][0] = *(GM const unsigned short*)(A8 + ((long long)(ke >> 1) * p.lda + i) * 2);       // (k is even for VNNI-2 A: a pair is whole)
              else {
                raw[mt][s][e][0] = A8[(long long)ke * p.lda + i];
                if (EXACT || ke + 1 < p.k) raw[mt][s][e][1] = A8[(long long)(ke + 1) * p.lda + i];
              }
            }
          }
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kb = k0 + 16 * h + 8 * s;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = job.j0 + 32 * nt + li;
          GM const unsigned short* col = B + (long long)j * p.ldb + kb;
          if constexpr (EXACT && BL) { (void)col; }
          else if constexpr (EXACT) bfr[nt][s] = *(GM const u32x4*)col;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned int lo = (j < p.n && kb + 2 * e < p.k) ? col[2 * e] : 0u;
              const unsigned int hi = (j < p.n && kb + 2 * e + 1 < p.k) ? col[2 * e + 1] : 0u;
              bfr[nt][s][e] = lo | (hi << 16);
            }
          }
        }
      }
      if constexpr (EXACT && BL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int s = 0; s < 2; ++s) { const int f = 32 * nt + li; bfr[nt][s] = *(const u32x4*)(ximage + f * 64 + (((2 * h + s) ^ ((f >> 1) & 3)) * 16)); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // (the image is refilled by the next chunk's requests)
      } else asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) af[mt][s][e] = w8_pair_to_bf16<KIND>(raw[mt][s][e][0] | (raw[mt][s][e][1] << 8), scf[mt]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[nt][s]), __builtin_bit_cast(bf16x8, af[mt][s]), acc[mt][nt], 0, 0, 0);
    }
  }
  // (ragged tiles: plain stores -- their columns are pieces of cache lines, which non-temporal stores would send to memory one by one)
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<EXACT, false, EXACT>(acc[mt][nt], p, q, tc[mt][nt]); });
}

// ------------------------------------------------------------------------------------------------
// bf16 streaming kernel: exact tiles, VNNI-2 A, flat B with 16-byte aligned columns.  Same arithmetic as
// gemm_mfma_bf16_kernel<MT,NT,true>; what differs is how B reaches the matrix core.  The MFMA operand wants
// 8 consecutive k of ONE column per lane, i.e. lanes 128 bytes apart in memory: fetched directly, every load
// instruction touches 32 cache lines for 16 bytes each and the tile is re-fetched from L2 several times.
// Here the 32-deep K chunk of the B tile is brought in with LDS-DMA (global_load_lds_dwordx4: four lanes
// cover the 64 bytes of one column, whole rows per instruction, no VGPRs), the 16-byte slots XOR-swizzled on
// the SOURCE side (the DMA destination is lane-linear) so that the per-column ds_read_b128 is conflict free.
// A (VNNI-2: a dword = two k of one row, rows contiguous) is already coalesced and goes straight to VGPRs.
// The LDS image is wave-private: no barrier, only s_waitcnt.
// ------------------------------------------------------------------------------------------------
// AUX: cache-policy bits of the operand loads (0 default, 2 = nt for launches whose operands cannot be cache resident, see launch_gemm)
template <int MT, int NT, int AUX, bool F16 = false>      // F16: IEEE halves, same layouts (the start value rounded to f16: tile_init)
__global__ __launch_bounds__(256) void gemm_bf16_stream_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds_all[4][2][NT * 2048];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  char* lds = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))][0];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h;
      tc[mt][nt].ivalid = true;
      tile_init<true, false, false, false, F16>(acc[mt][nt], p, q, tc[mt][nt]);
    }
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  // DMA: LDS slot L = lane + 64x holds column f = L>>2, 16-byte piece (L&3) ^ ((f>>1)&3) of the chunk's 64 bytes
  unsigned int offB[NT * 2];
#pragma unroll
  for (int x = 0; x < NT * 2; ++x) {
    const unsigned int L = (unsigned int)lane + 64u * x, f = L >> 2, pc = (L & 3u) ^ ((f >> 1) & 3u);
    offB[x] = (f * ldb) * 2u + pc * 16u;
  }
  // buffer addressing (wave-uniform 4-SGPR resources, loop-invariant 32-bit lane offsets, scalar k-chunk offset): no 64-bit
  // address VGPRs and no per-iteration address arithmetic
  unsigned int voffA[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) voffA[s][e] = ((8u * h + 4u * s + e) * lda + (unsigned int)li) * 4u;     // dword (k-pair 8h + 4s + e, row li)
  const int kchunks = p.k >> 5;
  // Two 32-deep chunks are issued together (second LDS image, second set of A registers): a k = 64 problem has ALL its operand
  // bytes in flight at once, so a wave pays one memory latency per problem instead of two.  An odd chunk count ends with a single.
  auto issue = [&](const __amdgpu_buffer_rsrc_t& ra, const __amdgpu_buffer_rsrc_t& rb, int kc, char* image, u32x4 (&af)[MT][2]) {
#pragma unroll
    for (int x = 0; x < NT * 2; ++x)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(image + 1024 * x), 16, (int)offB[x], 64 * kc, 0, AUX);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          af[mt][s][e] = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)voffA[s][e] + 128 * mt, 64 * kc * (int)lda, 0);   // dword loads: policy bits cost on sub-16-byte accesses
  };
  auto compute = [&](const char* image, const u32x4 (&af)[MT][2]) {
    u32x4 bfr[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int f = 32 * nt + li;
        bfr[nt][s] = *(const u32x4*)(image + f * 64 + (((2 * h + s) ^ ((f >> 1) & 3)) * 16));
      }
#pragma unroll
    for (int s = 0; s < 2; ++s)
      static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
        acc[mt][nt] = mfma_16bit<F16>(bfr[nt][s], af[mt][s], acc[mt][nt]); });
  };
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br + 2ull * (unsigned long long)job.j0 * ldb);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + 4ull * (unsigned long long)job.i0);
    int kc = 0;
    for (; kc + 1 < kchunks; kc += 2) {
      u32x4 af0[MT][2], af1[MT][2];
      issue(ra, rb, kc, lds, af0);
      issue(ra, rb, kc + 1, lds + NT * 2048, af1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      compute(lds, af0);
      compute(lds + NT * 2048, af1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // LDS reads retired before the images are refilled
    }
    if (kc < kchunks) {
      u32x4 af0[MT][2];
      issue(ra, rb, kc, lds, af0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      compute(lds, af0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<true, false>(acc[mt][nt], p, q, tc[mt][nt]); });
}

// ------------------------------------------------------------------------------------------------
// bf16 GEMM in the OTHER operand forms the dense loop accepts [ref: gemm ref :2127-2170 (the loop), :2149-2161 (transposed / VNNI addressing), :2803-2815 (C in
// VNNI)]: A flat (i contiguous, A[k * lda + i]) or transposed (k contiguous, A[i * lda + k]) instead of VNNI-2; B transposed (j contiguous, B[k * ldb + j]) or
// transposed in VNNI-2 (dword (k pair, j) at [kp * ldb + j]) instead of flat; C optionally written in VNNI-2.  Until round 4 all of these ran on the exact VALU
// kernel at 0.003 - 0.007 of the HBM roofline (measured: profiles/r04_forms_before.jsonl) against 0.74 for the VNNI-A / flat-B form.
// One wave per 32 x 32 tile.  Per 32-deep chunk both operand tiles (2 KiB each) are fetched AS THEY LIE IN MEMORY -- whole 64- or 128-byte rows, 16 bytes per lane --
// and parked in a wave-private LDS image of the same shape; the re-layout happens on the way OUT of the image, where it is free: a lane assembles its MFMA operand
// (eight consecutive k of its row / column) with the access the form asks for
//     k contiguous in the image (A transposed, B flat):      one ds_read_b128; the 16-byte slots of a row are XOR-swizzled by (row >> 2) & 3 (conflict free)
//     VNNI-2 dwords (A VNNI, B transposed VNNI):             four ds_read_b32 down a column of the image (lanes along the row: conflict free)
//     outer index contiguous (A flat, B transposed):         eight ds_read_u16 down a column, packed in pairs
// No barrier (wave-private images); the next chunk's global loads are issued before the chunk's two MFMAs.  Any batch form, batch-reduce mode and epilogue of the
// VNNI-A kernels (tile_init / tile_store); C in VNNI-2 with a plain epilogue.  m, n, k multiples of 32, 16-byte aligned rows.
// ------------------------------------------------------------------------------------------------
enum { AF_VNNI = 0, AF_FLAT = 1, AF_TRANS = 2, BF_FLAT = 0, BF_TRANS = 1, BF_TVNNI = 2 };
// operand tile in its memory shape: FORM 0 = [32 outer][32 k] halves, k contiguous; 1 = [32 k][32 outer] halves, outer contiguous; 2 = [16 k pairs][32 outer] dwords
template <int FORM> struct FormsTile {
  static constexpr unsigned int row_bytes = FORM == 2 ? 128u : 64u, ppr = row_bytes / 16u;          // 16-byte pieces per row
  __device__ static __forceinline__ unsigned int ld_bytes(unsigned int ld) { return FORM == 2 ? ld * 4u : ld * 2u; }
  // byte offsets of the wave's tile inside the operand block: outer origin o0, chunk kc
  __device__ static __forceinline__ unsigned long long origin(unsigned int ld, unsigned int o0, unsigned int kc) {
    return FORM == 0 ? 2ull * o0 * ld + 64ull * kc : (FORM == 1 ? 2ull * o0 + 64ull * kc * ld : 4ull * o0 + 64ull * kc * ld);
  }
  __device__ static __forceinline__ void lane_offsets(unsigned int lane, unsigned int ld, unsigned int (&goff)[2], unsigned int (&loff)[2]) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const unsigned int P = lane + 64u * x, row = P / ppr, c16 = P % ppr;
      goff[x] = row * ld_bytes(ld) + c16 * 16u;
      loff[x] = row * row_bytes + ((FORM == 0 ? (c16 ^ ((row >> 2) & 3u)) : c16) * 16u);
    }
  }
  // eight consecutive k (16 s + 8 h ..) of outer index o as the MFMA's 16-byte operand
  __device__ static __forceinline__ u32x4 fragment(const char* img, unsigned int o, unsigned int s, unsigned int h) {
    u32x4 f;
    if constexpr (FORM == 0) f = *(const u32x4*)(img + o * 64u + (((2u * s + h) ^ ((o >> 2) & 3u)) * 16u));
    else if constexpr (FORM == 2) {
#pragma unroll
      for (int d = 0; d < 4; ++d) f[d] = *(const unsigned int*)(img + (8u * s + 4u * h + d) * 128u + o * 4u);
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned int k = 16u * s + 8u * h + 2u * d;
        const unsigned int lo = *(const unsigned short*)(img + k * 64u + o * 2u), hi = *(const unsigned short*)(img + (k + 1u) * 64u + o * 2u);
        f[d] = lo | (hi << 16);
      }
    }
    return f;
  }
};
template <int AF, int BF, bool VNNI_C>
__global__ __launch_bounds__(256) void gemm_bf16_forms_kernel(GemmArgs p) {
  // A forms -> tile shapes: VNNI = dword rows (2), flat = outer contiguous (1), transposed = k contiguous (0); B: flat = k contiguous (0), transposed = (1), transposed VNNI = (2)
  using TA_ = FormsTile<AF == AF_VNNI ? 2 : (AF == AF_FLAT ? 1 : 0)>;
  using TB_ = FormsTile<BF == BF_FLAT ? 0 : (BF == BF_TRANS ? 1 : 2)>;
  __shared__ __attribute__((aligned(16))) char lds_all[4][2][2048];
  const WaveJob job = wave_job(p, 32, 32);
  if (!job.active) return;
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* ia = lds_all[wave][0]; char* ib = lds_all[wave][1];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc;
  TileCtx tc; tc.i = job.i0 + (int)li; tc.j0 = job.j0; tc.h = (int)h; tc.ivalid = true;
  if constexpr (VNNI_C) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  } else tile_init<true, false>(acc, p, q, tc);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  unsigned int ga[2], la[2], gb[2], lb[2];
  TA_::lane_offsets(lane, lda, ga, la); TB_::lane_offsets(lane, ldb, gb, lb);
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  const unsigned long long total = p.br_count * kchunks;
  gcptr ar = nullptr, br = nullptr;
  if (p.br_count != 0) br_base(p, q, 0, ar, br);
  u32x4 va[2], vb[2];
  auto request = [&](unsigned int kc) {
    gcptr at = ar + TA_::origin(lda, (unsigned int)job.i0, kc), bt = br + TB_::origin(ldb, (unsigned int)job.j0, kc);
#pragma unroll
    for (int x = 0; x < 2; ++x) { va[x] = *(GM const u32x4*)(at + ga[x]); vb[x] = *(GM const u32x4*)(bt + gb[x]); }
  };
  if (total != 0) request(0);
  unsigned long long r = 0; unsigned int kc = 0;
  for (unsigned long long t = 0; t < total; ++t) {
#pragma unroll
    for (int x = 0; x < 2; ++x) { *(u32x4*)(ia + la[x]) = va[x]; *(u32x4*)(ib + lb[x]) = vb[x]; }
    u32x4 fa[2], fb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { fa[s] = TA_::fragment(ia, li, (unsigned int)s, h); fb[s] = TB_::fragment(ib, li, (unsigned int)s, h); }
    if (++kc == kchunks) { kc = 0; if (++r < p.br_count) br_base(p, q, r, ar, br); }
    if (t + 1 < total) request(kc);
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = mfma_16bit<false>(fb[s], fa[s], acc);
  }
  if constexpr (VNNI_C) {
    // C in VNNI-2 [ref: gemm ref :2803-2815]: dword (j / 2, i) = (C(i, j), C(i, j + 1)); registers (2g, 2g + 1) of a lane are columns (j, j + 1), j even
    GM unsigned int* c32 = (GM unsigned int*)q.c;
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      const unsigned int j = (unsigned int)job.j0 + (unsigned int)jl_of(2 * g2, (int)h);
      st_stream(c32 + (unsigned long long)(j >> 1) * (unsigned int)p.ldc + (unsigned int)tc.i, bf16_pk_exact(acc[2 * g2], acc[2 * g2 + 1]));
    }
  } else tile_store<true, false>(acc, p, q, tc);
}

// ------------------------------------------------------------------------------------------------
// bf16 64 x 64 x K problems (VNNI-2 A, flat B), one problem per WORKGROUP: the bf16 sibling of gemm_f32_wg64_kernel and the kernel of
// BASELINE config #5.  gemm_bf16_stream_kernel<2,2> gives a wave the whole 64 x 64 tile and fetches A as 32 dword loads per lane; here the
// four waves share the problem, BOTH operands of a 32-deep K step arrive by LDS-DMA (A: [16 k-pairs][64 rows] dwords = 4 KiB, linear;
// B: [64 columns][32 k] = 4 KiB, 16-byte pieces XOR-swizzled on the source side), two DMA instructions per wave and step, and wave
// (wi, wj) runs its two 32x32x16 MFMAs from conflict-free LDS reads.  Two images: a k = 64 problem has all its operand bytes in flight
// before the first MFMA.  Any batch form, any epilogue (batch_ptrs / tile_init / tile_store).
// ------------------------------------------------------------------------------------------------
template <int AUX, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_bf16_wg64_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned int lds_all[2][2048];       // per image: A dwords [16][64], then B bytes [64][64]
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int bidx = logical_block(p);
  const BatchPtrs q = batch_ptrs(p, bidx);
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  // DMA pieces of this wave: L = lane + 64 w of the A image (row kp = L >> 4, dwords 4 (L & 15) ..) and of the B image (column f = L >> 2,
  // 16-byte piece (L & 3) ^ ((f >> 1) & 3) of the chunk's 64 bytes)
  const unsigned int L = lane + 64u * w;
  const unsigned int offA = ((L >> 4) * lda + 4u * (L & 15u)) * 4u;
  const unsigned int fB = L >> 2, offB = fB * ldb * 2u + (((L & 3u) ^ ((fB >> 1) & 3u)) * 16u);
  const unsigned int kchunks = (unsigned int)p.k >> 5;
  const unsigned long long total = p.br_count * kchunks;
  gcptr ar, br;
  auto issue = [&](unsigned long long t) {
    const unsigned int r = (unsigned int)t / kchunks, kc = (unsigned int)t - r * kchunks;
    br_base(p, q, r, ar, br);
    unsigned int* img = lds_all[t & 1ull];
    __builtin_amdgcn_global_load_lds((GM const void*)(ar + (unsigned long long)kc * 64ull * lda + offA), (lds_vptr)((char*)img + 1024 * w), 16, 0, AUX);
    __builtin_amdgcn_global_load_lds((GM const void*)(br + 64ull * kc + offB), (lds_vptr)((char*)img + 4096 + 1024 * w), 16, 0, AUX);
  };
  const unsigned int wi = w & 1u, wj = w >> 1;
  f32x16 acc;
  TileCtx tc; tc.i = (int)(32u * wi + li); tc.j0 = (int)(32u * wj); tc.h = (int)h; tc.ivalid = true;
  tile_init<true, false, false, false, F16>(acc, p, q, tc);
  if (total > 0) issue(0);
  if (total > 1) issue(1);
  for (unsigned long long t = 0; t < total; ++t) {
    if (t + 1 < total) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    const unsigned int* img = lds_all[t & 1ull];
    u32x4 af[2], bfr[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // A operand of MFMA step s2: row i = 32 wi + li, k = 16 s2 + 8 h .. + 7 = k-pairs 8 s2 + 4 h + e
#pragma unroll
      for (int e = 0; e < 4; ++e) af[s2][e] = img[(8u * s2 + 4u * h + e) * 64u + 32u * wi + li];
      const unsigned int f = 32u * wj + li;
      bfr[s2] = *(const u32x4*)((const char*)img + 4096 + f * 64u + (((2u * s2 + h) ^ ((f >> 1) & 3u)) * 16u));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 2 < total) { wg_barrier(); issue(t + 2); }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      acc = mfma_16bit<F16>(bfr[s2], af[s2], acc);
  }
  tile_store<true, false>(acc, p, q, tc);
}

// ------------------------------------------------------------------------------------------------
// bf16 macro-tile kernel: the operand-reuse regime for bf16 (libxsmm_hip_gemm_batch_strided_2d on 16^3 / 32 x 32 x K / 64 x 64 x K problems, VNNI-2 A,
// flat B, plain epilogue).  The bf16 matrix pipe does a 32 x 32 x 16 step in 32 cycles, 8 x the f32 rate per operand byte: with the f32 blocked
// kernel's 128 x 128 macro tile (64 x 64 per wave) the fragment reads plus the LDS-DMA writes would need 190 bytes per cycle of LDS.  Here a
// workgroup owns 256 x 256 of C (4 x 4 problems of 64, 8 x 8 of 32, 16 x 16 of 16) and every wave 128 x 128 of it -- 16 accumulators of 32 x 32,
// 256 AGPRs, one wave per SIMD -- so a fragment is used by four MFMAs.
//   * K advances in stages of 32: A as [16 k-pairs][256 rows] dwords (16 KiB), B as [256 columns][64 bytes] (16 KiB, 16-byte chunks XOR-swizzled
//     by the column so that the b128 fragment reads are conflict free), four stages in a 128 KiB ring, all of it by LDS-DMA three stages ahead.
//   * One wave per SIMD hides nothing behind another wave: the loop is software-pipelined on k-steps of 16 (the fragments of step u + 1 are read
//     while the 16 MFMAs of step u issue), ONE workgroup barrier per stage, and exactly one other instruction is placed behind each MFMA.
//   * Round 3 (profiles/r03_bf16_macro_ablation.txt): the round-2 kernel spent 15 % of a stage on its 8 requests -- each came with a 64-bit VALU
//     address and the stage base was rebuilt with scalar multiplies (54 SALU + 36 VALU per 32 MFMAs); barrier and bank conflicts cost nothing.  Now a
//     request is a raw BUFFER load to LDS: one resource per operand panel, the per-lane offsets are eight loop-invariant VGPRs, the stage offset is
//     ONE SGPR per operand advanced by an add; the stage loop is unrolled over the four ring slots, so every LDS fragment address is
//     "loop-invariant VGPR + immediate".  0.726 -> 0.68 us per stage; what remains is the issue cost of a request itself (about 55 cycles against the
//     32-cycle shadow of an MFMA, wherever it is placed and also with two waves per SIMD) on top of a body that runs at the chip's POWER roof: bare
//     MFMAs on register operands with these operand values sustain 73 % of the 2.5 PF figure (1.75 GHz), this kernel without requests 70 %.
//   * Round 4: the four waves run identical code, so all four SIMDs of the CU issue their request behind the same MFMA and could queue behind each other at
//     the one texture addresser.  Per-wave copies of the steady-state loop with the requests 0 / 1 / 3 / 2 MFMAs earlier (requests behind MFMAs {3,7,11,15},
//     {2,6,10,14}, {0,4,8,12}, {5,9,13,15} of a region) measured 95.0 / 355.2 / 718.1 us against 94.1 / 353.6 / 712.0 us (4096^3, K = 16 384, 8192^3; verified):
//     no collision to remove -- not adopted (profiles/r04b_bf16_macro_staggered_requests_not_adopted.jsonl).
//   * 16 x 16 x 16 problems (PE = 16, K16): a stage is two consecutive batch-reduce blocks.
// Accumulation order per output = the k order of the single-problem kernels: bitwise the same results.
// ------------------------------------------------------------------------------------------------
// FORM of the C stores (decided by the host): 0 f32, 1 bf16 as packed dwords (even ldc, 4-byte aligned tiles), 2 bf16 element by element.
// Wave layout: NW waves (4: one per SIMD, or 8: two per SIMD), each TI x TJ accumulator tiles of 32 x 32 (NW * TI * TJ = 64 tiles = 256 x 256).
// ABL: timing-only experiments (wrong results): 1 no A requests, 2 no B requests.
// The other instructions of a region (fragment reads, requests) are written in the order they are to issue and the group barriers place them one
// by one behind the region's MFMAs: bm_item_is_vmem(TI, TJ, RA, idx) = type of the idx-th of them (see the emission loops in the kernel).
__host__ __device__ constexpr bool bm_item_is_vmem(int TI, int TJ, int RA, int idx) {
  const int Q = TI > TJ ? TI : TJ;
  int n = 0;
  for (int q = 0; q < Q; ++q) {
    if (q < TI) { if (idx == n || idx == n + 1) return false; n += 2; }      // an A fragment = two ds_read2st64_b32
    if (q < TJ) { if (idx == n) return false; n += 1; }                      // a B fragment = one ds_read_b128
    if (q < RA) { if (idx == n) return true; n += 1; }                       // one request
  }
  return false;
}
template <int TI, int TJ, int RA> __device__ __forceinline__ void bm_region_schedule() {
  constexpr int M = TI * TJ, N = 2 * TI + TJ + RA;
  static_for<M>([&](auto ic) {
    constexpr int i = ic.value, lo = i * N / M, hi = (i + 1) * N / M;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    static_for<hi - lo>([&](auto jc) {
      if constexpr (bm_item_is_vmem(TI, TJ, RA, lo + jc.value)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    });
  });
}
template <int FORM, int PE, bool K16, int NW = 4, int TI = 4, int TJ = 4, int ABL = 0, bool F16 = false>
__global__ __launch_bounds__(64 * NW, 1) void gemm_bf16_macro_kernel(GemmArgs p) {
  static_assert(NW * TI * TJ == 64 && (NW == 4 || NW == 8), "256 x 256 macro tile");
  constexpr unsigned int PPM = 256 / PE;                           // problems per macro-tile edge
  constexpr int RA = 16 / NW;                                      // requests per wave for each half (A, B) of a stage
  constexpr unsigned int WI = 8 / TI;                              // waves along i
  extern __shared__ __attribute__((aligned(16))) unsigned int bm_lds[];
  constexpr unsigned int STAGE = 8192;                             // dwords per stage: A [16 k-pairs][256 rows] | B [256 columns][16 dwords]
  constexpr unsigned int NSLOT = 4;
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int lane = threadIdx.x & 63u, li = lane & 31u, h = lane >> 5;
  const unsigned int ni = p.batch_inner, MI = ni / PPM, MJ = (p.nbatch / ni) / PPM;
  // Hardware workgroup g runs on XCD g % 8 (its own 4 MiB L2).  The macro-tile grid is cut 4 x 2 so that every XCD owns a compact rectangle --
  // 16 x 16 macro tiles: 4 x 8 per XCD, an A panel shared by 8 of its workgroups and a B panel by 4 -- or, when the grid does not divide that
  // way, a contiguous band of macro columns.
  unsigned int g = blockIdx.x, mi, mj;
  if ((MI & 3u) == 0u && (MJ & 1u) == 0u) {
    const unsigned int x = g & 7u, k = g >> 3, RI = MI >> 2, rj = k / RI, ri = k - rj * RI;
    mi = (x & 3u) * RI + ri; mj = (x >> 2) * (MJ >> 1) + rj;
  } else {
    if ((gridDim.x & 7u) == 0u) g = (g & 7u) * (gridDim.x >> 3) + (g >> 3);
    mj = g / MI; mi = g - mj * MI;
  }
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int brs_a = p.br_mode == 3 ? (unsigned int)p.br_stride_a : 0u, brs_b = p.br_mode == 3 ? (unsigned int)p.br_stride_b : 0u;
  const __amdgpu_buffer_rsrc_t ra = wave_rsrc((gcptr)p.a + (long long)(mi * PPM) * p.bs_a);
  const __amdgpu_buffer_rsrc_t rb = wave_rsrc((gcptr)p.b + (long long)(mj * PPM) * p.bs_b);
  // --- DMA duty (launch_gemm checks that every offset below fits 32 bits).  A: request x brings k-pair RA w + x of the stage, lane = (problem
  // 4 lane / PE, rows 4 lane % PE .. + 3).  B: request x brings columns 16 (RA w + x) .. + 15, lane = (column lane / 4, chunk slot lane % 4); the
  // 16-byte chunk that lands in slot s of column c is chunk s ^ ((c >> 3) & 3) of the stage's 64 bytes: a ds_read_b128 is serviced in the lane
  // groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), and the four lanes of a group whose columns agree mod 4 (same 16 banks) are li / 4 in
  // {0, 3, 5, 6} or {1, 2, 4, 7}: (li >> 3) & 3 tells them apart.  K16: k-pairs 8 .. 15 and chunks 2, 3 belong to the stage's second block.
  unsigned int voffA[RA], voffB[RA];
#pragma unroll
  for (int x = 0; x < RA; ++x) {
    const unsigned int kp = (unsigned int)RA * w + (unsigned int)x;
    voffA[x] = ((4u * lane) / PE) * (unsigned int)p.bs_a + 4u * ((4u * lane) % PE) + (K16 ? (kp >> 3) * brs_a + (kp & 7u) * lda * 4u : kp * lda * 4u);
    const unsigned int col = 16u * kp + (lane >> 2), c = (lane & 3u) ^ ((col >> 3) & 3u);
    voffB[x] = (col / PE) * (unsigned int)p.bs_b + (col % PE) * ldb * 2u + (K16 ? (c >> 1) * brs_b + (c & 1u) * 16u : c * 16u);
  }
  const unsigned int kchunks = K16 ? 1u : ((unsigned int)p.k >> 5);
  const unsigned int total = K16 ? (unsigned int)(p.br_count >> 1) : (unsigned int)p.br_count * kchunks;
  // stage offsets (SGPRs): within a block a stage is 16 k-pairs of A (64 lda bytes) and 64 bytes of every B column
  const unsigned int stepA = K16 ? 2u * brs_a : 64u * lda, stepB = K16 ? 2u * brs_b : 64u;
  const unsigned int wrapA = brs_a - kchunks * 64u * lda, wrapB = brs_b - kchunks * 64u;       // (unsigned wrap-around is fine: added modulo 2^32)
  // Requests are written in HALVES of a stage (A half, B half; order A0 B0 A1 B1 ...), one request at a time between the fragment reads: the
  // compiler must see requests and reads in the order they are to issue (a request writes LDS, so it never moves across an LDS read).
  unsigned int sA = 0, sB = 0, kcA = 0, kcB = 0, nA = 0, nB = 0;    // nA / nB: stages whose A / B half has been requested
  auto reqA = [&](unsigned int slot, int x) {
    if (!(ABL & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)(bm_lds + slot * STAGE + 256u * ((unsigned int)RA * w + (unsigned int)x)), 16, (int)voffA[x], (int)sA, 0, 0);
    if (x == RA - 1) { sA += stepA; ++nA; if (!K16) { if (++kcA == kchunks) { kcA = 0; sA += wrapA; } } }
  };
  auto reqB = [&](unsigned int slot, int x) {
    if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(bm_lds + slot * STAGE + 4096u + 256u * ((unsigned int)RA * w + (unsigned int)x)), 16, (int)voffB[x], (int)sB, 0, 0);
    if (x == RA - 1) { sB += stepB; ++nB; if (!K16) { if (++kcB == kchunks) { kcB = 0; sB += wrapB; } } }
  };
  auto issueA = [&](unsigned int slot) { static_for<RA>([&](auto xc) { reqA(slot, xc.value); }); };
  auto issueB = [&](unsigned int slot) { static_for<RA>([&](auto xc) { reqB(slot, xc.value); }); };
  // every request up to and including the B half of stage `s` has landed when at most (nA + nB - 2 s - 2) halves are behind it
  auto wait_stage = [&](unsigned int s_) {
    const unsigned int behind = nA + nB - 2u * s_ - 2u;
    if (RA == 4) {
      if (behind >= 4u) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (behind == 3u) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (behind == 2u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (behind == 1u) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (behind >= 4u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (behind == 3u) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (behind == 2u) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (behind == 1u) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  // --- compute duty: the (32 TI) x (32 TJ) part (wi, wj)
  const unsigned int wi = w % WI, wj = w / WI;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
  if (total == 0) return;
  // fragment addresses inside a stage (dwords): A row 32 TI wi + 32 ti + li, k-pairs 8 s2 + 4 h + e (256 dwords apart); B column
  // 32 TJ wj + 32 tj + li, chunk (2 s2 + h) ^ ((li >> 3) & 3).  One opaque base per A fragment: its four k-pairs then pair up as
  // ds_read2st64_b32 into consecutive registers (left to itself the compiler pairs the same k-pair of two fragments and needs moves -- and a full LDS wait -- to build the operand); everything else is an immediate.
  unsigned int fa[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) { fa[ti] = 1024u * h + 32u * (unsigned int)TI * wi + 32u * (unsigned int)ti + li; asm volatile("" : "+v"(fa[ti])); }
  const unsigned int sw = (li >> 3) & 3u;
  unsigned int fb[2] = {4096u + (32u * (unsigned int)TJ * wj + li) * 16u + 4u * (h ^ sw), 4096u + (32u * (unsigned int)TJ * wj + li) * 16u + 4u * ((2u + h) ^ sw)};
  asm volatile("" : "+v"(fb[0])); asm volatile("" : "+v"(fb[1]));
  struct Frags { u32x4 a[TI]; u32x4 b[TJ]; };
  constexpr int Q = TI > TJ ? TI : TJ;
  // quarter q of a region's other work, in issue order: A fragment q, B fragment q, request q
  auto read_q = [&](Frags& f, unsigned int slot, int s2, int q) {
    const unsigned int* st = bm_lds + slot * STAGE;
    if (q < TI) {
#pragma unroll
      for (int e = 0; e < 4; ++e) f.a[q][e] = st[fa[q] + 2048u * (unsigned int)s2 + 256u * (unsigned int)e];
    }
    if (q < TJ) f.b[q] = *(const u32x4*)(st + fb[s2] + 512 * q);
  };
  auto read = [&](Frags& f, unsigned int slot, int s2) { static_for<Q>([&](auto qc) { read_q(f, slot, s2, qc.value); }); };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
        acc[ti][tj] = mfma_16bit<F16>(f.b[tj], f.a[ti], acc[ti][tj]);
  };
  // prologue: stages 0 .. 2 and the A half of stage 3 (as far as they exist)
  for (unsigned int u = 0; u < NSLOT; ++u) {
    if (u < total) issueA(u);
    if (u < total && u + 1u < NSLOT) issueB(u);
  }
  wait_stage(0);
  wg_barrier();
  Frags f0, f1;
  read(f0, 0, 0);
  unsigned int t = 0;
  // Steady state, four stages per trip (slot = compile-time).  Per stage two straight-line regions split by the one barrier:
  //   region 1: the MFMAs of k-step 0, the fragment reads of k-step 1, the B half of stage t + 3 (slot freed by the previous barrier)
  //   region 2: the MFMAs of k-step 1, the fragment reads of the next stage's k-step 0, the A half of stage t + 4 (slot freed by this barrier)
  for (; t + NSLOT + 3u < total; t += 4u) {
    static_for<4>([&](auto sc) {
      constexpr unsigned int S = (unsigned int)sc.value, SN = (S + 1u) & 3u;
      static_for<Q>([&](auto qc) { read_q(f1, S, 1, qc.value); if constexpr (qc.value < RA) reqB((S + 3u) & 3u, qc.value); });       // B half of stage t + 3
      mfmas(f0);
      bm_region_schedule<TI, TJ, RA>();
      __builtin_amdgcn_sched_barrier(0);                           // nothing moves across the region boundary (the unrolled stages are one basic block)
      if constexpr ((ABL & 3) == 0) { if (RA == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }   // stage t + 1 has landed
      else if constexpr ((ABL & 3) != 3) { if (RA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      wg_barrier();                                                // ... for every wave, and every wave has read all of stage t
      static_for<Q>([&](auto qc) { read_q(f0, SN, 0, qc.value); if constexpr (qc.value < RA) reqA(S, qc.value); });                  // A half of stage t + 4 takes stage t's place
      mfmas(f1);
      bm_region_schedule<TI, TJ, RA>();
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  unsigned int slot = 0;                                           // t % 4 == 0 here
  for (; t < total; ++t) {                                         // the last stages (fewer than eight): the same steps behind their conditions
    read(f1, slot, 1);
    if (nB < total) issueB(nB & 3u);
    mfmas(f0);
    const unsigned int nslot = slot == NSLOT - 1u ? 0u : slot + 1u;
    if (t + 1u < total) {
      wait_stage(t + 1u);
      wg_barrier();
      read(f0, nslot, 0);
      if (nA < total) issueA(nA & 3u);
    }
    mfmas(f1);
    slot = nslot;
  }
  // --- C.  Element (row, col) of the macro tile belongs to problem (row / PE, col / PE); a 32 x 32 accumulator tile lies inside one problem
  // for PE >= 32 and covers 2 x 2 problems for PE = 16 (the lane's row picks the problem row, the register's column the problem column).
  const bool odd = (lane & 1u) != 0;
  const unsigned int sel = odd ? 0x03020706u : 0x05040100u;
  constexpr int ES = FORM == 0 ? 4 : 2;
  auto row_off = [&](int ti, unsigned int sub) -> long long {     // byte offset of this lane's row (minus `sub` rows) inside C
    const unsigned int row = 32u * (unsigned int)TI * wi + 32u * (unsigned int)ti + li - sub;
    return (long long)(mi * PPM + row / PE) * p.bs_c + (long long)(row % PE) * ES;
  };
  auto col_off = [&](unsigned int col) -> long long {
    return (long long)(mj * PPM + col / PE) * p.bs_c2 + (long long)(col % PE) * p.ldc * ES;
  };
  static_for<TI * TJ>([&](auto idx) {
    constexpr int ti = idx.value / TJ, tj = idx.value % TJ;
    const unsigned int col0 = 32u * (unsigned int)TJ * wj + 32u * (unsigned int)tj + 4u * h;
    if constexpr (FORM == 0) {
      GM char* base = (GM char*)p.c + row_off(ti, 0u);
      static_for<16>([&](auto rc) { constexpr int r = rc.value; st_stream((GM float*)(base + col_off(col0 + (unsigned int)((r & 3) + 8 * (r >> 2)))), acc[ti][tj][r]); });
    } else if constexpr (FORM == 1) {
      // lanes (2q, 2q + 1) hold rows (2q, 2q + 1) of a column: the even lane stores the packed pair of the even register's column, the odd lane
      // that of the odd register's column (rows 2q, 2q + 1 both times): see tile_store_impl
      GM char* base = (GM char*)p.c + row_off(ti, odd ? 1u : 0u);
      static_for<8>([&](auto gc) {
        constexpr int r0 = 2 * gc.value, jr = (r0 & 3) + 8 * (r0 >> 2);
        const unsigned int wv = F16 ? cvt_pk_f16(acc[ti][tj][r0], acc[ti][tj][r0 + 1]) : bf16_pk_exact(acc[ti][tj][r0], acc[ti][tj][r0 + 1]);
        const unsigned int nv = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)wv, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        st_stream((GM unsigned int*)(base + col_off(col0 + (unsigned int)jr + (odd ? 1u : 0u))), (unsigned int)__builtin_amdgcn_perm(nv, wv, sel));
      });
    } else {
      GM char* base = (GM char*)p.c + row_off(ti, 0u);
      static_for<16>([&](auto rc) { constexpr int r = rc.value; st_stream((GM unsigned short*)(base + col_off(col0 + (unsigned int)((r & 3) + 8 * (r >> 2)))),
                                                                            F16 ? __builtin_bit_cast(unsigned short, (_Float16)acc[ti][tj][r]) : f32_to_bf16_rne(acc[ti][tj][r])); });
    }
    asm volatile("" ::: "memory");
  });
}

// ------------------------------------------------------------------------------------------------
// 8-bit integer streaming kernel (v_mfma_i32_32x32x32_i8): exact tiles, VNNI-4 A, flat B with 16-byte aligned columns,
// k % 64 == 0.  Structure = gemm_bf16_stream_kernel: B through LDS-DMA with source-side swizzle, A straight into
// operand registers.  The matrix core multiplies SIGNED bytes; an unsigned operand u is fed as (u ^ 0x80) = u - 128 and
// the missing 128 * sum_k(other operand) comes from one more MFMA against an all-ones operand (exact integer
// arithmetic): (a'+128) b = a'b + 128 sum b,  a (b'+128) = a b' + 128 sum a,  both: + 128*128*K as well.
// ------------------------------------------------------------------------------------------------
// I4 (round 3): interleaved 4-bit weights [ref: gemm ref :467-477, :1009-1088] -- a dword of A holds eight k of one row (byte t: low nibble k 8o + t, high
// nibble k 8o + 4 + t), every weight minus the row's zero point (one byte per row, a.quaternary) wrapped to a signed byte, B read as UNSIGNED bytes
// (UB = true).  Two dwords of A per lane and MFMA step become the four operand dwords: (w & 0x0f0f0f0f) and ((w >> 4) & 0x0f0f0f0f) are k 8o..8o+3 and
// 8o+4..8o+7 in order; the zero point is subtracted from all four bytes at once (byte-wise subtraction modulo 256 without borrows across bytes).
__device__ __forceinline__ int sub_bytes(unsigned int x, unsigned int y) {
  return (int)((((x | 0x80808080u) - (y & 0x7f7f7f7fu))) ^ ((x ^ ~y) & 0x80808080u));
}
// LB (round 3): the packed low-bit weights expanded to signed bytes in registers -- 1: interleaved 4-bit minus a zero point per row (I4X2); 2: interleaved 2-bit codes
// 0 / +1 / -1 / -1, the rows in four groups of m / 4 sharing a byte (I2X4: one dword per k-quad and lane, the lane's group selects the bit pair, v_perm_b32 maps the
// four codes at once); 3: 1-bit signs, a byte = four k of two rows (I1X8: one byte per k-quad and lane, the nibble spread to one bit per byte by a multiply).
template <int MT, int NT, bool UA, bool UB, int LB = 0>
__global__ __launch_bounds__(256) void gemm_i8_stream_kernel(GemmArgs p) {
  constexpr bool I4 = LB == 1;
  __shared__ __attribute__((aligned(16))) char lds_all[4][NT * 2048];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  char* lds = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  i32x16 acc[MT][NT], sum_b[NT], sum_a[MT];         // sum_b[nt]: 32x32 tile whose every column i holds sum_k b'(j,k); sum_a likewise per row
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; acc[mt][nt] = (i32x16)0; });
  static_for<NT>([&](auto idx) { sum_b[idx.value] = (i32x16)0; });
  static_for<MT>([&](auto idx) { sum_a[idx.value] = (i32x16)0; });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  unsigned int offB[NT * 2], offBt[NT * 2];
#pragma unroll
  for (int x = 0; x < NT * 2; ++x) {
    const unsigned int L = (unsigned int)lane + 64u * x, f = L >> 2, pc = (L & 3u) ^ ((f >> 1) & 3u);
    offB[x] = f * ldb + pc * 16u;
    offBt[x] = f * ldb + (pc & 1u) * 16u;             // half a chunk (k % 64 == 32): only the first 32 bytes of a column's 64 exist -- the pieces 2 / 3 re-read 0 / 1 and are never used
  }
  const unsigned int offA = ((I4 ? 2u * h : 4u * h) * lda + (unsigned int)li) * 4u;      // dword (k-quad 4h, row li); I4: dword (k-group-of-8 2h, row li)
  const i32x4 ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  const int kchunks = p.k >> 6, ktail = (p.k >> 5) & 1;       // whole 64-deep chunks, then one 32-deep half chunk: MFMA step 0 only (round 3: k = 32, 96, ... used to fall to the generic kernel)
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br + (unsigned long long)job.j0 * ldb);     // buffer addressing, see gemm_bf16_stream_kernel
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + 4ull * (unsigned long long)job.i0);
    const __amdgpu_buffer_rsrc_t ra0 = wave_rsrc(ar);          // LB 2 / 3: the lane's row decides where in a k-quad's bytes its bits sit
    unsigned int zz[MT];
    if (I4) {       // the zero points of this lane's rows for batch-reduce block r: one byte per row, stepped with A [run_gemm: bs_scf, (br_stride_a * 2) / k per block]
      const long long brs_a = p.br_mode == 3 ? p.br_stride_a : 0;
      GM const unsigned char* zp = (GM const unsigned char*)p.a_scf + (long long)job.bidx * p.bs_scf + ((brs_a * 2) / p.k) * (long long)r + job.i0 + li;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) zz[mt] = (unsigned int)zp[32 * mt] * 0x01010101u;
    }
    for (int kc = 0; kc < kchunks + ktail; ++kc) {
      const bool half = kc == kchunks;                 // wave-uniform
      const unsigned int s1 = half ? 0u : 1u;          // the k-step the "second" operand loads address: in a half chunk they repeat step 0 (in bounds) and are not multiplied
#pragma unroll
      for (int x = 0; x < NT * 2; ++x)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(lds + 1024 * x), 16, (int)(half ? offBt[x] : offB[x]), 64 * kc, 0, 0);
      i32x4 af[MT][2];
      if constexpr (I4) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const unsigned int w = (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)(offA + (4u * (s ? s1 : 0u) + e) * lda * 4u) + 128 * mt, 32 * kc * (int)lda, 0);
              af[mt][s][2 * e] = sub_bytes(w & 0x0f0f0f0fu, zz[mt]);
              af[mt][s][2 * e + 1] = sub_bytes((w >> 4) & 0x0f0f0f0fu, zz[mt]);
            }
      } else if constexpr (LB == 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned int i = (unsigned int)(job.i0 + 32 * mt + li), mq = (unsigned int)p.m >> 2, g = i / mq, rr = i - g * mq;
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned int w = (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(ra0, (int)((8u * (s ? s1 : 0u) + 4u * h + e) * lda + 4u * rr), 16 * kc * (int)lda, 0);
              af[mt][s][e] = (int)__builtin_amdgcn_perm(0u, 0xffff0100u, (w >> (2u * g)) & 0x03030303u);
            }
        }
      } else if constexpr (LB == 3) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned int i = (unsigned int)(job.i0 + 32 * mt + li);
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned int b = (unsigned int)__builtin_amdgcn_raw_buffer_load_b8(ra0, (int)((8u * (s ? s1 : 0u) + 4u * h + e) * (lda >> 1) + (i >> 1)), 16 * kc * (int)(lda >> 1), 0) & 0xffu;
              const unsigned int x = ((((b >> (4u * (i & 1u))) & 0xfu) * 0x00204081u) & 0x01010101u);
              af[mt][s][e] = (int)((x * 0xfeu) | 0x01010101u);
            }
        }
      } else
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)(offA + (8u * (s ? s1 : 0u) + e) * lda * 4u) + 128 * mt, 64 * kc * (int)lda, 0);
            af[mt][s][e] = UA ? (v ^ (int)0x80808080) : v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      i32x4 bfr[NT][2];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int f = 32 * nt + li;
          i32x4 v = *(const i32x4*)(lds + f * 64 + (((2 * s + h) ^ ((f >> 1) & 3)) * 16));
          if (UB) v ^= (i32x4)(int)0x80808080;
          bfr[nt][s] = v;
        }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 1 && half) break;
        static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
          acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bfr[nt][s], af[mt][s], acc[mt][nt], 0, 0, 0); });
        if (UA) static_for<NT>([&](auto idx) { sum_b[idx.value] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bfr[idx.value][s], ones, sum_b[idx.value], 0, 0, 0); });
        if (UB) static_for<MT>([&](auto idx) { sum_a[idx.value] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ones, af[idx.value][s], sum_a[idx.value], 0, 0, 0); });
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0, c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
  const int kconst = (UA && UB) ? (int)(16384u * (unsigned int)(p.br_count * (unsigned long long)p.k)) : 0;      // (mod 2^32 like the i32 sums themselves: unsigned arithmetic, no signed overflow)
  // (round 4: the 4-byte results through the wave's LDS image and out as whole 128-byte columns, 16 bytes per lane -- four store instructions per tile instead of
  //  sixteen, the f32 streaming kernel's epilogue -- measured no gain: u8 x i8 64^3 0.70 against 0.71, 32^3 0.76 against 0.74; not kept)
  static_for<MT * NT>([&](auto idx) {
    constexpr int mt = idx.value / NT, nt = idx.value % NT;
    GM char* ctile = (GM char*)q.c + 4ull * ((unsigned long long)(job.j0 + 32 * nt + 4 * h) * (unsigned int)p.ldc + job.i0 + 32 * mt + li);
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
      int v = acc[mt][nt][r2] + kconst;
      if (UA) v += 128 * sum_b[nt][r2];
      if (UB) v += 128 * sum_a[mt][r2];
      GM char* cp = ctile + 4ull * (unsigned long long)(((r2 & 3) + 8 * (r2 >> 2)) * p.ldc);
      if (c_f32) { float f = (float)v * p.scf; if (!beta0) f = f + *(GM const float*)cp; *(GM float*)cp = f; }
      else { if (!beta0) v += *(GM const int*)cp; *(GM int*)cp = v; }
    }
  });
}

// ------------------------------------------------------------------------------------------------
// Interleaved MXFP4 weights x signed 8-bit activations -> f32 / bf16 (round 3) [ref: gemm ref :467-477, :1009-1088]: a dword of A holds eight k of one row
// (byte t: low nibble k 8o + t, high nibble k 8o + 4 + t) as E2M1 codes that the reference maps to the INTEGER table {0, 11, 21, 32, 42, 64, 85, 127} with
// sign; the integer sum of a 32-deep block is scaled by the E8M0 scale of (row, block) and an f32 of (column, block) and added to the f32 result, block by
// block.  One v_mfma_i32_32x32x32_i8 IS one block: the accumulator starts at zero for every MFMA, its 16 integers are converted, scaled with the two
// multiplies and added in the reference's order -- bit-identical to the reference loop.  The codes are expanded in registers: v_perm_b32 looks four
// magnitudes up at once in the 8-byte table, the sign is applied byte-wise.  B through the LDS-DMA path of the int8 kernel; the f32 column scales of
// a 64-deep chunk (two blocks x 32 columns per tile) through a wave-private LDS image, read back as four 16-byte pieces per block.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mx4_codes_to_bytes(unsigned int codes) {       // four E2M1 codes (one per byte) -> four signed bytes of the reference's integer table
  // one look-up in the table {0, 11, 21, 32, 42, 64, 85, 127}, one in its negation, the sign bit of each code selects (v_bfi_b32): 7 instructions for four codes
  const unsigned int sel = codes & 0x07070707u;
  const unsigned int pos = (unsigned int)__builtin_amdgcn_perm(0x7f55402au, 0x20150b00u, sel);
  const unsigned int neg = (unsigned int)__builtin_amdgcn_perm(0x81abc0d6u, 0xe0ebf500u, sel);
  const unsigned int sm = ((codes >> 3) & 0x01010101u) * 0xffu;
  return (int)((neg & sm) | (pos & ~sm));
}
// Four waves per SIMD (118 registers instead of 152): the kernel is bound by the latency of one short wave per tile, not by its instructions -- 0.30 / 0.33 ->
// 0.38 / 0.46 (bf16 / f32 C); five waves per SIMD spill 19 registers and halve the speed.
template <int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void gemm_mx4i8_stream_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds_all[4][NT * 2048 + NT * 256];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  char* lds = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  float* lds_sb = (float*)(lds + NT * 2048);                       // [nt][block of the chunk][32 columns]
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  float facc[MT][NT][16];
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0, c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) facc[mt][nt][r2] = 0.0f; });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  unsigned int offB[NT * 2];
#pragma unroll
  for (int x = 0; x < NT * 2; ++x) {
    const unsigned int L = (unsigned int)lane + 64u * x, f = L >> 2, pc = (L & 3u) ^ ((f >> 1) & 3u);
    offB[x] = f * ldb + pc * 16u;
  }
  const unsigned int offA = ((2u * h) * lda + (unsigned int)li) * 4u;      // dword (k-group-of-8 2h, row li)
  const int kchunks = p.k >> 6;
  const long long brs_a = p.br_mode == 3 ? p.br_stride_a : 0, brs_b = p.br_mode == 3 ? p.br_stride_b : 0;
  const int nsb = p.ldb / 32;                                      // f32 scales per column of B
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br + (unsigned long long)job.j0 * ldb);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + 4ull * (unsigned long long)job.i0);
    // scales of this block of the chain [run_gemm: one E8M0 byte per (32 k, row) of A, stepped by (br_stride_a * 2) / 32 per block; one f32 per (column, 32 k) of B]
    GM const unsigned char* sa = (GM const unsigned char*)p.a_scf + (long long)job.bidx * p.bs_scf + ((brs_a * 2) / 32) * (long long)r + job.i0 + li;
    GM const float* sb = (GM const float*)(p.b_scf + (long long)job.bidx * p.bs_bscf) + (brs_b / 32) * (long long)r + (long long)job.j0 * nsb;
    for (int kc = 0; kc < kchunks; ++kc) {
#pragma unroll
      for (int x = 0; x < NT * 2; ++x)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(lds + 1024 * x), 16, (int)offB[x], 64 * kc, 0, 0);
      i32x4 af[MT][2];
      float rs[MT][2];                                             // 2^(scale - 127) of (this lane's row, block 2 kc + s)
      float colsc[NT];                                             // this lane's contribution to the column-scale image: column li, block h of the chunk
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) colsc[nt] = sb[(long long)(32 * nt + li) * nsb + 2 * kc + h];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          rs[mt][s] = __uint_as_float((unsigned int)sa[(long long)(2 * kc + s) * lda + 32 * mt] << 23);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const unsigned int w = (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)(offA + (4u * s + e) * lda * 4u) + 128 * mt, 32 * kc * (int)lda, 0);
            af[mt][s][2 * e] = mx4_codes_to_bytes(w & 0x0f0f0f0fu);
            af[mt][s][2 * e + 1] = mx4_codes_to_bytes((w >> 4) & 0x0f0f0f0fu);
          }
        }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) lds_sb[64 * nt + 32 * h + li] = colsc[nt];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        i32x4 bfr[NT];
        f32x4 cs[NT][4];                                           // column scales of registers 4 g .. 4 g + 3: columns 8 g + 4 h .. + 3
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int f = 32 * nt + li;
          bfr[nt] = *(const i32x4*)(lds + f * 64 + (((2 * s + h) ^ ((f >> 1) & 3)) * 16));
#pragma unroll
          for (int g = 0; g < 4; ++g) cs[nt][g] = *(const f32x4*)(lds_sb + 64 * nt + 32 * s + 8 * g + 4 * h);
        }
        static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
          const i32x16 t = __builtin_amdgcn_mfma_i32_32x32x32_i8(bfr[nt], af[mt][s], (i32x16)0, 0, 0, 0);
#pragma unroll
          for (int r2 = 0; r2 < 16; ++r2)
            facc[mt][nt][r2] = add_rn(facc[mt][nt][r2], mul_rn(mul_rn((float)t[r2], rs[mt][s]), cs[nt][r2 >> 2][r2 & 3]));
        });
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  static_for<MT * NT>([&](auto idx) {
    constexpr int mt = idx.value / NT, nt = idx.value % NT;
    const unsigned long long e0 = (unsigned long long)(job.j0 + 32 * nt + 4 * h) * (unsigned int)p.ldc + job.i0 + 32 * mt + li;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
      const unsigned long long e = e0 + (unsigned long long)(((r2 & 3) + 8 * (r2 >> 2)) * p.ldc);
      if (c_f32) { GM float* c = (GM float*)q.c + e; *c = add_rn(beta0 ? 0.0f : *c, facc[mt][nt][r2]); }
      else { GM unsigned short* c = (GM unsigned short*)q.c + e; *c = f32_to_bf16_rne(add_rn(beta0 ? 0.0f : bf16_to_f32(*c), facc[mt][nt][r2])); }
    }
  });
}


// ------------------------------------------------------------------------------------------------
// The same arithmetic with waves that walk G consecutive tiles (round 3).  gemm_mx4i8_stream_kernel is one short wave per 32 x 32 tile: a load round trip,
// two MFMAs, sixteen 2- or 4-byte stores per lane -- 5 us of wave lifetime for 5 KiB, and 16 + 9 vector-memory instructions whose addresses the texture unit
// works through at a quarter wave per clock.  Here (a) the chunks of the wave's tiles form ONE flat sequence and the operands of chunk f + 1 are requested
// before chunk f is multiplied (all through registers: B fragments are 16 contiguous bytes of a column, which is what the MFMA wants, so no LDS image and
// no hand-counted waits -- the compiler tracks every load), (b) with beta = 0 a finished tile leaves through a column-major LDS image as whole columns:
// 2 (bf16) or 4 (f32) 16-byte stores per lane instead of 16 element stores.  Bit-identical to the one-tile kernel (same sums in the same order).
// ------------------------------------------------------------------------------------------------
struct Mx4i8Tile { BatchPtrs q; unsigned int bidx; int i0, j0; };
__device__ __forceinline__ Mx4i8Tile mx4i8_tile(const GemmArgs& p, unsigned int id) {
  Mx4i8Tile t;
  const unsigned int per_gemm = (unsigned int)(p.tiles_m * p.tiles_n);
  t.bidx = per_gemm == 1 ? id : id / per_gemm;
  const unsigned int r = id - t.bidx * per_gemm, tn = r / (unsigned int)p.tiles_m;
  t.i0 = (int)(r - tn * (unsigned int)p.tiles_m) * 32; t.j0 = (int)tn * 32;
  t.q = batch_ptrs(p, t.bidx);
  return t;
}
struct Mx4i8Chunk { u32x4 b[2]; unsigned int a[4]; unsigned int sa[2]; float cs; };
template <int D>          // chunks in flight per wave (the ring): one round trip of ~2.5 us against ~0.3 us of work per chunk wants about ten per SIMD
__global__ __launch_bounds__(256) void gemm_mx4i8_pipe_kernel(GemmArgs p, unsigned int per_wave, unsigned int total_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_all[4][1024 + 64];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int first = (blockIdx.x * 4u + wave) * per_wave;
  if (first >= total_tiles) return;
  const unsigned int ntiles = total_tiles - first < per_wave ? total_tiles - first : per_wave;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  float* lds = lds_all[wave];
  float* lds_sb = lds + 1024;                                      // [block of the chunk][32 columns]
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0, c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  const unsigned int offA = ((2u * h) * lda + (unsigned int)li) * 4u, offB = (unsigned int)li * ldb + 16u * h;
  const unsigned int kchunks = (unsigned int)p.k >> 6;
  const long long brs_a = p.br_mode == 3 ? p.br_stride_a : 0, brs_b = p.br_mode == 3 ? p.br_stride_b : 0;
  const int nsb = p.ldb / 32;
  const unsigned long long per_tile = p.br_count * kchunks, total = per_tile * ntiles;
  // the chunk the NEXT request goes to
  Mx4i8Tile nx = mx4i8_tile(p, first);
  unsigned int n_tile = 0, n_kc = 0; unsigned long long n_r = 0;
  auto request = [&](Mx4i8Chunk& c) __attribute__((always_inline)) {
    gcptr ar, br; br_base(p, nx.q, n_r, ar, br);
    const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br + (unsigned long long)nx.j0 * ldb);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + 4ull * (unsigned long long)nx.i0);
    GM const unsigned char* sa = (GM const unsigned char*)p.a_scf + (long long)nx.bidx * p.bs_scf + ((brs_a * 2) / 32) * (long long)n_r + nx.i0 + li;
    GM const float* sb = (GM const float*)(p.b_scf + (long long)nx.bidx * p.bs_bscf) + (brs_b / 32) * (long long)n_r + (long long)nx.j0 * nsb;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) c.b[s2] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(offB + 32u * s2), 64 * (int)n_kc, 0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 2; ++e) c.a[2 * s2 + e] = (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)(offA + (4u * s2 + e) * lda * 4u), 32 * (int)n_kc * (int)lda, 0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) c.sa[s2] = sa[(long long)(2 * n_kc + s2) * lda];
    c.cs = sb[(long long)li * nsb + 2 * n_kc + h];
    if (++n_kc == kchunks) { n_kc = 0; if (++n_r == p.br_count) { n_r = 0; if (++n_tile < ntiles) nx = mx4i8_tile(p, first + n_tile); } }
  };
  Mx4i8Chunk ring[D];
  static_for<D>([&](auto uc) { if ((unsigned long long)uc.value < total) request(ring[uc.value]); });
  Mx4i8Tile cur = mx4i8_tile(p, first);                            // the tile whose chunks are being multiplied
  unsigned int c_tile = 0;
  typedef float f32x2p __attribute__((ext_vector_type(2)));
  f32x2p facc2[8];
#pragma unroll
  for (int r2 = 0; r2 < 8; ++r2) facc2[r2] = (f32x2p)0.0f;
  unsigned long long left_in_tile = per_tile;
  for (unsigned long long f0 = 0; f0 < total; f0 += D) {
    static_for<D>([&](auto uc) {
      constexpr int u = uc.value;
      const unsigned long long f = f0 + u;
      if (f < total) {
        const Mx4i8Chunk c = ring[u];
        if (f + D < total) request(ring[u]);
        i32x4 af[2]; float rs[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          rs[s2] = __uint_as_float(c.sa[s2] << 23);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            af[s2][2 * e] = mx4_codes_to_bytes(c.a[2 * s2 + e] & 0x0f0f0f0fu);
            af[s2][2 * e + 1] = mx4_codes_to_bytes((c.a[2 * s2 + e] >> 4) & 0x0f0f0f0fu);
          }
        }
        lds_sb[32 * h + li] = c.cs;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          f32x4 cs[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) cs[g] = *(const f32x4*)(lds_sb + 32 * s2 + 8 * g + 4 * h);
          const i32x16 t = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, c.b[s2]), af[s2], (i32x16)0, 0, 0, 0);
          // two elements per instruction (v_pk_mul_f32 / v_pk_add_f32: each lane of the pair rounds like the scalar operation; contraction is off in this file)
          const f32x2p rs2 = {rs[s2], rs[s2]};
#pragma unroll
          for (int r2 = 0; r2 < 8; ++r2) {
            const f32x2p tv = {(float)t[2 * r2], (float)t[2 * r2 + 1]};
            const f32x2p c2 = {cs[r2 >> 1][2 * (r2 & 1)], cs[r2 >> 1][2 * (r2 & 1) + 1]};
            facc2[r2] = add_rn(facc2[r2], mul_rn(mul_rn(tv, rs2), c2));
          }
        }
        if (--left_in_tile == 0) {
          left_in_tile = per_tile;
          const unsigned int ldc = (unsigned int)p.ldc;
          float facc[16];
#pragma unroll
          for (int r2 = 0; r2 < 16; ++r2) facc[r2] = facc2[r2 >> 1][r2 & 1];
          if (beta0) {
            // column-major image [j][i] (lanes along i: conflict free), then whole columns: 64 (bf16) / 128 (f32) contiguous bytes from 4 / 8 lanes
            const __amdgpu_buffer_rsrc_t rc = wave_rsrc((gcptr)cur.q.c + ((unsigned long long)cur.j0 * ldc + (unsigned long long)cur.i0) * (c_f32 ? 4ull : 2ull));
            if (c_f32) {
#pragma unroll
              for (int r2 = 0; r2 < 16; ++r2) lds[li + jl_of(r2, h) * 32] = add_rn(0.0f, facc[r2]);
#pragma unroll
              for (int x = 0; x < 4; ++x) {
                const unsigned int L = (unsigned int)lane + 64u * x;
                __builtin_amdgcn_raw_buffer_store_b128(((const u32x4*)lds)[L], rc, (int)(((L >> 3) * ldc + (L & 7u) * 4u) * 4u), 0, 0);
              }
            } else {
              unsigned short* l16 = (unsigned short*)lds;
              unsigned int pk[8]; float yv[16];
#pragma unroll
              for (int r2 = 0; r2 < 16; ++r2) yv[r2] = add_rn(0.0f, facc[r2]);
              bf16_pk_exact_n<8>(yv, pk);
#pragma unroll
              for (int r2 = 0; r2 < 8; ++r2) { l16[li + jl_of(2 * r2, h) * 32] = (unsigned short)pk[r2]; l16[li + jl_of(2 * r2 + 1, h) * 32] = (unsigned short)(pk[r2] >> 16); }
#pragma unroll
              for (int x = 0; x < 2; ++x) {
                const unsigned int L = (unsigned int)lane + 64u * x;
                __builtin_amdgcn_raw_buffer_store_b128(((const u32x4*)lds)[L], rc, (int)(((L >> 2) * ldc + (L & 3u) * 8u) * 2u), 0, 0);
              }
            }
          } else {
            const unsigned long long e0 = (unsigned long long)(cur.j0 + 4 * h) * ldc + cur.i0 + li;
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
              const unsigned long long e = e0 + (unsigned long long)(((r2 & 3) + 8 * (r2 >> 2)) * p.ldc);
              if (c_f32) { GM float* cp = (GM float*)cur.q.c + e; *cp = add_rn(*cp, facc[r2]); }
              else { GM unsigned short* cp = (GM unsigned short*)cur.q.c + e; *cp = f32_to_bf16_rne(add_rn(bf16_to_f32(*cp), facc[r2])); }
            }
          }
#pragma unroll
          for (int r2 = 0; r2 < 8; ++r2) facc2[r2] = (f32x2p)0.0f;
          if (++c_tile < ntiles) cur = mx4i8_tile(p, first + c_tile);
        }
      }
    });
  }
}


// ------------------------------------------------------------------------------------------------
// 8-bit float streaming kernel (v_mfma_f32_32x32x16_bf8_bf8 / _fp8_fp8; CDNA4's fp8 = OCP E4M3 = the reference's HF8, bf8 =
// E5M2 = BF8): exact tiles, VNNI-4 A, flat B with 16-byte aligned columns, k % 64 == 0, f32 accumulate and output.
// Structure = gemm_i8_stream_kernel; an MFMA consumes 16 k (8 bytes per lane and operand), four steps per 64-deep chunk.
// ------------------------------------------------------------------------------------------------
// C8: C in the operands' 8-bit type [ref: gemm ref :2511-2619]: beta * C comes in through the type, the f32 sum leaves through the reference's two-step
// conversion f32 -> IEEE half (v_cvt_f16_f32: RNE; f32 denormals vanish either way) -> E5M2 / E4M3 (lowp.hpp, bit-identical to the reference's helpers).
// Round 4 (tools/fp8_cvt_probe.hip, all 65 536 halves): v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 fed with the half widened back to f32 give the reference's byte for EVERY
// half that is not a NaN -- ties, subnormal results, overflow (E4M3: 0x7f, E5M2: infinity) and infinities included; a NaN comes out with its sign bit set, the
// reference's without: those lanes take the software rounding.  Two results per conversion instruction instead of ~25 vector instructions per element.
template <int MT, int NT, bool HF8, bool C8 = false>
__global__ __launch_bounds__(256) void gemm_fp8_stream_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds_all[4][NT * 2048];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  char* lds = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
  static_for<MT * NT>([&](auto idx) {
    constexpr int mt = idx.value / NT, nt = idx.value % NT;
    tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h; tc[mt][nt].ivalid = true;
    if constexpr (C8) tile_init_c8<true>(acc[mt][nt], p, q, tc[mt][nt], HF8);
    else tile_init<true, true>(acc[mt][nt], p, q, tc[mt][nt]);
  });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  unsigned int offB[NT * 2], offBt[NT * 2];
#pragma unroll
  for (int x = 0; x < NT * 2; ++x) {
    const unsigned int L = (unsigned int)lane + 64u * x, f = L >> 2, pc = (L & 3u) ^ ((f >> 1) & 3u);
    offB[x] = f * ldb + pc * 16u;
    offBt[x] = f * ldb + (pc & 1u) * 16u;             // half a chunk (k % 64 == 32): the pieces 2 / 3 of a column re-read 0 / 1 and are never used
  }
  const unsigned int offA = ((2u * h) * lda + (unsigned int)li) * 4u;      // dword (k-quad 2h, row li)
  const int kchunks = p.k >> 6, ktail = (p.k >> 5) & 1;       // whole chunks, then one 32-deep half chunk: MFMA steps 0 and 1 only (round 3)
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br + (unsigned long long)job.j0 * ldb);     // buffer addressing, see gemm_bf16_stream_kernel
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + 4ull * (unsigned long long)job.i0);
    for (int kc = 0; kc < kchunks + ktail; ++kc) {
      const bool half = kc == kchunks;                 // wave-uniform
#pragma unroll
      for (int x = 0; x < NT * 2; ++x)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(lds + 1024 * x), 16, (int)(half ? offBt[x] : offB[x]), 64 * kc, 0, 0);
      long af[MT][4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned int sq = (s >= 2 && half) ? 4u * (s - 2) : 4u * s;      // in a half chunk the steps 2 / 3 repeat 0 / 1 (in bounds) and are not multiplied
          const unsigned int lo = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(offA + sq * lda * 4u) + 128 * mt, 64 * kc * (int)lda, 0);
          const unsigned int hi = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(offA + (sq + 1u) * lda * 4u) + 128 * mt, 64 * kc * (int)lda, 0);
          af[mt][s] = (long)(((unsigned long long)hi << 32) | lo);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      long bfr[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int f = 32 * nt + li;
          bfr[nt][s] = *(const long*)(lds + f * 64 + ((s ^ ((f >> 1) & 3)) * 16) + 8 * h);
        }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s == 2 && half) break;
        static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
          if (HF8) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bfr[nt][s], af[mt][s], acc[mt][nt], 0, 0, 0);
          else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(bfr[nt][s], af[mt][s], acc[mt][nt], 0, 0, 0); });
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if constexpr (C8) {
    // byte results through the wave's LDS image [32 NT columns][32 MT rows] and out as 16-byte pieces (64 byte stores per lane measured 0.27 of the roofline on
    // 64^3 bf8 problems: the store instructions, not the bytes, were the bound); columns that are not 16-byte aligned in memory leave byte by byte
    constexpr unsigned int pitch = 32u * MT;
    static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
      tile_activate<true>(acc[mt][nt], p, q, tc[mt][nt]);          // fused ReLU (+ bitmask) / sigmoid (round 6)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const unsigned int two = f32x2_to_fp8_ref(acc[mt][nt][r], acc[mt][nt][r + 1], HF8);
        ((unsigned char*)lds)[(32u * nt + (unsigned int)jl_of(r, h)) * pitch + 32u * mt + (unsigned int)li] = (unsigned char)two;
        ((unsigned char*)lds)[(32u * nt + (unsigned int)jl_of(r + 1, h)) * pitch + 32u * mt + (unsigned int)li] = (unsigned char)(two >> 8);
      } });
    GM unsigned char* ct = (GM unsigned char*)q.c + (long long)job.j0 * p.ldc + job.i0;
    const bool wide = ((((unsigned long long)(size_t)ct) | (unsigned long long)p.ldc) & 15ull) == 0ull;        // wave-uniform
    if (wide) {
#pragma unroll
      for (int x = 0; x < MT * NT; ++x) {
        const unsigned int P = (unsigned int)lane + 64u * x, j = P / (pitch / 16u), c16 = P % (pitch / 16u);
        *(GM u32x4*)(ct + (long long)j * p.ldc + c16 * 16u) = *(const u32x4*)(lds + j * pitch + c16 * 16u);
      }
    } else {
      static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const unsigned int j = 32u * nt + (unsigned int)jl_of(r, h), i = 32u * mt + (unsigned int)li; ct[(long long)j * p.ldc + i] = ((const unsigned char*)lds)[j * pitch + i]; } });
    }
  } else {
    static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<true, true>(acc[mt][nt], p, q, tc[mt][nt]); });
  }
}

// ------------------------------------------------------------------------------------------------
// 8-bit operands of ANY shape on the matrix cores (round 4): what the two streaming kernels above refuse -- m / n that are not whole tiles (40^3 ...),
// k % 32 != 0 (k % 4 == 0: a VNNI-4 quad of A is whole), operands that are not 16-byte aligned, pointer and offset lists.  These ran on the
// one-element-per-thread kernel (0.006 / 0.02 of the HBM roofline on 40^3 problems).
// KIND 0: u8 / i8 -> i32 on v_mfma_i32_32x32x32_i8 (integer sums: bit-exact in any order), 1: BF8, 2: HF8 on v_mfma_f32_32x32x16 (f32 sums).
// Per 32-deep chunk a lane fetches four dwords (= four k-quads) of its row of A -- lanes run along i, so a load instruction covers whole 128-byte
// rows of the VNNI-4 image -- and four dwords of its column of B.  A quad at or beyond k, a row beyond m, a column beyond n reads as ZERO, and the
// unsigned -> signed shift of the integer kernel (see gemm_i8_stream_kernel) is applied to real elements only, so padding adds nothing to any sum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int load_u32_any(gcptr p4) {          // a dword at any byte alignment
  if ((((unsigned long long)(size_t)p4) & 3ull) == 0ull) return *(GM const unsigned int*)p4;
  GM const unsigned char* b = (GM const unsigned char*)p4;
  return (unsigned int)b[0] | ((unsigned int)b[1] << 8) | ((unsigned int)b[2] << 16) | ((unsigned int)b[3] << 24);
}
// The unsigned -> signed correction terms 128 * sum_k b'(j, k) (unsigned A) and 128 * sum_k a'(i, k) (unsigned B) are plain byte sums of what a lane holds anyway
// (lane = column j of B, lane = row i of A): one v_dot4_i32_i8 against 0x01010101 per operand dword into ONE register per tile row / column, the two k halves of
// the wave added at the end; the column sums reach the accumulator layout (column = register) through 16 ds_bpermute per column tile in the epilogue.  (The
// streaming kernel spends an MFMA and 16 accumulators per tile row / column on them: here that cost a wave per SIMD -- 216 against 152 registers.)
// one 32-deep chunk of products of the masked 8-bit kernel: operand dwords aw / bw (padding already zero) into the accumulators, the byte sums of the unsigned forms
template <int MT, int NT, int KIND, bool UA, bool UB, bool BL = false>
__global__ __launch_bounds__(256, 3) void gemm_mfma_8bit_kernel(GemmArgs p) {
  constexpr bool INT = KIND == 0, HF8 = KIND == 2;
  __shared__ __attribute__((aligned(16))) char lds_img[4][BL ? NT * 1024 : 16];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const bool c8 = !INT && p.c_type != LIBXSMM_DATATYPE_F32;          // 8-bit float C: wave-uniform
  i32x16 iacc[INT ? MT : 1][INT ? NT : 1];
  int sum_b[NT], sum_a[MT];                                          // this lane's column j / row i, its k half
  f32x16 facc[INT ? 1 : MT][INT ? 1 : NT];
  TileCtx tc[MT][NT];
  static_for<MT * NT>([&](auto idx) {
    constexpr int mt = idx.value / NT, nt = idx.value % NT;
    tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h; tc[mt][nt].ivalid = tc[mt][nt].i < p.m;
    if constexpr (INT) iacc[mt][nt] = (i32x16)0;
    else {
      if (c8) tile_init_c8<false>(facc[mt][nt], p, q, tc[mt][nt], HF8);
      else tile_init<false, true>(facc[mt][nt], p, q, tc[mt][nt]);
    }
  });
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) sum_b[nt] = 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) sum_a[mt] = 0;
  const int kquads = p.k >> 2, kchunks = (p.k + 31) >> 5;
  // Dword-aligned operands (the usual case): every load of a chunk is issued UNCONDITIONALLY -- a lane beyond m / n reads the last real row / column, a quad beyond k the
  // last real quad: the same addresses its neighbours request, so no byte more is fetched -- and the padding is a select afterwards.  With a branch around every load
  // (the first form) the compiler waited for each of the sixteen before it issued the next.  (An earlier branch-free attempt clamped to row / column 0 instead: other
  // cache lines, 0.51 -> 0.24.)  Misaligned operands assemble their dwords byte by byte.
  long long aoff[MT], boff[NT]; bool iok[MT], jok[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { const int i = job.i0 + 32 * mt + li; iok[mt] = i < p.m; aoff[mt] = 4ll * (iok[mt] ? i : p.m - 1); }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { const int j = job.j0 + 32 * nt + li; jok[nt] = j < p.n; boff[nt] = (long long)(jok[nt] ? j : p.n - 1) * p.ldb; }
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    if constexpr (BL) {
      char* image = lds_img[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
      const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar), rb = wave_rsrc(br);
      const unsigned int dq = (unsigned int)lane & 7u, fb = (unsigned int)lane >> 3;
      for (int kc = 0; kc < kchunks; ++kc) {
        unsigned int aw[MT][4], bw[NT][4];
#pragma unroll
        for (int x = 0; x < NT * 4; ++x) {                      // LDS slot lane + 64 x: column fb + 8 x, k-quad dq of the chunk
          const int jc = job.j0 + (int)fb + 8 * x, kq = 8 * kc + (int)dq;
          const unsigned int voff = (unsigned int)(jc < p.n ? jc : p.n - 1) * (unsigned int)p.ldb + 4u * (unsigned int)(kq < kquads ? kq : kquads - 1);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(image + 256 * x), 4, (int)voff, 0, 0, 0);
        }
        bool kok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kq = INT ? 8 * kc + 4 * h + e : 8 * kc + 4 * (e >> 1) + 2 * h + (e & 1);
          kok[e] = kq < kquads;
          const unsigned int krow = (unsigned int)(kok[e] ? kq : kquads - 1) * (unsigned int)p.lda * 4u;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) aw[mt][e] = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(krow + (unsigned int)aoff[mt]), 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const char* col = image + (32 * nt + li) * 32;
          if constexpr (INT) { const u32x4 t = *(const u32x4*)(col + 16 * h); bw[nt][0] = t[0]; bw[nt][1] = t[1]; bw[nt][2] = t[2]; bw[nt][3] = t[3]; }
          else {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { const u32x2 t = *(const u32x2*)(col + 16 * s2 + 8 * h); bw[nt][2 * s2] = t[0]; bw[nt][2 * s2 + 1] = t[1]; }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) aw[mt][e] = (kok[e] && iok[mt]) ? ((INT && UA) ? (aw[mt][e] ^ 0x80808080u) : aw[mt][e]) : 0u;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bw[nt][e] = (kok[e] && jok[nt]) ? ((INT && UB) ? (bw[nt][e] ^ 0x80808080u) : bw[nt][e]) : 0u;
        }
        m8_products<MT, NT, KIND, UA, UB>(aw, bw, iacc, facc, sum_a, sum_b);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // LDS reads retired before the image is refilled
      }
      continue;
    }
    bool b_al = true;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_al = b_al && ((((unsigned long long)(size_t)(br + boff[nt])) & 3ull) == 0ull);
    // (measured on 40^3 problems, branch per load -> unconditional loads: bf8 0.29 -> 0.47, u8 x i8 0.34 -> 0.42, hf8 -> hf8 0.15 -> 0.22, but i8 x i8 0.51 -> 0.45: the
    //  signed-signed variant has no correction code and fits four waves per SIMD only in the branchy form (120 against 140 registers) -- it keeps that form)
    constexpr bool CLAMPED = !(INT && !UA && !UB);
    const bool fast = CLAMPED && ((((unsigned long long)(size_t)ar) & 3ull) == 0ull) && (__builtin_amdgcn_ballot_w64(!b_al) == 0ull);       // wave-uniform
    for (int kc = 0; kc < kchunks; ++kc) {
      unsigned int aw[MT][4], bw[NT][4];
      if (fast) {
        bool kok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kq = INT ? 8 * kc + 4 * h + e : 8 * kc + 4 * (e >> 1) + 2 * h + (e & 1);       // the k-quad of operand dword e (fp8: MFMA step e / 2)
          kok[e] = kq < kquads;
          const long long kqs = kok[e] ? (long long)kq : (long long)(kquads - 1);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) aw[mt][e] = *(GM const unsigned int*)(ar + kqs * p.lda * 4 + aoff[mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bw[nt][e] = *(GM const unsigned int*)(br + boff[nt] + 4ll * kqs);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) aw[mt][e] = (kok[e] && iok[mt]) ? ((INT && UA) ? (aw[mt][e] ^ 0x80808080u) : aw[mt][e]) : 0u;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bw[nt][e] = (kok[e] && jok[nt]) ? ((INT && UB) ? (bw[nt][e] ^ 0x80808080u) : bw[nt][e]) : 0u;
        }
      } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kq = INT ? 8 * kc + 4 * h + e : 8 * kc + 4 * (e >> 1) + 2 * h + (e & 1);
        const bool kok = kq < kquads;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          unsigned int v = 0u;
          if (kok && iok[mt]) { v = load_u32_any(ar + (long long)kq * p.lda * 4 + aoff[mt]); if (INT && UA) v ^= 0x80808080u; }
          aw[mt][e] = v;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          unsigned int v = 0u;
          if (kok && jok[nt]) { v = load_u32_any(br + boff[nt] + 4ll * kq); if (INT && UB) v ^= 0x80808080u; }
          bw[nt][e] = v;
        }
      }
      }
      m8_products<MT, NT, KIND, UA, UB>(aw, bw, iacc, facc, sum_a, sum_b);
    }
  }
  if constexpr (INT) {
    const bool c_f32 = p.c_type == LIBXSMM_DATATYPE_F32;
    const int kconst = (UA && UB) ? (int)(16384u * (unsigned int)(p.br_count * (unsigned long long)p.k)) : 0;      // (mod 2^32 like the i32 sums themselves: unsigned arithmetic, no signed overflow)
    if constexpr (UA) {       // both k halves of a column: lane j + lane j + 32
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sum_b[nt] += __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), sum_b[nt]);
    }
    if constexpr (UB) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) sum_a[mt] += __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), sum_a[mt]);
    }
    static_for<MT * NT>([&](auto idx) {
      constexpr int mt = idx.value / NT, nt = idx.value % NT;
      const TileCtx& t = tc[mt][nt];
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) {
        const int jl = jl_of(r2, h), j = t.j0 + jl;
        int v = iacc[mt][nt][r2] + kconst;
        if constexpr (UA) v += 128 * __builtin_amdgcn_ds_bpermute(4 * jl, sum_b[nt]);       // the sum of column jl lives in lane jl (all lanes execute the exchange)
        if constexpr (UB) v += 128 * sum_a[mt];
        if (!(t.ivalid && j < p.n)) continue;
        GM char* cp = (GM char*)q.c + 4ll * ((long long)j * p.ldc + t.i);
        if (c_f32) { float f = mul_rn((float)v, p.scf); if (!beta0) f = add_rn(f, *(GM const float*)cp); *(GM float*)cp = f; }
        else { if (!beta0) v += *(GM const int*)cp; *(GM int*)cp = v; }
      }
    });
  } else if (c8) {
    static_for<MT * NT>([&](auto idx) {
      constexpr int mt = idx.value / NT, nt = idx.value % NT;
      const TileCtx& t = tc[mt][nt];
      tile_activate<false>(facc[mt][nt], p, q, t);               // fused ReLU (+ bitmask) / sigmoid (round 6)
#pragma unroll
      for (int r2 = 0; r2 < 16; r2 += 2) {
        const unsigned int two = f32x2_to_fp8_ref(facc[mt][nt][r2], facc[mt][nt][r2 + 1], HF8);
        const int j = t.j0 + jl_of(r2, h);                                                                       // registers (2q, 2q + 1) are columns (j, j + 1)
        if (t.ivalid && j < p.n) ((GM unsigned char*)q.c)[(long long)j * p.ldc + t.i] = (unsigned char)two;
        if (t.ivalid && j + 1 < p.n) ((GM unsigned char*)q.c)[(long long)(j + 1) * p.ldc + t.i] = (unsigned char)(two >> 8);
      }
    });
  } else {
    // (plain stores: the columns of a ragged C are pieces of cache lines, which non-temporal stores would send to memory one by one)
    static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<false, true, false>(facc[mt][nt], p, q, tc[mt][nt]); });
  }
}

// ------------------------------------------------------------------------------------------------
// MXFP4 weight streaming kernel: A = packed E2M1 pairs + one E8M0 scale per (32-deep k-block, row), B bf16, exact tiles.
// Structure = gemm_bf16_stream_kernel (B through LDS-DMA).  In this operand order a lane owns ONE row i of A, so the scale
// of a k-block is one value per lane: v_cvt_scalef32_pk_bf16_fp4 turns a byte (two k of row i) into a scaled bf16 pair
// in one instruction (value * 2^e is exact in bf16), i.e. the MFMA consumes exactly what the reference multiplies
// [ref: gemm ref :976-982]; only the summation order differs (matrix-core tree instead of a serial chain).
// HBM bytes per 64^3 problem: A 2 KiB + scales 128 B instead of 8 KiB of bf16 weights.
// ------------------------------------------------------------------------------------------------
template <int MT, int NT>
__global__ __launch_bounds__(256) void gemm_mxfp4_stream_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds_all[4][NT * 2048];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  char* lds = lds_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
  static_for<MT * NT>([&](auto idx) {
    constexpr int mt = idx.value / NT, nt = idx.value % NT;
    tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h; tc[mt][nt].ivalid = true;
    tile_init<true, false>(acc[mt][nt], p, q, tc[mt][nt]);
  });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  unsigned int offB[NT * 2];
#pragma unroll
  for (int x = 0; x < NT * 2; ++x) {
    const unsigned int L = (unsigned int)lane + 64u * x, f = L >> 2, pc = (L & 3u) ^ ((f >> 1) & 3u);
    offB[x] = (f * ldb) * 2u + pc * 16u;
  }
  // Buffer addressing: every base is wave-uniform (a 4-SGPR resource per operand and batch-reduce element), lanes contribute
  // loop-invariant 32-bit offsets and the k-chunk is a scalar offset -- no 64-bit address VGPRs, no per-iteration address math.
  unsigned int voffA[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) voffA[s][e] = (8u * h + 4u * s + e) * lda + (unsigned int)li;   // k-pair row 8h + 4s + e: k = 16h + 8s + 2e, as B's pieces
  const int kchunks = p.k >> 5;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t rb = wave_rsrc(br + 2ull * (unsigned long long)job.j0 * ldb);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + (unsigned long long)job.i0);
    const __amdgpu_buffer_rsrc_t rs = wave_rsrc(mx_scale_base(p, job.bidx, r, false) + (unsigned long long)job.i0);
    for (int kc = 0; kc < kchunks; ++kc) {
#pragma unroll
      for (int x = 0; x < NT * 2; ++x)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(lds + 1024 * x), 16, (int)offB[x], 64 * kc, 0, 0);
      unsigned int raw[MT][2][4], sc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) sc[mt] = __builtin_amdgcn_raw_buffer_load_b8(rs, li + 32 * mt, kc * (int)lda, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            raw[mt][s][e] = __builtin_amdgcn_raw_buffer_load_b8(ra, (int)voffA[s][e] + 32 * mt, 16 * kc * (int)lda, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      u32x4 af[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float scale = __uint_as_float(sc[mt] << 23);                    // the instruction reads only the exponent field: byte 0 acts as 2^-127
                                                                              // (reference: 0.0f) -- at most 6 * 2^-127 per weight, far below f32 resolution
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            af[mt][s][e] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(raw[mt][s][e], scale, 0));
      }
      u32x4 bfr[NT][2];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int f = 32 * nt + li;
          bfr[nt][s] = *(const u32x4*)(lds + f * 64 + (((2 * h + s) ^ ((f >> 1) & 3)) * 16));
        }
#pragma unroll
      for (int s = 0; s < 2; ++s)
        static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[nt][s]), __builtin_bit_cast(bf16x8, af[mt][s]), acc[mt][nt], 0, 0, 0); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<true, false>(acc[mt][nt], p, q, tc[mt][nt]); });
}

// ------------------------------------------------------------------------------------------------
// MX x MX streaming kernel (v_mfma_scale_f32_32x32x64_f8f6f4): the OCP microscaling GEMM is what this instruction computes -- a lane
// supplies 32 consecutive k of ONE row (lane & 31) of its operand for the k-half (lane >> 5), plus that block's E8M0 scale byte; the
// matrix core applies 2^(sa-127) * 2^(sb-127) to the block's partial product.  A and B arrive in the reference's k-grouped layout
// (a dword per row and k-group: 4 x fp8 or 8 x fp4), which is already coalesced along the rows: both go straight to operand
// registers through buffer loads, no LDS.  FMT: 0 = E4M3 (MXHF8), 1 = E5M2 (MXBF8), 4 = E2M1 (MXFP4).  Exact tiles, k % 64 == 0.
// HBM bytes per 64^3 fp4 problem: 2 x 2 KiB operands + 256 B of scales + 16 KiB of f32 C.
// ------------------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
// Round 3: FMT 2 = E2M3 (MXHF6): a 6-bit format of the same instruction (six operand registers per lane; lane half h holds block h -- measured, FP6MAP 1 = the
// 8-bit convention gives wrong sums).  FMT 3 = E3M2 (MXBF6) works too but exceeds the reference driver's error bound (see plan_gemm) and is not dispatched.
template <int MT, int NT, int FMT, int FP6MAP = 0>
__global__ __launch_bounds__(256) void gemm_mx_stream_kernel(GemmArgs p) {
  constexpr bool FP6 = FMT == 2 || FMT == 3;
  constexpr int NDW = (FMT == 4) ? 4 : 8;                  // dwords (k-groups) per lane and 64-deep step
  constexpr int kImgA = FP6 ? ((MT * 1536 + 1023) / 1024) * 1024 : 0, kImgB = FP6 ? ((NT * 1536 + 1023) / 1024) * 1024 : 0;     // 6-bit operand images of a chunk, whole requests
  __shared__ __attribute__((aligned(16))) char lds6_all[4][kImgA + kImgB + 16];
  const WaveJob job = wave_job(p, 32 * MT, 32 * NT);
  if (!job.active) return;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  char* lds6 = lds6_all[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const BatchPtrs q = batch_ptrs(p, job.bidx);
  f32x16 acc[MT][NT];
  TileCtx tc[MT][NT];
  static_for<MT * NT>([&](auto idx) {
    constexpr int mt = idx.value / NT, nt = idx.value % NT;
    tc[mt][nt].i = job.i0 + 32 * mt + li; tc[mt][nt].j0 = job.j0 + 32 * nt; tc[mt][nt].h = h; tc[mt][nt].ivalid = true;
    tile_init<true, true>(acc[mt][nt], p, q, tc[mt][nt]);
  });
  const unsigned int lda = (unsigned int)p.lda, ldb = (unsigned int)p.ldb;
  unsigned int voffA[NDW], voffB[NDW];
#pragma unroll
  for (int e = 0; e < NDW; ++e) {
    // k-group (one dword) held in operand register e of this lane.  4-bit: the lane half h owns the whole 32-deep block h (4 dwords).
    // 8-bit: registers 0-3 belong to block 0 and 4-7 to block 1, each half of the wave holding 16 k of either block (measured: with
    // per-block scales, "all 32 k of block h in lane half h" gives wrong sums); the scale of block b still comes from lane half b.
    const unsigned int kg = (FMT == 4) ? (unsigned int)(4 * h + e) : (unsigned int)(8 * (e >> 2) + 4 * h + (e & 3));
    voffA[e] = (kg * lda + (unsigned int)li) * 4u; voffB[e] = (kg * ldb + (unsigned int)li) * 4u;
  }
  const int ksteps = p.k >> 6;
  for (unsigned long long r = 0; r < p.br_count; ++r) {
    gcptr ar, br; br_base(p, q, r, ar, br);
    const __amdgpu_buffer_rsrc_t ra = wave_rsrc(ar + 4ull * (unsigned long long)job.i0), rb = wave_rsrc(br + 4ull * (unsigned long long)job.j0);
    const __amdgpu_buffer_rsrc_t rsa = wave_rsrc(mx_scale_base(p, job.bidx, r, false) + (unsigned long long)job.i0);
    const __amdgpu_buffer_rsrc_t rsb = wave_rsrc(mx_scale_base(p, job.bidx, r, true) + (unsigned long long)job.j0);
    // 6-bit formats: resources that END with the operand (reads past it return zero): the 3-byte groups are fetched as the two dwords around them
    __amdgpu_buffer_rsrc_t ra6 = ra, rb6 = rb;
    if constexpr (FP6) {
      const long long bytes_a = (long long)(p.k / 4) * lda * 3 - 3ll * job.i0, bytes_b = (long long)(p.k / 4) * ldb * 3 - 3ll * job.j0;
      ra6 = __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)uniform_u64((unsigned long long)(size_t)(ar + 3ull * (unsigned long long)job.i0)), (short)0, (int)bytes_a, 0x00020000);
      rb6 = __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)uniform_u64((unsigned long long)(size_t)(br + 3ull * (unsigned long long)job.j0)), (short)0, (int)bytes_b, 0x00020000);
    }
    const bool stage6 = FP6 && !(p.tune & 1) && ((uniform_u64((unsigned long long)(size_t)ar) | uniform_u64((unsigned long long)(size_t)br)) & 15ull) == 0 && (lda & 15u) == 0 && (ldb & 15u) == 0;
    for (int kc = 0; kc < ksteps; ++kc) {
      i32x8 af[MT], bf[NT];
      int sa[MT], sb[NT];
      if constexpr (FP6) {
        // [k/4][ld][3 bytes]: four 6-bit values of a row per k-group, little-endian and dense -- the eight groups of a 32-deep block in k order ARE the
        // 192-bit operand image of that block.  Register e of the operand (6 used): FP6MAP 0: lane half h holds block h (groups 8 h ..); FP6MAP 1: registers
        // 0-2 = block 0, 3-5 = block 1, each lane half 16 k (four groups) of either block (the 8-bit convention).
        auto fetch6 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned int ld, int tile, i32x8& out) {
          unsigned int v[8];
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const unsigned int kg = (FP6MAP == 0) ? (unsigned int)(8 * h + g) : (unsigned int)(8 * (g >> 2) + 4 * h + (g & 3));
            const unsigned int off = ((kg + 16u * (unsigned int)kc) * ld + (unsigned int)(li + 32 * tile)) * 3u;
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t w2 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(off & ~3u), 0, 0);          // the two dwords around the 3-byte group (dword alignment is all a buffer load needs)
            v[g] = (unsigned int)(((((unsigned long long)w2[1]) << 32) | w2[0]) >> (8u * (off & 3u))) & 0x00ffffffu;
          }
          out[0] = (int)(v[0] | (v[1] << 24)); out[1] = (int)((v[1] >> 8) | (v[2] << 16)); out[2] = (int)((v[2] >> 16) | (v[3] << 8));
          out[3] = (int)(v[4] | (v[5] << 24)); out[4] = (int)((v[5] >> 8) | (v[6] << 16)); out[5] = (int)((v[6] >> 16) | (v[7] << 8));
          out[6] = 0; out[7] = 0;
        };
        // Round 3, second step: with 16-byte aligned operands the 16 groups x (32 MT rows x 3 bytes) of a chunk travel global -> LDS as whole 16-byte pieces
        // (3 requests per 64-row operand instead of 16 two-dword gathers per lane) into a packed image [group][row][3 bytes]; a lane then picks its groups
        // out of the image with the same two-dwords-around-the-group reads.
        if (stage6) {
          auto dma6 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned int ld, auto tiles_c, char* img) __attribute__((always_inline)) {
            constexpr int T = decltype(tiles_c)::value, P = 6 * T, NI6 = (T * 1536 + 1023) / 1024;
#pragma unroll
            for (int x = 0; x < NI6; ++x) {
              const unsigned int Lx = (unsigned int)lane + 64u * x, g = Lx / (unsigned int)P, piece = Lx - g * (unsigned int)P;
              const unsigned int voff = g < 16u ? g * ld * 3u + piece * 16u : 0x7ffffff0u;                   // past the image: out of bounds, reads as zero
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(img + 1024 * x), 16, (int)voff, 16 * kc * (int)ld * 3, 0, 0);
            }
          };
          auto pick6 = [&](const char* img, int rows_bytes, int tile, i32x8& out) __attribute__((always_inline)) {
            unsigned int v[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
              const unsigned int off = (unsigned int)((8 * h + g) * rows_bytes + (li + 32 * tile) * 3);
              const unsigned int* w = (const unsigned int*)(img + (off & ~3u));
              v[g] = (unsigned int)(((((unsigned long long)w[1]) << 32) | w[0]) >> (8u * (off & 3u))) & 0x00ffffffu;
            }
            out[0] = (int)(v[0] | (v[1] << 24)); out[1] = (int)((v[1] >> 8) | (v[2] << 16)); out[2] = (int)((v[2] >> 16) | (v[3] << 8));
            out[3] = (int)(v[4] | (v[5] << 24)); out[4] = (int)((v[5] >> 8) | (v[6] << 16)); out[5] = (int)((v[6] >> 16) | (v[7] << 8));
            out[6] = 0; out[7] = 0;
          };
          static_assert(FP6MAP == 0, "the staged form is written for lane half h = block h");
          dma6(ra6, lda, std::integral_constant<int, MT>{}, lds6);
          dma6(rb6, ldb, std::integral_constant<int, NT>{}, lds6 + kImgA);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) sa[mt] = (int)__builtin_amdgcn_raw_buffer_load_b8(rsa, h * (int)lda + li + 32 * mt, 2 * kc * (int)lda, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) sb[nt] = (int)__builtin_amdgcn_raw_buffer_load_b8(rsb, h * (int)ldb + li + 32 * nt, 2 * kc * (int)ldb, 0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pick6(lds6, 96 * MT, mt, af[mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) pick6(lds6 + kImgA, 96 * NT, nt, bf[nt]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the image is free for the next chunk's requests
        } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { sa[mt] = (int)__builtin_amdgcn_raw_buffer_load_b8(rsa, h * (int)lda + li + 32 * mt, 2 * kc * (int)lda, 0); fetch6(ra6, lda, mt, af[mt]); }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { sb[nt] = (int)__builtin_amdgcn_raw_buffer_load_b8(rsb, h * (int)ldb + li + 32 * nt, 2 * kc * (int)ldb, 0); fetch6(rb6, ldb, nt, bf[nt]); }
        }
      } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        sa[mt] = (int)__builtin_amdgcn_raw_buffer_load_b8(rsa, h * (int)lda + li + 32 * mt, 2 * kc * (int)lda, 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) af[mt][e] = (e < NDW) ? (int)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)voffA[e < NDW ? e : 0] + 128 * mt, 2 * NDW * kc * (int)lda * 4, 0) : 0;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        sb[nt] = (int)__builtin_amdgcn_raw_buffer_load_b8(rsb, h * (int)ldb + li + 32 * nt, 2 * kc * (int)ldb, 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[nt][e] = (e < NDW) ? (int)__builtin_amdgcn_raw_buffer_load_b32(rb, (int)voffB[e < NDW ? e : 0] + 128 * nt, 2 * NDW * kc * (int)ldb * 4, 0) : 0;
      }
      }
      static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[nt], af[mt], acc[mt][nt], FMT, FMT, 0, sb[nt], 0, sa[mt]); });
    }
  }
  static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT; tile_store<true, true, (MT * NT == 1)>(acc[mt][nt], p, q, tc[mt][nt]); });
}

// ------------------------------------------------------------------------------------------------
// host-side selection
// ------------------------------------------------------------------------------------------------
#if defined(XAMD_GEMM_SHARD)
// A shard translation unit (Makefile: gemm_shard_<i>.o, tools/gen_gemm_shards.py): the kernel definitions above + the explicit instantiations of this shard, nothing else.
#include XAMD_GEMM_SHARD_INC
}  // namespace xamd
#else
// The main translation unit generates no code for the kernel templates above: every instantiation launch_gemm uses is declared extern here and defined in a shard.
#if !defined(XAMD_GEMM_MONO)
#include "gemm_shards/extern.inc"
#endif

// the fused operators of the ext ABI this library implements on every type the reference's loop fuses them on (f32, BF32, bf16, IEEE halves, 8-bit floats):
// column bias (binary ADD, broadcast column), ReLU (+ bitmask) / sigmoid on C [ref: gemm ref :294-372; the accept list of libxsmm_dispatch_brgemm_ext, runtime.cpp]
static bool fused_ops_ok(const libxsmm_gemm_descriptor& d) {
  const bool bin_ok = d.bin_type == LIBXSMM_MELTW_TYPE_BINARY_NONE ||
    (d.bin_type == LIBXSMM_MELTW_TYPE_BINARY_ADD && (d.bin_flags & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0 | LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_1)));
  const bool cp_ok = d.cp_type == LIBXSMM_MELTW_TYPE_UNARY_NONE || d.cp_type == LIBXSMM_MELTW_TYPE_UNARY_RELU || d.cp_type == LIBXSMM_MELTW_TYPE_UNARY_SIGMOID;
  if (!bin_ok || !cp_ok || d.ap_type != 0 || d.bp_type != 0) return false;
  return (d.flags & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI) || (d.bin_type == 0 && d.cp_type == 0);
}

bool gemm_supported(const libxsmm_gemm_descriptor& d_in) {
  libxsmm_gemm_descriptor d = d_in;
  if (d.flags & LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK) {
    // A arrives as (non-zeros, one bit per element) [ref: gemm ref :857-948]: one plain GEMM (no batch-reduce, no transposes, no fused
    // operators), f32 or 16-bit operands; the 16-bit image is the VNNI-2 one whatever the flag says, and its leading dimension is m.
    // Expanded on the device into a dense image, then the dense kernel of the equivalent descriptor runs.
    const bool f32x = d.a_type == LIBXSMM_DATATYPE_F32 && d.b_type == LIBXSMM_DATATYPE_F32 && d.c_type == LIBXSMM_DATATYPE_F32;
    const bool h16 = (d.a_type == LIBXSMM_DATATYPE_BF16 || d.a_type == LIBXSMM_DATATYPE_F16) && d.b_type == d.a_type && (d.c_type == d.a_type || d.c_type == LIBXSMM_DATATYPE_F32);
    if (!f32x && !h16) return false;
    if (d.flags & (LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE |
                   LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C)) return false;
    const int kb = f32x ? 1 : 2;
    if ((d.k % kb) != 0 || (((long long)d.m * kb) & 7) != 0 || (long long)d.m * d.k >= (1ll << 31)) return false;
    d.flags &= ~(unsigned int)LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK;
    if (h16) d.flags |= LIBXSMM_GEMM_FLAG_VNNI_A;
    d.lda = d.m;
  }
  const bool f32 = d.a_type == LIBXSMM_DATATYPE_F32 && d.b_type == LIBXSMM_DATATYPE_F32 && d.c_type == LIBXSMM_DATATYPE_F32;
  const bool f64 = d.a_type == LIBXSMM_DATATYPE_F64 && d.b_type == LIBXSMM_DATATYPE_F64 && d.c_type == LIBXSMM_DATATYPE_F64;
  const bool bf16 = d.a_type == LIBXSMM_DATATYPE_BF16 && d.b_type == LIBXSMM_DATATYPE_BF16 &&
                    (d.c_type == LIBXSMM_DATATYPE_F32 || d.c_type == LIBXSMM_DATATYPE_BF16);
  const bool i8 = (d.a_type == LIBXSMM_DATATYPE_I8 || d.a_type == LIBXSMM_DATATYPE_U8) && (d.b_type == LIBXSMM_DATATYPE_I8 || d.b_type == LIBXSMM_DATATYPE_U8) &&
                  (d.c_type == LIBXSMM_DATATYPE_I32 || d.c_type == LIBXSMM_DATATYPE_F32);
  if ((d.a_type == LIBXSMM_DATATYPE_I1X8 || d.a_type == LIBXSMM_DATATYPE_I2X4) && (d.b_type == LIBXSMM_DATATYPE_I8 || d.b_type == LIBXSMM_DATATYPE_U8)) {
    // [ref: gemm ref :480-486, :1100-1300]: packed 1- / 2-bit weights (VNNI; the 2-bit form interleaved) x 8-bit activations -> i32
    const unsigned int fl = d.flags;
    if (d.c_type != LIBXSMM_DATATYPE_I32 || d.comp_type != LIBXSMM_DATATYPE_I32 || !(fl & LIBXSMM_GEMM_FLAG_VNNI_A)) return false;
    if (((fl & LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT) != 0) != (d.a_type == LIBXSMM_DATATYPE_I2X4)) return false;
    if (fl & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C | LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI)) return false;
    if ((d.k & 3) || (d.ldb & 3) || (d.m % (d.a_type == LIBXSMM_DATATYPE_I2X4 ? 4 : 2)) || (d.lda & (d.a_type == LIBXSMM_DATATYPE_I2X4 ? 3 : 1))) return false;
    if (d.bin_type != 0 || d.cp_type != 0 || d.ap_type != 0 || d.bp_type != 0) return false;
    return d.lda >= d.m && d.ldb >= d.k && d.ldc >= d.m;
  }
  {   // the dense loop in further types: all on the exact generic kernel, plain ABI, no fused operators
    const unsigned int fl = d.flags;
    const bool plain = !(fl & (LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI | LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C)) && d.bin_type == 0 && d.cp_type == 0 && d.ap_type == 0 && d.bp_type == 0;
    const bool ta_ = fl & LIBXSMM_GEMM_FLAG_TRANS_A, tb_ = fl & LIBXSMM_GEMM_FLAG_TRANS_B, va_ = fl & LIBXSMM_GEMM_FLAG_VNNI_A;
    const bool lds_ok = (ta_ ? d.lda >= d.k : d.lda >= d.m) && (tb_ ? d.ldb >= d.n : d.ldb >= d.k) && d.ldc >= d.m;
    if (d.a_type == LIBXSMM_DATATYPE_BF32 || d.b_type == LIBXSMM_DATATYPE_BF32)      // f32 storage, operands rounded to bf16 [ref: gemm ref :1359-1426]; fused operators as f32 (round 6)
      return d.a_type == d.b_type && d.c_type == LIBXSMM_DATATYPE_F32 && d.comp_type == LIBXSMM_DATATYPE_F32 && !(fl & (LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C)) && fused_ops_ok(d) && !va_ && lds_ok;
    if (d.a_type == LIBXSMM_DATATYPE_I16 || d.b_type == LIBXSMM_DATATYPE_I16)        // [ref: gemm ref :1427-1450]
      return d.a_type == d.b_type && d.c_type == LIBXSMM_DATATYPE_I32 && d.comp_type == LIBXSMM_DATATYPE_I32 && plain && !ta_ && !tb_ && !(va_ && (d.k & 1)) && lds_ok;
    if (d.a_type == LIBXSMM_DATATYPE_I8 && d.b_type == LIBXSMM_DATATYPE_BF16)       // row-scaled i8 weights x bf16 [ref: gemm ref :1684-1730]
      return (d.c_type == LIBXSMM_DATATYPE_BF16 || d.c_type == LIBXSMM_DATATYPE_F32) && d.comp_type == LIBXSMM_DATATYPE_F32 && plain && !ta_ && !tb_ && !va_ && lds_ok;
    if ((d.a_type == LIBXSMM_DATATYPE_BF8 || d.a_type == LIBXSMM_DATATYPE_HF8) && d.b_type == LIBXSMM_DATATYPE_BF16)      // 8-bit float weights x bf16 [ref: gemm ref :2171-2366]
      return (d.c_type == LIBXSMM_DATATYPE_BF16 || d.c_type == LIBXSMM_DATATYPE_F32) && d.comp_type == LIBXSMM_DATATYPE_F32 && plain && !ta_ && !tb_ && !(va_ && (d.k & 1)) && lds_ok;
  }
  if (i8) {   // [ref: gemm ref :1452-1683]: i32 accumulation; no transposes, no fused ops, f32 output needs VNNI-4 A
    const unsigned int fl8 = d.flags;
    if (d.comp_type != LIBXSMM_DATATYPE_I32) return false;
    if (fl8 & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C)) return false;
    const bool va8 = (fl8 & LIBXSMM_GEMM_FLAG_VNNI_A) != 0 || d.c_type == LIBXSMM_DATATYPE_F32;      // f32 result: A is read as VNNI-4 with or without the flag [ref: gemm ref :1556-1683] (effective_gemm_flags)
    if (va8 && (d.k & 3)) return false;
    if (d.bin_type != 0 || d.cp_type != 0 || d.ap_type != 0 || d.bp_type != 0) return false;
    return d.lda >= d.m && d.ldb >= d.k && d.ldc >= d.m;
  }
  if (d.a_type == LIBXSMM_DATATYPE_F16 && d.b_type == LIBXSMM_DATATYPE_F16) {   // [ref: gemm ref :2025-2124]: f32 accumulation, F16 or F32 out, VNNI-2 A optional, B may be transposed; fused operators [:294-372] (round 6)
    const unsigned int fl = d.flags;
    if (d.c_type != LIBXSMM_DATATYPE_F16 && d.c_type != LIBXSMM_DATATYPE_F32) return false;
    // comp F32, F16 (the running sum rounded to f16 after every product: generic kernel) or IMPLICIT (= F32 here; the reference means F16 by it on
    // AVX512-FP16 hosts only)
    if (d.comp_type != LIBXSMM_DATATYPE_F32 && d.comp_type != LIBXSMM_DATATYPE_F16 && d.comp_type != LIBXSMM_DATATYPE_IMPLICIT) return false;
    if (fl & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_VNNI_B)) return false;
    // VNNI_C: the finished F16 result re-laid as VNNI-2 [ref: gemm ref :2802-2815]; beta = 0, no fused operator (the reference's driver and test generator: gemm_kernel.c:3842, tpl :82)
    if ((fl & LIBXSMM_GEMM_FLAG_VNNI_C) && (d.c_type != LIBXSMM_DATATYPE_F16 || !(fl & LIBXSMM_GEMM_FLAG_BETA_0) || d.bin_type != 0 || d.cp_type != 0 || d.ap_type != 0 || d.bp_type != 0)) return false;
    if ((fl & LIBXSMM_GEMM_FLAG_VNNI_A) && (d.k & 1)) return false;
    if (!fused_ops_ok(d)) return false;
    if ((fl & LIBXSMM_GEMM_FLAG_TRANS_B) ? (d.ldb < d.n) : (d.ldb < d.k)) return false;
    return d.lda >= d.m && d.ldc >= d.m;
  }
  const bool fp8 = (d.a_type == LIBXSMM_DATATYPE_BF8 || d.a_type == LIBXSMM_DATATYPE_HF8) && d.b_type == d.a_type && (d.c_type == LIBXSMM_DATATYPE_F32 || d.c_type == d.a_type);      // C f32 or the operands' type [ref: :2511-2619]
  if (fp8) {   // [ref: gemm ref :2420-2510]: f32 accumulate and output, VNNI-4 A optional; fused operators [:294-372] (round 6)
    const unsigned int fl8 = d.flags;
    if (d.comp_type != LIBXSMM_DATATYPE_F32) return false;
    const bool ta8 = fl8 & LIBXSMM_GEMM_FLAG_TRANS_A, tb8 = fl8 & LIBXSMM_GEMM_FLAG_TRANS_B, va8 = fl8 & LIBXSMM_GEMM_FLAG_VNNI_A, vb8 = fl8 & LIBXSMM_GEMM_FLAG_VNNI_B;
    // VNNI_C: a result of the operands' 8-bit type re-laid as VNNI-4 [ref: gemm ref :2802-2815]; beta = 0, no fused operator
    if ((fl8 & LIBXSMM_GEMM_FLAG_VNNI_C) && (d.c_type != d.a_type || !(fl8 & LIBXSMM_GEMM_FLAG_BETA_0) || d.bin_type != 0 || d.cp_type != 0 || d.ap_type != 0 || d.bp_type != 0)) return false;
    if ((va8 && ta8) || (vb8 && !tb8) || ((va8 || vb8) && (d.k & 3))) return false;
    if (!fused_ops_ok(d)) return false;
    if (ta8 ? (d.lda < d.k) : (d.lda < d.m)) return false;
    if (tb8 ? (d.ldb < d.n) : (d.ldb < d.k)) return false;
    return d.ldc >= d.m;
  }
  if ((d.a_type == LIBXSMM_DATATYPE_MXFP4X2 || d.a_type == LIBXSMM_DATATYPE_MXBF8 || d.a_type == LIBXSMM_DATATYPE_MXHF8 || d.a_type == LIBXSMM_DATATYPE_MXBF6 || d.a_type == LIBXSMM_DATATYPE_MXHF6) && d.b_type == d.a_type) {
    // MX x MX -> f32 [ref: gemm ref :2620-2790]: A in VNNI, B in VNNI and transposed, no address/offset batch-reduce, no fused ops [:836-855]
    const unsigned int flx = d.flags;
    // C: f32, or the operands' own MX type for E2M1 / E5M2 (quantised in 32-row blocks, scales to c.tertiary; beta = 0 only: the reference
    // accumulates into an uninitialised buffer otherwise) [ref: gemm ref :2624-2678, :2731-2798]
    const bool mx_out = d.c_type == d.a_type && (d.a_type == LIBXSMM_DATATYPE_MXFP4X2 || d.a_type == LIBXSMM_DATATYPE_MXBF8);
    if ((d.c_type != LIBXSMM_DATATYPE_F32 && !mx_out) || d.comp_type != LIBXSMM_DATATYPE_F32) return false;
    if (mx_out && (!(flx & LIBXSMM_GEMM_FLAG_BETA_0) || (d.m % 32) != 0 || (d.ldc % 32) != 0 || (flx & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI))) return false;
    const unsigned int need = LIBXSMM_GEMM_FLAG_VNNI_A | LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_TRANS_B;
    if ((flx & need) != need || (flx & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_VNNI_C | LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT |
        LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET))) return false;
    if ((d.k % 32) != 0 || d.bin_type != 0 || d.cp_type != 0 || d.ap_type != 0 || d.bp_type != 0) return false;
    if ((d.a_type == LIBXSMM_DATATYPE_MXBF6 || d.a_type == LIBXSMM_DATATYPE_MXHF6) && (((d.lda * 6) % 8) != 0 || ((d.ldb * 6) % 8) != 0)) return false;   // a k-group row of the image is ld * 6 / 8 bytes
    return d.lda >= d.m && d.ldb >= d.n && d.ldc >= d.m;
  }
  if ((d.flags & LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT) && (d.a_type == LIBXSMM_DATATYPE_I4X2 || d.a_type == LIBXSMM_DATATYPE_U4X2 || d.a_type == LIBXSMM_DATATYPE_MXFP4X2) &&
      (d.b_type == LIBXSMM_DATATYPE_I8 || d.b_type == LIBXSMM_DATATYPE_U8)) {
    // interleaved 4-bit weights x 8-bit activations [ref: gemm ref :467-477, :1009-1088, :1272-1330]: I4X2 (minus a zero point per row,
    // a.quaternary; B read as unsigned bytes) -> i32; MXFP4 (integer table, E8M0 scales in a.tertiary, f32 scales of B in b.tertiary) -> f32 / bf16
    const unsigned int fl = d.flags;
    const bool mx = d.a_type == LIBXSMM_DATATYPE_MXFP4X2;
    if (mx ? !((d.c_type == LIBXSMM_DATATYPE_F32 || d.c_type == LIBXSMM_DATATYPE_BF16) && (d.comp_type == LIBXSMM_DATATYPE_F32 || d.comp_type == LIBXSMM_DATATYPE_I32) && d.b_type == LIBXSMM_DATATYPE_I8)
           : !(d.c_type == LIBXSMM_DATATYPE_I32 && d.comp_type == LIBXSMM_DATATYPE_I32)) return false;
    if (!(fl & LIBXSMM_GEMM_FLAG_VNNI_A) || (fl & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C |
        LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET))) return false;
    if ((d.k % (mx ? 32 : 8)) != 0 || (d.ldb & 3) || (mx && (d.ldb % 32) != 0)) return false;
    if (d.bin_type != 0 || d.cp_type != 0 || d.ap_type != 0 || d.bp_type != 0) return false;
    return d.lda >= d.m && d.ldb >= d.k && d.ldc >= d.m;
  }
  if (d.a_type == LIBXSMM_DATATYPE_MXFP4X2) {   // MXFP4 weights x bf16/f32 activations [ref: gemm ref :457-465, :949-1008; names libxsmm_main.c:1829-1848]
    const unsigned int flx = d.flags;
    const bool okt = (d.b_type == LIBXSMM_DATATYPE_BF16 && (d.c_type == LIBXSMM_DATATYPE_BF16 || d.c_type == LIBXSMM_DATATYPE_F32)) ||
                     (d.b_type == LIBXSMM_DATATYPE_F32 && d.c_type == LIBXSMM_DATATYPE_F32);
    if (!okt || d.comp_type != LIBXSMM_DATATYPE_F32) return false;
    if (!(flx & LIBXSMM_GEMM_FLAG_VNNI_A) || (flx & (LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT | LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B |
        LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C))) return false;
    if ((d.k % 32) != 0 || (d.lda & 1)) return false;                       // one scale per 32-deep k-block; k*lda/2 bytes per block
    if (d.bin_type != 0 || d.cp_type != 0 || d.ap_type != 0 || d.bp_type != 0) return false;
    return d.lda >= d.m && d.ldb >= d.k && d.ldc >= d.m;
  }
  if (!(f32 || f64 || bf16)) return false;
  if (f32 && d.comp_type != LIBXSMM_DATATYPE_F32) return false;
  if (f64 && d.comp_type != LIBXSMM_DATATYPE_F64) return false;
  if (bf16 && d.comp_type != LIBXSMM_DATATYPE_F32) return false;
  const unsigned int fl = d.flags;
  const bool ta = fl & LIBXSMM_GEMM_FLAG_TRANS_A, tb = fl & LIBXSMM_GEMM_FLAG_TRANS_B;
  const bool va = fl & LIBXSMM_GEMM_FLAG_VNNI_A, vb = fl & LIBXSMM_GEMM_FLAG_VNNI_B, vc = fl & LIBXSMM_GEMM_FLAG_VNNI_C;
  if (!bf16 && vc) return false;                      // (VNNI_A / VNNI_B on f32 / f64 operands: accepted and ignored, as the reference does -- run_gemm clears them)
  if (bf16 && va && ta) return false;                 // [ref: src/generator_gemm.c:1010-1013]
  if (bf16 && va && (d.k & 1)) return false;
  if (bf16 && vb && !tb) return false;                // VNNI_B only defined together with TRANS_B [ref: gemm ref :2156-2163]
  if (bf16 && vb && (d.k & 1)) return false;
  if (vc && d.c_type != LIBXSMM_DATATYPE_BF16) return false;
  if (fl & (LIBXSMM_GEMM_FLAG_USE_COL_VEC_SCF | LIBXSMM_GEMM_FLAG_USE_COL_VEC_ZPT | LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT |
            LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK | LIBXSMM_GEMM_FLAG_USE_MxK_ZPT | LIBXSMM_GEMM_FLAG_USE_MxK_SCF)) return false;
  // leading dimensions [ref: src/generator_gemm.c:990-1040]; VNNI-2 A keeps lda >= m (in k-pairs)
  if (ta ? (d.lda < d.k) : (d.lda < d.m)) return false;
  if (tb ? (d.ldb < d.n) : (d.ldb < d.k)) return false;
  if (d.ldc < d.m) return false;
  if (f64 && (d.bin_type != 0 || d.cp_type != 0)) return false;
  return true;
}

enum GemmPath { P_GENERIC, P_F32_T16, P_F32_1x1, P_F32_2x2, P_BF16_1x1, P_BF16_2x2, P_I8_1x1, P_I8_2x2, P_FP8_1x1, P_FP8_2x2, P_MX4_1x1, P_MX4_2x2, P_MXMX_1x1, P_MXMX_2x2, P_MX4I8_1x1, P_MX4I8_2x2, P_M8_1x1, P_M8_2x2, P_W8_1x1, P_W8_2x2 };
struct GemmPlan { GemmPath path; bool exact; };

static GemmPlan plan_gemm(int m, int n, int k, unsigned int flags, int a_type, int b_type, int c_type, int vnni_c) {
  GemmPlan pl{P_GENERIC, false};
  const bool ta = flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = flags & LIBXSMM_GEMM_FLAG_TRANS_B;
  const bool va = flags & LIBXSMM_GEMM_FLAG_VNNI_A, vb = flags & LIBXSMM_GEMM_FLAG_VNNI_B;
  (void)c_type;
  if (vnni_c || k <= 0) return pl;
  // MXBF6 (E3M2) stays on the exact generic kernel: the matrix core aligns the 32 products of a block before adding them and E3M2 spans eight binades --
  // measured 1.7e-5 .. 2.2e-5 (normf) against the reference, above the 1.2e-5 its own gemm_kernel driver accepts; E2M3 (MXHF6) measures 0 .. 3e-8
  if ((a_type == LIBXSMM_DATATYPE_MXFP4X2 || a_type == LIBXSMM_DATATYPE_MXBF8 || a_type == LIBXSMM_DATATYPE_MXHF8 || a_type == LIBXSMM_DATATYPE_MXHF6) && b_type == a_type) {
    if ((m % 32) || (n % 32) || (k % 64)) return pl;
    if (a_type == LIBXSMM_DATATYPE_MXHF6 && c_type != LIBXSMM_DATATYPE_F32) return pl;
    pl.exact = true;
    pl.path = ((m % 64) == 0 && (n % 64) == 0) ? P_MXMX_2x2 : P_MXMX_1x1;
    return pl;
  }
  if (a_type == LIBXSMM_DATATYPE_MXFP4X2 && (flags & LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT) && b_type == LIBXSMM_DATATYPE_I8 && va && !ta && !tb && !vb) {
    // interleaved MXFP4 x i8: one int8 MFMA per 32-deep block, scaled and added block by block (gemm_mx4i8_stream_kernel)
    if ((m % 32) || (n % 32) || (k % 64)) return pl;
    pl.exact = true;
    // 2 x 2 tiles per wave need 320 registers (64 f32 sums + the MFMA results in flight): one wave per SIMD, 0.17 of the HBM roofline on 64^3 problems;
    // one tile per wave (120 registers, four waves per SIMD) re-reads operands from L2 and runs several times faster
    pl.path = P_MX4I8_1x1;
    return pl;
  }
  if (a_type == LIBXSMM_DATATYPE_MXFP4X2) {
    if (b_type != LIBXSMM_DATATYPE_BF16 || (m % 32) || (n % 32) || (k % 32)) return pl;
    pl.exact = true;
    pl.path = ((m % 64) == 0 && (n % 64) == 0) ? P_MX4_2x2 : P_MX4_1x1;
    return pl;
  }
  if (a_type == LIBXSMM_DATATYPE_F32) {
    const bool ex32 = (m % 32 == 0) && (n % 32 == 0) && (k % 32 == 0);
    if (!ex32 && !ta && !tb && (m % 16 == 0) && (n % 16 == 0) && (k % 16 == 0) && m <= 48 && n <= 48) { pl.path = P_F32_T16; pl.exact = true; return pl; }
    pl.exact = ex32;
    pl.path = (m > 32 && n > 32) ? P_F32_2x2 : P_F32_1x1;
    if (pl.path == P_F32_2x2) {
      pl.exact = (m % 64 == 0) && (n % 64 == 0) && (k % 32 == 0);
      if (!pl.exact && ex32) { pl.path = P_F32_1x1; pl.exact = true; }     // e.g. 96^3: exact 32x32 tiles beat masked 64x64 tiles
    }
    return pl;
  }
  if ((a_type == LIBXSMM_DATATYPE_BF8 || a_type == LIBXSMM_DATATYPE_HF8 || a_type == LIBXSMM_DATATYPE_I8) && b_type == LIBXSMM_DATATYPE_BF16 && !ta && !tb && !vb &&
      (c_type == LIBXSMM_DATATYPE_F32 || c_type == LIBXSMM_DATATYPE_BF16) && !(a_type == LIBXSMM_DATATYPE_I8 && va)) {
    pl.path = (m > 32 && n > 32) ? P_W8_2x2 : P_W8_1x1;         // 8-bit weights x bf16 activations: converted in registers, bf16 matrix cores (round 4)
    return pl;
  }
  if ((a_type == LIBXSMM_DATATYPE_BF8 || a_type == LIBXSMM_DATATYPE_HF8) && (b_type != a_type || (c_type != LIBXSMM_DATATYPE_F32 && c_type != a_type))) return pl;      // mixed operands: generic kernel (C of the operands' type: round 4)
  if ((a_type == LIBXSMM_DATATYPE_BF8 || a_type == LIBXSMM_DATATYPE_HF8) && va && !ta && !tb && !vb) {
    pl.path = (m > 32 && n > 32) ? P_FP8_2x2 : P_FP8_1x1;
    const int t = (pl.path == P_FP8_2x2) ? 64 : 32;
    pl.exact = (m % t == 0) && (n % t == 0) && (k % 32 == 0);
    if (!pl.exact && pl.path == P_FP8_2x2 && (m % 32 == 0) && (n % 32 == 0) && (k % 32 == 0)) { pl.path = P_FP8_1x1; pl.exact = true; }      // e.g. 96 x 64: whole 32 x 32 tiles
    if (!pl.exact) pl.path = (k % 4 == 0) ? (pl.path == P_FP8_2x2 ? P_M8_2x2 : P_M8_1x1) : P_GENERIC;        // any shape with whole k-quads: the masked matrix-core kernel (round 4)
    return pl;
  }
  if ((a_type == LIBXSMM_DATATYPE_I4X2 || a_type == LIBXSMM_DATATYPE_U4X2) && (flags & LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT) && va && !ta && !tb && !vb &&
      (b_type == LIBXSMM_DATATYPE_I8 || b_type == LIBXSMM_DATATYPE_U8) && c_type == LIBXSMM_DATATYPE_I32) {
    // interleaved 4-bit weights: the int8 streaming kernel with the nibbles expanded (and the row's zero point subtracted) in registers
    pl.path = (m > 32 && n > 32) ? P_I8_2x2 : P_I8_1x1;
    const int t = (pl.path == P_I8_2x2) ? 64 : 32;
    pl.exact = (m % t == 0) && (n % t == 0) && (k % 32 == 0);
    if (!pl.exact) pl.path = P_GENERIC;
    return pl;
  }
  if ((a_type == LIBXSMM_DATATYPE_I1X8 || a_type == LIBXSMM_DATATYPE_I2X4) && va && !ta && !tb && !vb && (b_type == LIBXSMM_DATATYPE_I8 || b_type == LIBXSMM_DATATYPE_U8) && c_type == LIBXSMM_DATATYPE_I32) {
    // 1- / 2-bit weights: the int8 streaming kernel with the bits expanded to +-1 / 0 bytes in registers (round 3)
    pl.path = (m > 32 && n > 32) ? P_I8_2x2 : P_I8_1x1;
    const int t = (pl.path == P_I8_2x2) ? 64 : 32;
    pl.exact = (m % t == 0) && (n % t == 0) && (k % 32 == 0);
    if (!pl.exact && pl.path == P_I8_2x2 && (m % 32 == 0) && (n % 32 == 0) && (k % 32 == 0)) { pl.path = P_I8_1x1; pl.exact = true; }
    if (!pl.exact) pl.path = P_GENERIC;
    return pl;
  }
  if ((a_type == LIBXSMM_DATATYPE_I8 || a_type == LIBXSMM_DATATYPE_U8) && va && !ta && !tb && !vb) {
    if (b_type != LIBXSMM_DATATYPE_I8 && b_type != LIBXSMM_DATATYPE_U8) return pl;
    pl.path = (m > 32 && n > 32) ? P_I8_2x2 : P_I8_1x1;
    const int t = (pl.path == P_I8_2x2) ? 64 : 32;
    pl.exact = (m % t == 0) && (n % t == 0) && (k % 32 == 0);
    if (!pl.exact && pl.path == P_I8_2x2 && (m % 32 == 0) && (n % 32 == 0) && (k % 32 == 0)) { pl.path = P_I8_1x1; pl.exact = true; }
    if (!pl.exact) pl.path = (k % 4 == 0) ? (pl.path == P_I8_2x2 ? P_M8_2x2 : P_M8_1x1) : P_GENERIC;         // any shape with whole k-quads: the masked matrix-core kernel (round 4)
    return pl;
  }
  if ((a_type == LIBXSMM_DATATYPE_BF16 || (a_type == LIBXSMM_DATATYPE_F16 && b_type == LIBXSMM_DATATYPE_F16)) && va && !ta && !tb && !vb) {
    pl.path = (m > 32 && n > 32) ? P_BF16_2x2 : P_BF16_1x1;
    const int t = (pl.path == P_BF16_2x2) ? 64 : 32;
    pl.exact = (m % t == 0) && (n % t == 0) && (k % 32 == 0);
    if (!pl.exact && pl.path == P_BF16_2x2 && (m % 32 == 0) && (n % 32 == 0) && (k % 32 == 0)) { pl.path = P_BF16_1x1; pl.exact = true; }
    // (round 5, measured and dropped: 32 x 32 tiles for ragged shapes above 32 -- nine light waves cover 96 x 96 of a 72^3 problem instead of four waves 128 x 128 --
    //  72^3 0.38 against 0.43, 40^3 0.44 against 0.57, both in the bounded form: profiles/r05_ragged16_switches.jsonl)
    return pl;
  }
  return pl;
}

static const char* path_name(GemmPath p) {
  switch (p) {
    case P_F32_T16: return "gemm_mfma_f32_t16_kernel";
    case P_F32_1x1: return "gemm_mfma_f32_kernel<1,1>";
    case P_F32_2x2: return "gemm_mfma_f32_kernel<2,2>";
    case P_BF16_1x1: return "gemm_mfma_bf16_kernel<1,1>";
    case P_BF16_2x2: return "gemm_mfma_bf16_kernel<2,2>";
    case P_FP8_1x1: return "gemm_fp8_stream_kernel<1,1>";
    case P_FP8_2x2: return "gemm_fp8_stream_kernel<2,2>";
    case P_MX4I8_1x1: return "gemm_mx4i8_stream_kernel<1,1>";
    case P_MX4I8_2x2: return "gemm_mx4i8_stream_kernel<2,2>";
    case P_I8_1x1: return "gemm_i8_stream_kernel<1,1>";
    case P_I8_2x2: return "gemm_i8_stream_kernel<2,2>";
    case P_MX4_1x1: return "gemm_mxfp4_stream_kernel<1,1>";
    case P_MX4_2x2: return "gemm_mxfp4_stream_kernel<2,2>";
    case P_MXMX_1x1: return "gemm_mx_stream_kernel<1,1>";
    case P_MXMX_2x2: return "gemm_mx_stream_kernel<2,2>";
    case P_W8_1x1: return "gemm_w8_bf16_kernel<1,1>";
    case P_W8_2x2: return "gemm_w8_bf16_kernel<2,2>";
    case P_M8_1x1: return "gemm_mfma_8bit_kernel<1,1>";
    case P_M8_2x2: return "gemm_mfma_8bit_kernel<2,2>";
    default: return "gemm_generic_kernel";
  }
}

const char* gemm_kernel_name(const libxsmm_gemm_descriptor& d, bool) {
  if (d.a_type == LIBXSMM_DATATYPE_F64) return gemm_f64_kernel_name(d);
  return path_name(plan_gemm((int)d.m, (int)d.n, (int)d.k, d.flags, d.a_type, d.b_type, d.c_type, (d.flags & LIBXSMM_GEMM_FLAG_VNNI_C) != 0).path);
}

template <int MT, int NT, int MODE>
static void launch_f32(const GemmArgs& a, dim3 grid, hipStream_t st) {
  const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B;
  if (!ta && !tb) hipLaunchKernelGGL((gemm_mfma_f32_kernel<MT, NT, false, false, MODE>), grid, dim3(256), 0, st, a);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_mfma_f32_kernel<MT, NT, true, false, MODE>), grid, dim3(256), 0, st, a);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_mfma_f32_kernel<MT, NT, false, true, MODE>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_mfma_f32_kernel<MT, NT, true, true, MODE>), grid, dim3(256), 0, st, a);
}
// The staged path needs every operand tile 16-byte aligned.  That is decidable on the host for the
// strided forms; pointer lists / offset arrays live on the device and take the direct-load kernel.
static bool operands_aligned16(const GemmArgs& a, int elem_size) {
  if (a.br_mode == 1 || a.br_mode == 2) return false;
  if (a.list_a) {                                   // pointer lists live on the device; lists the library built itself come with their alignment
    if (!a.lists_aligned16) return false;
    const unsigned long long lbits = (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0) | (unsigned long long)((long long)a.lda * elem_size) | (unsigned long long)((long long)a.ldb * elem_size);
    return (lbits & 15ull) == 0ull;
  }
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a |
    (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0) |
    (unsigned long long)((long long)a.lda * elem_size) | (unsigned long long)((long long)a.ldb * elem_size);
  return (bits & 15ull) == 0ull;
}

// ragged bf16 / f16 shapes: every B block and column starts on a dword (strided forms only: the alignment of listed blocks is not known on the host), so that the
// wave's B panel can be fetched a dword per lane by LDS-DMA (gemm_mfma_bf16_kernel<.., BL = true>); LIBXSMM_HIP_RAGGED16_LDS=0 keeps B in registers (measurement switch)
static bool ragged16_b_dwords(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || a.list_a || a.br_mode == 1 || a.br_mode == 2 || (a.ldb & 1)) return false;
  const unsigned long long bits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? a.br_stride_b : 0);
  return (bits & 3ull) == 0ull && (unsigned long long)a.n * (unsigned long long)a.ldb < (1ull << 30) && (unsigned long long)a.k * (unsigned long long)a.lda < (1ull << 30);
}
// BND form (round 5: adopted -- bf16 40^3 0.49 -> 0.57, 24^3 0.60 -> 0.70, 72^3 0.35 -> 0.43 of the HBM roofline, profiles/r05_ragged16_switches.jsonl): plain results
// (beta = 0, no bias, no bitmask, no VNNI C), A blocks on dwords like B's (ragged16_b_dwords), extents that fit a 32-bit buffer resource
static bool ragged16_bounded(const GemmArgs& a) {
  const unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | (unsigned long long)(a.br_mode == 3 ? a.br_stride_a : 0);
  return (abits & 3ull) == 0ull && (a.flags & LIBXSMM_GEMM_FLAG_BETA_0) && !a.colbias && a.act != 2 && !a.relu_mask && !a.vnni_c && (unsigned long long)a.n * (unsigned long long)a.ldc < (1ull << 29);
}
// one masked tile, blobs of at most 1024 dwords, dword-aligned operands, no transposes
static bool f32_blob_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B))) return false;
  return a.m <= 32 && a.n <= 32 && a.k <= 32 && a.lda <= 32 && a.ldb <= 32 && a.k * a.lda <= 1024 && a.n * a.ldb <= 1024;
}
// the lean streaming kernel: one 32x32 tile per problem, 1-D strided batch, plain or STRIDE batch-reduce with at least one block,
// beta = 0, no fused epilogue, every 32-bit offset inside a tile representable
static bool f32_lean_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || a.m != 32 || a.n != 32 || a.batch_inner || a.list_a || (a.br_mode != 0 && a.br_mode != 3) || a.br_count < 1) return false;
  if (!(a.flags & LIBXSMM_GEMM_FLAG_BETA_0) || a.colbias || a.act || a.vnni_c) return false;
  if ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c) & 3ull) != 0) return false;
  return a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22) && a.br_count * (unsigned long long)(a.k >> 5) < (1ull << 31);
}
// the workgroup-cooperative blocked kernel: 2-D batch whose grid divides into 128 x 128 macro tiles, square 32^3 / 64^3 problems,
// no transposes, plain or STRIDE batch-reduce, 16-byte aligned operands, 32-bit offsets inside a block
// the blocked kernel on 16 x 16 x K tiles: grid divisible into 8 x 8 problems, an even number of 16-deep sub-steps, plain epilogue, f32, NN, strided
// macro-tile kernel (gemm_bf16_macro_kernel): 16 / 32 / 64 tiles; every request offset (lane part + stage part) must fit 32 bits
static bool bf16_macro_ok(const GemmArgs& a, bool& k16) {
  constexpr bool off = false;
  if (off || !a.batch_inner || a.list_a || (a.br_mode != 0 && a.br_mode != 3) || a.br_count == 0) return false;
  const bool f16 = a.a_type == LIBXSMM_DATATYPE_F16;            // IEEE halves: same layouts, v_mfma_f32_32x32x16_f16 (64 / 32 tiles, f32 accumulation only)
  if ((a.a_type != LIBXSMM_DATATYPE_BF16 && !f16) || a.b_type != a.a_type || !(a.flags & LIBXSMM_GEMM_FLAG_VNNI_A)) return false;
  if ((a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B)) || a.vnni_c || a.colbias || a.act || !(a.flags & LIBXSMM_GEMM_FLAG_BETA_0)) return false;
  if (a.c_type != a.a_type && a.c_type != LIBXSMM_DATATYPE_F32) return false;
  if (f16 && (a.comp_f16 || a.m == 16)) return false;
  if (a.m != a.n || (a.m != 64 && a.m != 32 && a.m != 16) || a.k <= 0) return false;
  k16 = (a.k % 32) != 0;
  if (k16 && (a.k != 16 || (a.br_count & 1ull) || a.br_mode != 3)) return false;       // a stage of 32 k = two consecutive blocks of 16
  const unsigned int ni = a.batch_inner, nj = a.nbatch / a.batch_inner, ppm = 256u / (unsigned int)a.m;
  const unsigned long long stages = k16 ? a.br_count / 2ull : a.br_count * (unsigned long long)(a.k >> 5);
  if (ni % ppm || nj % ppm || stages >= (1ull << 31)) return false;
  const unsigned long long bra = a.br_mode == 3 ? (unsigned long long)a.br_stride_a : 0ull, brb = a.br_mode == 3 ? (unsigned long long)a.br_stride_b : 0ull;
  if (a.bs_a < 0 || a.bs_b < 0 || (a.br_mode == 3 && (a.br_stride_a < 0 || a.br_stride_b < 0))) return false;
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b | bra | brb |
    (unsigned long long)((long long)a.lda * 4) | (unsigned long long)((long long)a.ldb * 2);
  if ((bits & 15ull) != 0ull || a.lda >= (1 << 20) || a.ldb >= (1 << 20)) return false;
  // largest byte offset a request can form inside the macro tile's operand panel: (ppm - 1) problems + the whole chain + one stage
  const unsigned long long spanA = (unsigned long long)(ppm - 1u) * (unsigned long long)a.bs_a + a.br_count * bra + (unsigned long long)a.k * (unsigned long long)a.lda * 2ull + 4096ull;
  const unsigned long long spanB = (unsigned long long)(ppm - 1u) * (unsigned long long)a.bs_b + a.br_count * brb + (unsigned long long)a.m * (unsigned long long)a.ldb * 2ull + 4096ull;
  return spanA < (1ull << 32) && spanB < (1ull << 32);
}
static bool f32_blocked16_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || !a.batch_inner || a.list_a || (a.br_mode != 0 && a.br_mode != 3) || a.a_type != LIBXSMM_DATATYPE_F32 || a.b_type != LIBXSMM_DATATYPE_F32 || a.c_type != LIBXSMM_DATATYPE_F32) return false;
  if ((a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B)) || a.vnni_c || !(a.flags & LIBXSMM_GEMM_FLAG_BETA_0) || a.colbias || a.act) return false;
  if (a.m != 16 || a.n != 16 || a.k <= 0 || (a.k % 16) != 0) return false;
  const unsigned long long subs = a.br_count * (unsigned long long)(a.k / 16);
  if (subs == 0 || (subs & 1ull) || subs >= (1ull << 31)) return false;
  const unsigned int ni = a.batch_inner, nj = a.nbatch / a.batch_inner;
  if (ni % 8u || nj % 8u) return false;
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0) | (unsigned long long)((long long)a.lda * 4) | (unsigned long long)((long long)a.ldb * 4);
  return (bits & 15ull) == 0ull && a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22);
}
static bool f32_blocked_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || !a.batch_inner || a.list_a || (a.br_mode != 0 && a.br_mode != 3)) return false;
  if ((a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B)) || a.vnni_c) return false;
  if (!((a.m == 32 && a.n == 32) || (a.m == 64 && a.n == 64)) || (a.k % 32) != 0 || a.k <= 0) return false;
  const unsigned int ppw = 4 / (a.m / 32), ni = a.batch_inner, nj = a.nbatch / a.batch_inner;
  if (ni % ppw || nj % ppw) return false;
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
    (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0) | (unsigned long long)((long long)a.lda * 4) | (unsigned long long)((long long)a.ldb * 4);
  return (bits & 15ull) == 0ull && a.lda < (1 << 22) && a.ldb < (1 << 22);
}
// Non-temporal operand loads for the streaming kernels?  The calling thread's hint decides (libxsmm_hip_set_streaming_hint), by default the
// size of the launch: operands of a launch that moves more than the 256 MiB Infinity Cache holds cannot be resident in it, so nothing is lost
// by not keeping them (measured: +7..10 % on such launches), while smaller launches keep cacheable operands (nt reads of RESIDENT operands
// measured 50 % slower).  Operands shared by the batch (stride 0) or re-used by a 2-D batch are always cacheable.
static int typesize_c(const GemmArgs& a) { return a.c_type == LIBXSMM_DATATYPE_F32 ? 4 : 2; }
static bool stream_nt(const GemmArgs& a, int elem_bytes_ab, int elem_bytes_c) {
  static const int force = []() { const char* e = getenv("LIBXSMM_HIP_NT"); return e ? atoi(e) : -1; }();      // experiments: 0 never, 1 always
  if (force == 0) return false;
  if (force == 1) return true;
  if (a.batch_inner || a.list_a || a.bs_a == 0 || a.bs_b == 0 || a.stream_hint == 1) return false;
  if (a.stream_hint == 2) return true;
  const unsigned long long per = a.br_count * (unsigned long long)a.k * (unsigned long long)(a.m + a.n) * elem_bytes_ab + (unsigned long long)a.m * a.n * elem_bytes_c;
  return per * a.nbatch > (256ull << 20) || rt_recent_operands_exceed_cache(a.a, per * a.nbatch, a.c);      // (round 6: ... or this thread's recent launches together)
}
// the four-problems-per-wave kernel for 16 x 16 problems: m = n = 16, k % 16 == 0, no transposes, beta = 0, no fused epilogue, operands whose
// vector loads are aligned (B columns and C columns 16 bytes for f32, 8 bytes for bf16; bf16 A in VNNI-2 with dword-aligned rows)
static bool p16_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || a.m != 16 || a.n != 16 || a.k <= 0 || (a.k % 16) != 0 || a.vnni_c || a.colbias || a.act) return false;
  if (!(a.flags & LIBXSMM_GEMM_FLAG_BETA_0) || (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B))) return false;
  if (a.br_mode == 1 || a.br_mode == 2 || a.list_a) return false;            // strided forms only: alignment is decidable on the host
  const bool f32 = a.a_type == LIBXSMM_DATATYPE_F32 && a.b_type == LIBXSMM_DATATYPE_F32 && a.c_type == LIBXSMM_DATATYPE_F32 && !(a.flags & LIBXSMM_GEMM_FLAG_VNNI_A);
  const bool bf16 = a.a_type == LIBXSMM_DATATYPE_BF16 && a.b_type == LIBXSMM_DATATYPE_BF16 && (a.flags & LIBXSMM_GEMM_FLAG_VNNI_A) &&
    (a.c_type == LIBXSMM_DATATYPE_BF16 || a.c_type == LIBXSMM_DATATYPE_F32);
  if (!f32 && !bf16) return false;
  const unsigned long long brs = a.br_mode == 3 ? (unsigned long long)(a.br_stride_a | a.br_stride_b) : 0ull;
  const unsigned long long eb = f32 ? 4ull : 2ull, ec = a.c_type == LIBXSMM_DATATYPE_F32 ? 4ull : 2ull;
  const unsigned long long bbits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (a.br_mode == 3 ? (unsigned long long)a.br_stride_b : 0ull) | (unsigned long long)a.ldb * eb;
  const unsigned long long cbits = (unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)a.bs_c2 | (unsigned long long)a.ldc * ec;
  const unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | brs;
  if (f32) return (bbits & 15ull) == 0 && (cbits & 15ull) == 0 && (abits & 3ull) == 0;
  return (bbits & 7ull) == 0 && (cbits & (ec == 4 ? 15ull : 7ull)) == 0 && (abits & 3ull) == 0;
}
static bool f32_wg64_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || (a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B)) || a.list_a || (a.br_mode != 0 && a.br_mode != 3)) return false;
  return a.lda < (1 << 22) && a.ldb < (1 << 22) && a.k >= 32 && a.br_count * (unsigned long long)(a.k >> 5) < (1ull << 31);
}
// bf16 64 x 64 problems, one per workgroup: VNNI-2 A with 16-byte aligned rows (lda % 4 == 0), flat B with 16-byte aligned columns, strided forms
static bool bf16_wg64_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || a.list_a || (a.br_mode != 0 && a.br_mode != 3)) return false;
  const unsigned long long brs = a.br_mode == 3 ? (unsigned long long)(a.br_stride_a | a.br_stride_b) : 0ull;
  const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b | brs |
    (unsigned long long)((long long)a.lda * 4) | (unsigned long long)((long long)a.ldb * 2);
  return (bits & 15ull) == 0 && a.lda < (1 << 22) && a.ldb < (1 << 22) && a.k >= 32 && a.br_count * (unsigned long long)(a.k >> 5) < (1ull << 31);
}
// rounds per operand of the instantiation that serves (waves, tile pairs per wave, rounds needed) -- see the dispatch in launch_gemm
static unsigned int ragged_rounds_cap(int waves, int np, unsigned int rounds) {
  if (waves == 1) return (np == 1 && rounds <= 4u) ? 4u : (rounds <= 8u ? 8u : (rounds <= 12u ? 12u : 17u));
  if (np == 2 && rounds <= 7u) return 7u;
  if (np <= 4 && rounds <= 10u) return 10u;
  return rounds <= 14u ? 14u : 24u;
}
// the ragged kernel: f32 NN with a plain epilogue, 2 <= m <= 128, n <= 128, at most 24 tile pairs; fills in the chunk plan
static bool f32_ragged_plan(const GemmArgs& a, RaggedCfg& c, int& waves, int& np, int& rounds, unsigned int& lds_bytes, bool& single) {
  constexpr bool off = false;
  if (off || a.a_type != LIBXSMM_DATATYPE_F32 || a.b_type != LIBXSMM_DATATYPE_F32 || a.c_type != LIBXSMM_DATATYPE_F32) return false;
  if ((a.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_A | LIBXSMM_GEMM_FLAG_VNNI_B)) || a.vnni_c || a.colbias || a.act) return false;
  // leading dimensions below 2^20: every round offset (columns x leading dimension x 4 bytes) stays below 2^31
  if (a.m < 2 || a.m > 128 || a.n < 1 || a.n > 128 || a.k < 1 || a.lda >= (1 << 20) || a.ldb >= (1 << 20) || a.ldc >= (1 << 20)) return false;
  const unsigned int m = (unsigned int)a.m, n = (unsigned int)a.n, K = (unsigned int)a.k;
  const unsigned int ti = (m + 15u) / 16u, tj = (n + 15u) / 16u;
  c.tpi = (ti + 1u) / 2u; c.npairs = c.tpi * tj;
  if (c.npairs > 24u) return false;
  waves = (m <= 32u && n <= 32u) ? 1 : 4;
  const unsigned int per_wave = (c.npairs + (unsigned int)waves - 1u) / (unsigned int)waves;
  np = waves == 1 ? (per_wave <= 1u ? 1 : 2) : (per_wave <= 2u ? 2 : (per_wave <= 4u ? 4 : 6));
  const bool beta0 = (a.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const unsigned int T = 64u * (unsigned int)waves, rmax = waves == 1 ? 17u : 24u;
  c.cpr_a = T / m;                                                  // columns of A / C per round
  if (c.cpr_a == 0 || (n + c.cpr_a - 1u) / c.cpr_a > rmax) return false;      // C (beta = 1 in, always out) in at most rmax rounds
  // depth of a chunk: A in at most rmax rounds (kc <= rmax * cpr_a), B too (columns per round T / kc >= n / rmax)
  unsigned int kmax = std::min(rmax * c.cpr_a, T / ((n + rmax - 1u) / rmax));
  if (kmax >= K) { c.kc = K; c.kchunks = 1; }
  else {
    kmax &= ~3u;
    if (kmax == 0) return false;
    c.kchunks = (K + kmax - 1u) / kmax; c.kc = (((K + c.kchunks - 1u) / c.kchunks) + 3u) & ~3u; c.kchunks = (K + c.kc - 1u) / c.kc;
  }
  c.cpr_b = T / c.kc;
  const unsigned int ra = (c.kc + c.cpr_a - 1u) / c.cpr_a, rb = (n + c.cpr_b - 1u) / c.cpr_b, rcn = (n + c.cpr_a - 1u) / c.cpr_a;
  rounds = (int)std::max(ra, std::max(rb, beta0 ? 0u : rcn));
  if (rounds > (int)rmax) return false;
  const unsigned int k4 = (c.kc + 3u) & ~3u;
  c.pb = (k4 & 1u) ? k4 : k4 + 2u;                                  // B column pitch == 2 (mod 4): the 16 columns of a fragment read hit different banks
  c.pc = 16u * ti;
  // LDS images: every round of the chosen instantiation is written (rows / columns behind the block receive zeros)
  const unsigned int rcap = ragged_rounds_cap(waves, np, (unsigned int)rounds);
  const unsigned int rows_a = std::max(rcap * c.cpr_a, k4), cols_b = rcap * c.cpr_b, cols_c = std::max(rcap * c.cpr_a, (n + 3u) & ~3u);
  // every scalar round offset below 2^31 bytes
  if ((unsigned long long)rcap * c.cpr_a * a.lda * 4ull >= (1ull << 31) || (unsigned long long)rcap * c.cpr_b * a.ldb * 4ull >= (1ull << 31) || (unsigned long long)rcap * c.cpr_a * a.ldc * 4ull >= (1ull << 31)) return false;
  single = c.kchunks == 1 && a.br_count == 1;                      // one chunk per problem: one workgroup per problem, the C image in place of the operands
  c.off_b = rows_a * m;
  c.off_ci = c.off_b + cols_b * c.pb;
  const unsigned int operands = c.off_ci + (beta0 ? 0u : cols_c * c.pc);
  c.off_co = single ? 0u : operands;
  c.spare = single ? std::max(operands, cols_c * c.pc) : operands + cols_c * c.pc;
  rounds = (int)rcap;
  lds_bytes = 4u * ((c.spare + T + 3u) & ~3u);
  return lds_bytes <= 65536u;
}
template <int W, int NP, int R2>
static void launch_ragged(const GemmArgs& a, const RaggedCfg& c, unsigned int lds_bytes, bool single, hipStream_t st) {
  const bool beta0 = (a.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  if (single) {
    if (beta0) hipLaunchKernelGGL((gemm_f32_ragged_kernel<W, NP, R2, false, true>), dim3(a.nbatch), dim3(64 * W), lds_bytes, st, a, c);
    else hipLaunchKernelGGL((gemm_f32_ragged_kernel<W, NP, R2, true, true>), dim3(a.nbatch), dim3(64 * W), lds_bytes, st, a, c);
  }
  // several chunks per problem: the chunk loop with its register prefetch, still one workgroup per problem (a grid of resident
  // workgroups striding over the problems measured slower: 0.59 against 0.62 on 23^3 x 8 blocks, and a workgroup that does not fit
  // next to the others at launch waits for a whole stride of problems)
  else if (beta0) hipLaunchKernelGGL((gemm_f32_ragged_kernel<W, NP, R2, false, false>), dim3(a.nbatch), dim3(64 * W), lds_bytes, st, a, c);
  else hipLaunchKernelGGL((gemm_f32_ragged_kernel<W, NP, R2, true, false>), dim3(a.nbatch), dim3(64 * W), lds_bytes, st, a, c);
}
// the bf16 blocked kernel: 2-D batch of 64 x 64 x K problems (K % 32 == 0) whose grid divides into 4 x 4 problems, VNNI-2 A, flat B, plain
// epilogue, beta = 0, strided forms, 16-byte aligned rows / columns, at least one block
static int f32_dma_mode() {   // LIBXSMM_HIP_F32_DMA: 0 never, 1 (default) 64x64 tiles, 2 also 32x32 tiles
  constexpr int mode = 1;
  return mode;
}
// B columns 16-byte aligned, A rows dword aligned (always), every offset inside one tile below 4 GiB
// the bf16 forms kernel: whole 32-tiles, every row of both operand tiles a 16-byte aligned 64- / 128-byte run, strided forms (alignment decidable here)
static bool bf16_forms_ok(const GemmArgs& a, int& af, int& bf) {
  constexpr bool off = false;
  if (off || a.a_type != LIBXSMM_DATATYPE_BF16 || a.b_type != LIBXSMM_DATATYPE_BF16 || (a.c_type != LIBXSMM_DATATYPE_BF16 && a.c_type != LIBXSMM_DATATYPE_F32)) return false;
  if ((a.m % 32) || (a.n % 32) || (a.k % 32) || a.k <= 0 || a.comp_f16) return false;
  const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B, va = a.flags & LIBXSMM_GEMM_FLAG_VNNI_A, vb = a.flags & LIBXSMM_GEMM_FLAG_VNNI_B;
  if ((va && ta) || (vb && !tb)) return false;
  // (the reference packs k by A's VNNI factor: with a non-VNNI A the VNNI_B flag changes nothing, B is plainly transposed [ref: gemm ref :2134, :2157])
  af = va ? AF_VNNI : (ta ? AF_TRANS : AF_FLAT); bf = !tb ? BF_FLAT : ((vb && va) ? BF_TVNNI : BF_TRANS);
  if (a.vnni_c && (a.c_type != LIBXSMM_DATATYPE_BF16 || !(a.flags & LIBXSMM_GEMM_FLAG_BETA_0) || a.colbias || a.act)) return false;
  if (af == AF_VNNI && bf == BF_FLAT && !a.vnni_c) return false;                  // the fast form has its own kernels
  if ((a.list_a && !a.lists_aligned16) || a.br_mode == 1 || a.br_mode == 2) return false;
  unsigned long long bits = (unsigned long long)((long long)a.lda * (af == AF_VNNI ? 4 : 2)) | (unsigned long long)((long long)a.ldb * (bf == BF_TVNNI ? 4 : 2));
  if (!a.list_a) bits |= (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b;
  if (a.br_mode == 3) bits |= (unsigned long long)a.br_stride_a | (unsigned long long)a.br_stride_b;
  if (bits & 15ull) return false;
  if (a.vnni_c && ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)a.bs_c2) & 3ull) != 0)) return false;
  return a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22);
}
template <int AF, int BF> static void launch_bf16_forms_b(const GemmArgs& a, dim3 grid, hipStream_t st) {
  if (a.vnni_c) hipLaunchKernelGGL((gemm_bf16_forms_kernel<AF, BF, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_bf16_forms_kernel<AF, BF, false>), grid, dim3(256), 0, st, a);
}
template <int AF> static void launch_bf16_forms_a(const GemmArgs& a, int bf, dim3 grid, hipStream_t st) {
  if (bf == BF_FLAT) launch_bf16_forms_b<AF, BF_FLAT>(a, grid, st); else if (bf == BF_TRANS) launch_bf16_forms_b<AF, BF_TRANS>(a, grid, st); else launch_bf16_forms_b<AF, BF_TVNNI>(a, grid, st);
}

static bool bf16_stream_ok(const GemmArgs& a) {
  constexpr bool off = false;
  if (off || a.list_a || a.br_mode == 1 || a.br_mode == 2) return false;
  const unsigned long long bits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? a.br_stride_b : 0) |
    (unsigned long long)((long long)a.ldb * 2);
  if (bits & 15ull) return false;
  const unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | (unsigned long long)(a.br_mode == 3 ? a.br_stride_a : 0);
  if (abits & 3ull) return false;
  return (long long)a.lda * a.k * 2 < (1ll << 31) && (long long)a.ldb * a.n * 2 < (1ll << 31);
}

// ------------------------------------------------------------------------------------------------
// LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK: A = (non-zeros in memory order, one bit per element).  Row s of the bit matrix (the
// k-group s: m * kb elements) starts at byte s * row_bytes.  Three small passes expand it to the dense image the dense kernels read:
// set bits per row -> exclusive scan over the rows -> one workgroup per row places its values (or zeros).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bitmask_row_count_kernel(const unsigned char* bitmap_, unsigned int* count, int row_bytes) {
  GM const unsigned char* row = (GM const unsigned char*)bitmap_ + (long long)blockIdx.x * row_bytes;
  unsigned int c = 0;
  for (int b = threadIdx.x; b < row_bytes; b += 64) c += (unsigned int)__builtin_popcount((unsigned int)row[b]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (threadIdx.x == 0) ((GM unsigned int*)count)[blockIdx.x] = c;
}
__global__ __launch_bounds__(1024) void bitmask_row_scan_kernel(const unsigned int* count_, unsigned int* start_, int rows) {
  __shared__ unsigned int part[1024];
  GM const unsigned int* count = (GM const unsigned int*)count_; GM unsigned int* start = (GM unsigned int*)start_;
  const int per = (rows + 1023) / 1024, r0 = (int)threadIdx.x * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  unsigned int sum = 0;
  for (int r = r0; r < r1; ++r) sum += count[r];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                     // inclusive scan of the 1024 partial sums
    const unsigned int v = (threadIdx.x >= (unsigned int)o) ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned int at = part[threadIdx.x] - sum;
  for (int r = r0; r < r1; ++r) { start[r] = at; at += count[r]; }
}
// One workgroup per bit row; the row is walked in segments of 256 bitmap bytes with thread t on byte t: its 8 elements leave as ONE 16- / 32-byte store, so a wave
// writes 1 / 2 KiB of the dense image contiguously (round 3 -- before, a thread walked row_bytes / 256 consecutive bytes and its 2- / 4-byte stores landed 128+
// bytes apart from its neighbours': 8192 x 8192 bf16 took about a millisecond).  Ranks: popcount of the byte, wave-wide inclusive scan by shuffles, the four wave
// totals through LDS, a running carry from segment to segment.
template <typename T>
__global__ __launch_bounds__(256) void bitmask_expand_kernel(const unsigned char* bitmap_, const T* vals_, T* dense_, const unsigned int* start_, int row_bytes) {
  __shared__ unsigned int wave_sum[2][4];
  GM const unsigned char* row = (GM const unsigned char*)bitmap_ + (long long)blockIdx.x * row_bytes;
  GM const T* vals = (GM const T*)vals_ + ((GM const unsigned int*)start_)[blockIdx.x];
  GM T* out = (GM T*)dense_ + (long long)blockIdx.x * row_bytes * 8;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned int carry = 0;
  for (int seg = 0, par = 0; seg < row_bytes; seg += 256, par ^= 1) {
    const int b = seg + t;
    const unsigned int bits = b < row_bytes ? (unsigned int)row[b] : 0u;
    const unsigned int cnt = (unsigned int)__builtin_popcount(bits);
    unsigned int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int v = (unsigned int)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wave_sum[par][wave] = incl;
    __syncthreads();                                   // (two buffers: the next segment's totals cannot overwrite what a slower wave still reads)
    unsigned int before = carry + incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const unsigned int ws = wave_sum[par][w]; if (w < wave) before += ws; total += ws; }
    carry += total;
    if (b < row_bytes) {
      T v[8];
      unsigned int at = before;
#pragma unroll
      for (int e = 0; e < 8; ++e) { T x = (T)0; if ((bits >> e) & 1u) x = vals[at++]; v[e] = x; }      // (only set bits read: the value array ends with its last non-zero)
      typedef T Tx8 __attribute__((ext_vector_type(8)));
      Tx8 o8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = v[e];
      *(GM Tx8*)(out + (long long)b * 8) = o8;
    }
  }
}
int launch_bitmask_expand(const void* bitmap, const void* vals, void* dense, unsigned int* rows_scratch, int rows, int row_bytes, int elem_size, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  unsigned int* count = rows_scratch; unsigned int* start = rows_scratch + rows;
  hipLaunchKernelGGL(bitmask_row_count_kernel, dim3((unsigned int)rows), dim3(64), 0, st, (const unsigned char*)bitmap, count, row_bytes);
  hipLaunchKernelGGL(bitmask_row_scan_kernel, dim3(1), dim3(1024), 0, st, count, start, rows);
  if (elem_size == 4) hipLaunchKernelGGL(bitmask_expand_kernel<unsigned int>, dim3((unsigned int)rows), dim3(256), 0, st, (const unsigned char*)bitmap, (const unsigned int*)vals, (unsigned int*)dense, start, row_bytes);
  else hipLaunchKernelGGL(bitmask_expand_kernel<unsigned short>, dim3((unsigned int)rows), dim3(256), 0, st, (const unsigned char*)bitmap, (const unsigned short*)vals, (unsigned short*)dense, start, row_bytes);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Bitmask-compressed A WITHOUT the dense image (round 3): 16-bit operands, m % 16 == 0, k % 64 == 0, B columns 16-byte aligned.  What travels from HBM is
// what the caller stored -- the non-zeros and one bit per element -- so a weight matrix pruned to density d costs (d + 1/16) of its dense bytes instead of
// (d + 1/16) + 2 (the image written by bitmask_expand_kernel and read back by the dense kernel).
//   pass 1 (bitmask_tile_prefix_kernel): per (bit row r = k-pair, tile of 128 rows of A): set bits of row r before the tile; per row: its total
//   pass 2 (bitmask_row_scan_kernel, shared with the expanding path): exclusive scan of the totals = where row r's values start
//   pass 3 (gemm_bitmask16_kernel): one workgroup per (128 rows of A, 64 columns of C, slice of k).  Per 64-deep chunk (32 bit rows): every wave fetches the
//     values of its 8 bit rows as ONE 8-byte request per lane and row (a window over the row's <= 256 values that starts AT its first value) into LDS; thread
//     (bit row kp, dword p of the row's 8 bitmap dwords) ranks its 32 bits (popcount of the bits below, plus the exclusive prefix over the 8 lanes of its row)
//     and expands them four at a time -- a 16-entry table indexed by the nibble gives the byte selectors of two v_perm_b32 over the three dwords around the
//     nibble's values -- into 16 VNNI-2 words of the dense chunk image [32 k-pairs][128 rows]; B's chunk [64 columns][64 k] lands in LDS with its 16-byte
//     pieces XOR-swizzled; wave w multiplies rows 32 w .. + 31 by the 64 columns on v_mfma_f32_32x32x16_bf16 / _f16.  Requests run ahead of the chunk being
//     multiplied: bitmap dword / value offsets 2 D chunks, value windows and B pieces D chunks (they need the ranks), in rings of D register sets.
//   Measured (8192 x 8192 @50 %, 64 columns): 40 us for this kernel, 0.95 TB/s of compressed bytes -- 512 bytes of LDS traffic per lane and chunk (stage, table,
//   picks, image, operand reads) against 41 bytes from HBM: the expansion through LDS bounds it (LDS 44 % busy, half of that bank conflicts of the
//   data-dependent picks; vector unit 21 %), not the memory system.
//   C is small next to A (m x n against m x k), so the chip is filled by slicing k: every slice writes an f32 partial tile, brsplit_reduce_kernel adds the
//   slices in order and applies beta / the output type (one slice: the kernel writes C itself).  f32 accumulation in the matrix core's order: the
//   tolerance of the dense kernels, not bitwise the reference's serial chain.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bitmask_tile_prefix_kernel(const unsigned char* bitmap_, unsigned int* tile_prefix_, unsigned int* count_, int row_bytes, int tiles) {
  __shared__ unsigned int part[256];
  GM const unsigned char* row = (GM const unsigned char*)bitmap_ + (long long)blockIdx.x * row_bytes;
  const int t = (int)threadIdx.x;
  unsigned int c = 0;
  if (t < tiles) {
    const int b0 = t * 32, b1 = (b0 + 32 < row_bytes) ? b0 + 32 : row_bytes;        // 128 rows x 2 bits = 32 bytes; row_bytes is a multiple of 4
    for (int b = b0; b < b1; b += 4) c += (unsigned int)__builtin_popcount(*(GM const unsigned int*)(row + b));
  }
  part[t] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned int v = (t >= o) ? part[t - o] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  if (t < tiles) ((GM unsigned int*)tile_prefix_)[(long long)blockIdx.x * tiles + t] = part[t] - c;
  if (t == 255) ((GM unsigned int*)count_)[blockIdx.x] = part[255];
}

typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));
template <bool F16, int D, int WGS>
__global__ __launch_bounds__(256, WGS) void gemm_bitmask16_kernel(GemmArgs p, const unsigned char* bitmap_, const unsigned int* start_, const unsigned int* tile_prefix_, const unsigned int* count_, int tiles, int row_bytes,
                                                                int chunks_per_slice, float* partial) {
  constexpr int kStageRow = 528;                                            // bytes per staged value window: 64 lanes x 8 bytes (+ 16: rows land in different banks)
  __shared__ __attribute__((aligned(16))) unsigned int a_img[32 * 128];     // [k-pair][row] VNNI-2 words
  __shared__ __attribute__((aligned(16))) char b_img[64 * 128];             // [column][64 k], 16-byte pieces XOR-swizzled with the column
  __shared__ __attribute__((aligned(16))) char stage[32 * kStageRow];
  __shared__ __attribute__((aligned(16))) u32x2b lut[16];                   // per 4-bit pattern: the byte selectors of the two output words
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  if (tid < 16) {
    unsigned int selw[2] = {0u, 0u}, rank = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned int bit = ((unsigned int)tid >> e) & 1u;
      const unsigned int two = bit ? ((2u * rank) | ((2u * rank + 1u) << 8)) : 0x0c0cu;      // source bytes of this half, or two zero bytes
      selw[e >> 1] |= two << (16 * (e & 1));
      rank += bit;
    }
    lut[tid] = (u32x2b){selw[0], selw[1]};
  }
  const int kp_l = tid >> 3, piece = tid & 7;                           // this thread's bit row of the chunk and bitmap dword of the tile
  const int tm = (int)blockIdx.x, i0 = tm * 128, j0 = (int)blockIdx.y * 64;
  const bool word_ok = (i0 / 4 + piece * 4) < row_bytes;               // the tile's last dwords may lie past the row (m not a multiple of 128)
  GM const unsigned char* bitmap = (GM const unsigned char*)bitmap_;
  GM const unsigned int* start = (GM const unsigned int*)start_; GM const unsigned int* tpre = (GM const unsigned int*)tile_prefix_;
  const unsigned long long vals0 = (unsigned long long)(size_t)p.a;
  const int chunks = p.k >> 6;
  // one past the last non-zero (the tables know), as an offset from the 8-byte boundary at or below the array; the last position an 8-byte request may start at
  const unsigned long long base0 = vals0 & ~7ull;
  const unsigned int row_off0 = (unsigned int)(vals0 - base0);
  const unsigned int end_off = row_off0 + 2u * (start[p.k / 2 - 1] + ((GM const unsigned int*)count_)[p.k / 2 - 1]);
  const unsigned int last_off = end_off >= 8u ? end_off - 8u : 0u;
  const int cbeg = (int)blockIdx.z * chunks_per_slice, cend = (cbeg + chunks_per_slice < chunks) ? cbeg + chunks_per_slice : chunks;
  const unsigned int ldb = (unsigned int)p.ldb;
  // B: thread -> (column f = tid / 4, 16-byte pieces tid % 4 and tid % 4 + 4) of the chunk
  const int bf_ = tid >> 2, bpc = tid & 3;
  GM const char* bsrc = (GM const char*)p.b + ((long long)((j0 + bf_ < p.n) ? j0 + bf_ : p.n - 1) * ldb) * 2 + bpc * 16;
  f32x16 acc[2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
  // What is loaded is kept raw until it is consumed one stage later: an operation on a value right behind its load (a select, an add, a copy at the end of a
  // conditional block) makes the compiler wait there, and every load is unconditional (an address that is always valid instead of a branch around the load):
  // a load inside divergent control flow makes it wait for ALL outstanding loads at the next use of any of them.  Either would end the requests that run ahead.
  struct Bits { unsigned int wv, s0, t0; int c; };
  struct Prep { unsigned int w, before, shifts; u32x4 b[2]; u32x2b win[8]; };
  auto load_bits = [&](int c_in) __attribute__((always_inline)) {
    Bits x;
    const int c = c_in < cend ? c_in : cend - 1;                                          // past the slice: the last chunk again (never consumed)
    const long long r = (long long)c * 32 + kp_l;
    x.c = c;
    x.wv = *(GM const unsigned int*)(bitmap + r * row_bytes + (word_ok ? i0 / 4 + piece * 4 : 0));
    x.s0 = start[r]; x.t0 = tpre[r * tiles + tm];
    return x;
  };
  auto prepare = [&](const Bits& x) __attribute__((always_inline)) {
    Prep q;
    q.w = word_ok ? x.wv : 0u; q.shifts = 0u;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) q.b[qq] = *(GM const u32x4*)(bsrc + 128ll * x.c + 64 * qq);      // (columns past n read column n - 1: their products are never stored)
    const unsigned int base = x.s0 + x.t0;
    // ranks: bits of this row below this thread's dword (exclusive prefix over the 8 lanes of the row); the row's total ends up in its last lane
    const unsigned int cnt = (unsigned int)__builtin_popcount(q.w);
    unsigned int incl = cnt;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { const unsigned int v = (unsigned int)__shfl_up((int)incl, o, 8); if (piece >= o) incl += v; }
    q.before = incl - cnt;
    // the value windows of the wave's eight bit rows start AT the row's first value: 8-byte requests on 2-byte boundaries (the memory pipeline serves unaligned
    // global accesses), so 64 lanes cover the 256 values a row can have; lanes past the row's values read the start of the value array (always mapped; what
    // they stage is never picked)
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const unsigned int rbase = (unsigned int)__builtin_amdgcn_readlane((int)base, 8 * rr), rtot = (unsigned int)__builtin_amdgcn_readlane((int)incl, 8 * rr + 7);
      // offsets from the 8-byte boundary at or below the value array (< 4 GiB: m k < 2^31).  A request that would end past the LAST non-zero of the whole
      // array (only the last non-empty rows can) is moved back to end there and its bytes shifted down: nothing outside the caller's array is ever read,
      // without a branch (a branch around a load would end the requests that run ahead, see above)
      const unsigned int want = (rtot ? row_off0 + 2u * rbase : row_off0) + (((unsigned int)lane * 4u < rtot) ? (unsigned int)lane * 8u : 0u);
      const unsigned int at = want < last_off ? want : last_off;
      q.win[rr] = *(GM const u32x2b*)((GM const char*)(size_t)base0 + at);              // kept raw: the shift happens where the window is consumed
      q.shifts |= ((want - at) >> 1) << (2 * rr);                                       // 0 / 2 / 4 / 6 bytes, two bits per row
    }
    return q;
  };
  auto process = [&](const Prep& cur) __attribute__((always_inline)) {
    wg_barrier();                                                                   // the previous chunk's MFMAs are done with the images (raw barrier: the requests ahead stay in flight)
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const unsigned long long xs = (((unsigned long long)cur.win[rr][1] << 32) | cur.win[rr][0]) >> (16u * ((cur.shifts >> (2 * rr)) & 3u));
      *(u32x2b*)(stage + (8 * wave + rr) * kStageRow + lane * 8) = (u32x2b){(unsigned int)xs, (unsigned int)(xs >> 32)};
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *(u32x4*)(b_img + bf_ * 128 + (((bpc + 4 * q) ^ (bf_ & 7)) * 16)) = cur.b[q];
    // Four bits (two VNNI words) at a time: the values of a nibble are <= 4 consecutive halves from rank r on.  Three aligned dwords around them, shifted by
    // 16 bits when r is odd, are the 8 source bytes of two v_perm_b32 whose selectors -- which source half goes to which half of which word, zero bytes
    // where the bit is clear -- come from a 16-entry table indexed by the nibble: 15 instructions per four elements instead of ~44 with one pick per element.
    const char* winb = stage + kp_l * kStageRow;
    const unsigned int w = cur.w;
    unsigned int out[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned int nib = (w >> (4 * q)) & 15u;
      const u32x2b sel = lut[nib];
      const unsigned int r = cur.before + (unsigned int)__builtin_popcount(w & ((1u << (4 * q)) - 1u));
      const unsigned int* t = (const unsigned int*)(winb + ((r >> 1) << 2));
      const unsigned int t0 = t[0], t1 = t[1], t2 = t[2], sh = (r & 1u) << 4;
      const unsigned int d0 = __builtin_amdgcn_alignbit(t1, t0, sh), d1 = __builtin_amdgcn_alignbit(t2, t1, sh);
      out[2 * q] = __builtin_amdgcn_perm(d1, d0, sel[0]);
      out[2 * q + 1] = __builtin_amdgcn_perm(d1, d0, sel[1]);
    }
#pragma unroll
    // image rows are rotated by 4 (row % 8) words: the eight bit rows a wave writes at once would otherwise start in the same bank (a row is 128 words)
    for (int x = 0; x < 4; ++x) { const u32x4 v = {out[4 * x], out[4 * x + 1], out[4 * x + 2], out[4 * x + 3]}; *(u32x4*)(a_img + kp_l * 128 + ((piece * 16 + 4 * x + 4 * (kp_l & 7)) & 127)) = v; }
    wg_barrier();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      u32x4 af, bfr[2];
#pragma unroll
      for (int d = 0; d < 4; ++d) af[d] = a_img[(8 * s2 + 4 * h + d) * 128 + ((32 * wave + li + 4 * (4 * h + d)) & 127)];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) { const int f = 32 * nt + li; bfr[nt] = *(const u32x4*)(b_img + f * 128 + (((2 * s2 + h) ^ (f & 7)) * 16)); }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma_16bit<F16>(bfr[nt], af, acc[nt]);
    }
  };
  if (cbeg >= cend) return;                                                              // (cannot happen: the launcher sizes the slices)
  Prep ring[D]; Bits bits[D];
  // prologue: windows of the first D chunks, bits of the D after them (indices past the slice repeat its last chunk and are never consumed).  In the loop the
  // bits of chunk c + 2 D are requested while chunk c is multiplied and turned into window requests D chunks later: both stages are D chunks deep
  static_for<D>([&](auto uc) { bits[uc.value] = load_bits(cbeg + uc.value); });
  static_for<D>([&](auto uc) { ring[uc.value] = prepare(bits[uc.value]); bits[uc.value] = load_bits(cbeg + D + uc.value); });
  int c0 = cbeg;
  for (; c0 + D <= cend; c0 += D) {
    static_for<D>([&](auto uc) {
      constexpr int u = uc.value;
      const Prep cur = ring[u];
      ring[u] = prepare(bits[u]);                 // chunk c0 + u + D
      bits[u] = load_bits(c0 + u + 2 * D);
      __builtin_amdgcn_sched_barrier(0);          // the requests go out BEFORE this chunk's work (left alone the scheduler sinks them below it)
      process(cur);
    });
  }
  static_for<D>([&](auto uc) { if (c0 + uc.value < cend) process(ring[uc.value]); });
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const int i = i0 + 32 * wave + li;
  if (i >= p.m) return;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
      const int j = j0 + 32 * nt + jl_of(r2, h);
      if (j >= p.n) continue;
      const float v = acc[nt][r2];
      if (partial) { ((GM float*)partial)[((long long)blockIdx.z * p.n + j) * p.m + i] = v; continue; }
      const long long e = (long long)j * p.ldc + i;
      if (p.c_type == LIBXSMM_DATATYPE_F32) { GM float* cp = (GM float*)p.c + e; *cp = beta0 ? v : v + *cp; }
      else if (F16) { GM _Float16* cp = (GM _Float16*)p.c + e; *cp = (_Float16)(beta0 ? v : v + (float)*cp); }
      else { GM unsigned short* cp = (GM unsigned short*)p.c + e; *cp = f32_to_bf16_rne(beta0 ? v : v + bf16_to_f32(*cp)); }
    }
  }
}

// *taken = 1: launched (the return value is the launch status); 0: shape / alignment not taken (the caller expands A and runs the dense kernel).
// scratch: rows * (tiles + 2) words of counts / offsets, followed by the f32 partial tiles of the k-slices.
int launch_gemm_bitmask16(const GemmArgs& a, const void* bitmap, unsigned int* scratch, size_t scratch_bytes, void* stream, const char** name, int* taken) {
  hipStream_t st = (hipStream_t)stream;
  *taken = 0;
  constexpr bool off = false;
  const bool t16 = (a.a_type == LIBXSMM_DATATYPE_BF16 || a.a_type == LIBXSMM_DATATYPE_F16) && a.b_type == a.a_type;
  // (every 64-column tile of C expands A again: beyond a few tiles the dense image, built once, is the cheaper form)
  if (off || !t16 || (a.m % 16) || (a.k % 64) || a.m <= 0 || a.n <= 0 || a.n > 256 || a.k <= 0) return 0;
  if ((((size_t)a.b) & 15) || (((long long)a.ldb * 2) & 15) || (((size_t)a.a) & 1) || (((size_t)bitmap) & 3)) return 0;
  const int rows = a.k / 2, row_bytes = a.m / 4, tiles = (a.m + 127) / 128, chunks = a.k / 64;
  if (tiles > 256 || (long long)a.m * a.k >= (1ll << 31)) return 0;
  const size_t table_words = ((size_t)rows * (size_t)(tiles + 2) + 63) & ~(size_t)63;
  // slices of k: about 768 workgroups on the chip (three per CU; 256 ... 4096 measured within 20 % of each other), at least four chunks per slice, the partial tiles inside the scratch
  const long long wgs = (long long)tiles * ((a.n + 63) / 64);
  constexpr long long want = 768ll;
  long long slices = std::max<long long>(1, std::min<long long>(want / std::max<long long>(wgs, 1), chunks / 4));
  const size_t tile_bytes = (size_t)a.m * (size_t)a.n * sizeof(float);
  while (slices > 1 && table_words * 4 + (size_t)slices * tile_bytes > scratch_bytes) --slices;
  if (table_words * 4 > scratch_bytes) return 0;
  const int cps = (int)((chunks + slices - 1) / slices);
  slices = (chunks + cps - 1) / cps;
  unsigned int* count = scratch; unsigned int* start = scratch + rows; unsigned int* tpre = scratch + 2 * (size_t)rows;
  float* partial = slices > 1 ? (float*)(scratch + table_words) : nullptr;
  const dim3 grid((unsigned int)tiles, (unsigned int)((a.n + 63) / 64), (unsigned int)slices);
  // (A second form -- every wave owning 32 rows outright: wave-private stage and image, B fragments straight from global memory, no barrier in the loop, a 16-bit
  // table per 32-row tile -- was written and measured in round 3: correct, 136 registers, and SLOWER, 118 / 82 us against 81 / 62 us on 8192^2 @50 % x 64 / 16
  // columns: the barriers are not what bounds this kernel.  Removed.)
  hipLaunchKernelGGL(bitmask_tile_prefix_kernel, dim3((unsigned int)rows), dim3(256), 0, st, (const unsigned char*)bitmap, tpre, count, row_bytes, tiles);
  hipLaunchKernelGGL(bitmask_row_scan_kernel, dim3(1), dim3(1024), 0, st, count, start, rows);
  // ring depth 2 at three workgroups per CU (162 registers) measured 81 us against 89 us for depth 3 at two (206 registers) on 8192 x 8192 @50 %, n = 64
  if (a.a_type == LIBXSMM_DATATYPE_F16) hipLaunchKernelGGL((gemm_bitmask16_kernel<true, 2, 3>), grid, dim3(256), 0, st, a, (const unsigned char*)bitmap, start, tpre, count, tiles, row_bytes, cps, partial);
  else hipLaunchKernelGGL((gemm_bitmask16_kernel<false, 2, 3>), grid, dim3(256), 0, st, a, (const unsigned char*)bitmap, start, tpre, count, tiles, row_bytes, cps, partial);
  int err = (int)hipGetLastError();
  if (err == 0 && slices > 1) err = launch_brsplit_reduce(a, partial, (int)slices, st);
  if (name) *name = "gemm_bitmask16_kernel";
  *taken = 1;
  return err;
}

// libxsmm_hip_probe_mfma: the matrix pipe alone (register operands, 16 accumulators per wave, one wave per SIMD): the power roof of a data set
template <bool BF16>
__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(const void* operands, float* sink, int iterations) {
  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  GM const u32x4* ops = (GM const u32x4*)operands;                 // 4096 x 16 bytes
  u32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = ops[(threadIdx.x + 256 * i + 37 * blockIdx.x) & 4095]; b[i] = ops[(threadIdx.x + 256 * (i + 4) + 37 * blockIdx.x) & 4095]; }
  for (int it = 0; it < iterations; ++it) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        if constexpr (BF16) acc[ti * 4 + tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[tj]), __builtin_bit_cast(bf16x8, a[ti]), acc[ti * 4 + tj], 0, 0, 0);
        else acc[ti * 4 + tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(b[tj][0]), __uint_as_float(a[ti][0]), acc[ti * 4 + tj], 0, 0, 0);
      }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;                                // keeps the accumulators alive; never true for finite sums of this size in practice
}
int launch_mfma_probe(int bf16, const void* operands, int iterations, void* stream, double* flop) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  static float* sink = nullptr;
  if (!sink && hipMalloc((void**)&sink, 256) != hipSuccess) return (int)hipGetLastError();
  if (bf16) hipLaunchKernelGGL(mfma_probe_kernel<true>, dim3((unsigned int)cus), dim3(256), 0, (hipStream_t)stream, operands, sink, iterations);
  else hipLaunchKernelGGL(mfma_probe_kernel<false>, dim3((unsigned int)cus), dim3(256), 0, (hipStream_t)stream, operands, sink, iterations);
  if (flop) *flop = (double)cus * 4.0 * 16.0 * (double)iterations * (bf16 ? 32768.0 : 4096.0);
  return (int)hipGetLastError();
}

// the partial products of one long f32 chain (see gemm_f32_brchain_kernel); *nslices = partial tiles written; 0 slices = shape not taken
int launch_brchain_f32(const GemmArgs& a_in, float* partial, size_t partial_capacity_tiles, int* nslices, void* stream, const char** kernel_name) {
  *nslices = 0;
  constexpr bool off = false;
  if (off || a_in.a_type != LIBXSMM_DATATYPE_F32 || a_in.b_type != LIBXSMM_DATATYPE_F32 || a_in.br_mode != 3 || (a_in.flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B))) return 0;
  if ((a_in.m % 32) || (a_in.n % 32) || (a_in.k % 32) || a_in.k <= 0 || a_in.m <= 0 || a_in.n <= 0) return 0;
  const unsigned long long bits = (unsigned long long)(size_t)a_in.a | (unsigned long long)(size_t)a_in.b | (unsigned long long)a_in.br_stride_a | (unsigned long long)a_in.br_stride_b |
    (unsigned long long)((long long)a_in.lda * 4) | (unsigned long long)((long long)a_in.ldb * 4);
  if (bits & 15ull) return 0;                                  // 16-byte operand loads
  GemmArgs a = a_in;
  a.tiles_m = a.m / 32; a.tiles_n = a.n / 32;
  const unsigned long long tiles = (unsigned long long)a.tiles_m * a.tiles_n;
  // about 2048 waves over the chip (one workgroup of 8 per CU), chunk blocks per wave, 8 waves per slice: measured on 32^3 chains (round 3): br = 4096 12.6 us with
  // 1024 ... 2731 waves against 14.6 us with >= 4096 (twice the partial tiles for the second kernel) and 16.4 us with 512; br = 65 536: 91 - 103 us with 1024 ... 2048,
  // 102 - 111 us with >= 4096
  constexpr unsigned long long want_waves = 2048ull;
  unsigned long long chunk = (tiles * a.br_count + want_waves - 1ull) / want_waves; if (chunk == 0) chunk = 1;
  const unsigned long long slices = (a.br_count + 8ull * chunk - 1ull) / (8ull * chunk);
  if (slices > partial_capacity_tiles || slices * tiles >= (1ull << 31)) return 0;
  hipLaunchKernelGGL(gemm_f32_brchain_kernel, dim3((unsigned int)(slices * tiles)), dim3(512), 0, (hipStream_t)stream, a, partial, (unsigned int)chunk, (unsigned int)slices);
  if (kernel_name) *kernel_name = "gemm_f32_brchain_kernel";
  *nslices = (int)slices;
  return (int)hipGetLastError();
}

int launch_brsplit_reduce(const GemmArgs& a, const float* partial, int nsplit, void* stream) {
  const long long blocks = (long long)((a.m + 63) / 64) * a.n;
  hipLaunchKernelGGL(brsplit_reduce_kernel, dim3((unsigned int)blocks), dim3(64, 16), 0, (hipStream_t)stream, a, partial, nsplit);
  return (int)hipGetLastError();
}

int launch_gemm(const GemmArgs& a_in, void* stream, const char** kernel_name) {
  const GemmArgs& a0 = a_in;
  hipStream_t st = (hipStream_t)stream;
  if (a0.nbatch == 0 || a0.m <= 0 || a0.n <= 0) { if (kernel_name) *kernel_name = "(empty)"; return 0; }
  if (a0.a_type == LIBXSMM_DATATYPE_F64 && a0.b_type == LIBXSMM_DATATYPE_F64 && a0.c_type == LIBXSMM_DATATYPE_F64) return launch_gemm_f64(a0, stream, kernel_name);   // gemm_f64_kernels.hip
  GemmPlan pl_ = plan_gemm(a0.m, a0.n, a0.k, a0.flags, a0.a_type, a0.b_type, a0.c_type, a0.vnni_c);
  if (a0.comp_f16) pl_ = GemmPlan{P_GENERIC, false};              // a rounding after every product: only the generic kernel does that
  const GemmPlan pl = pl_;
  if ((long long)((a0.m + 15) / 16) * ((a0.n + 15) / 16) * (long long)a0.nbatch >= (1ll << 31)) return (int)hipErrorInvalidValue;
  if (kernel_name) *kernel_name = path_name(pl.path);
  GemmArgs a = a_in;
  auto wave_grid = [&](int tm, int tn) {
    a.tiles_m = (a.m + tm - 1) / tm; a.tiles_n = (a.n + tn - 1) / tn;
    a.map2d_shift = 0;
    if (a.batch_inner && a.tiles_m * a.tiles_n == 1) {           // 2-D batch, one tile per element: deal compact super-tiles to the XCDs
      constexpr int want = 16;
      const unsigned int ni = a.batch_inner, nj = a.nbatch / a.batch_inner;
      for (int t = want; t >= 8 && !a.map2d_shift; t >>= 1)
        if (ni % t == 0 && nj % t == 0 && ((ni / t) * (nj / t)) % 8 == 0) a.map2d_shift = t == 32 ? 5 : (t == 16 ? 4 : 3);
    }
    const long long tiles = (long long)a.tiles_m * a.tiles_n * (long long)a.nbatch;
    return dim3((unsigned int)((tiles + 3) / 4));
  };
  dim3 grid;
  // BF32 on whole 32 x 32 x 32 tiles: the f32 streaming kernel with the operands rounded to bf16 in registers (bit-identical to the reference loop)
  if (a.a_type == LIBXSMM_DATATYPE_BF32 && a.b_type == LIBXSMM_DATATYPE_BF32 && a.c_type == LIBXSMM_DATATYPE_F32 && (a.m % 32) == 0 && (a.n % 32) == 0 && (a.k % 32) == 0 && a.k > 0 &&
      !a.vnni_c && !(a.flags & (LIBXSMM_GEMM_FLAG_VNNI_A | LIBXSMM_GEMM_FLAG_VNNI_B)) && operands_aligned16(a, 4) && (!a.colbias || ((((unsigned long long)(size_t)a.d | (unsigned long long)a.bs_d) & 3ull) == 0ull)) &&
      a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22) &&
      ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)a.bs_c2) & 3ull) == 0ull)) {
    const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B;
    grid = wave_grid(32, 32);
    if (kernel_name) *kernel_name = "gemm_bf32_stream_kernel";
    if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_stream_kernel<false, false, true>), grid, dim3(256), 0, st, a);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_stream_kernel<true, false, true>), grid, dim3(256), 0, st, a);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_stream_kernel<false, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_f32_stream_kernel<true, true, true>), grid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
  }
  // 2-D batches of f32 16 x 16 x K tiles with a plain epilogue: the blocked kernel on 2 x 2 sub-blocks
  if (a.batch_inner && f32_blocked16_ok(a)) {
    a.tiles_m = a.tiles_n = 1; a.map2d_shift = 0;
    grid = dim3((a.batch_inner / 8u) * ((a.nbatch / a.batch_inner) / 8u));
    if (kernel_name) *kernel_name = "gemm_f32_blocked16_kernel";
    hipLaunchKernelGGL(gemm_f32_blocked16_kernel, grid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
  }
  // 2-D batches of bf16 16^3 / 32 x 32 x K / 64 x 64 x K problems, plain epilogue: the 256 x 256 macro-tile kernel
  { bool k16 = false;
    if (a.batch_inner && bf16_macro_ok(a, k16)) {
      const bool f32c = a.c_type == LIBXSMM_DATATYPE_F32;
      const bool pack2 = ((a.ldc & 1) == 0) && ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)a.bs_c2) & 3ull) == 0ull);
      const int form = f32c ? 0 : (pack2 ? 1 : 2);
      using kfn = void (*)(GemmArgs);
#define BM_(F_) { gemm_bf16_macro_kernel<F_, 64, false>, gemm_bf16_macro_kernel<F_, 32, false>, gemm_bf16_macro_kernel<F_, 16, false>, gemm_bf16_macro_kernel<F_, 16, true> }
      static const kfn table[3][4] = { BM_(0), BM_(1), BM_(2) };
#undef BM_
      static const bool lds_ok = []() {
        for (int f = 0; f < 3; ++f) for (int v = 0; v < 4; ++v)
          if (hipFuncSetAttribute((const void*)table[f][v], hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return false;
        return true; }();
      if (lds_ok) {
        const unsigned int ppm = 256u / (unsigned int)a.m;
        a.tiles_m = a.tiles_n = 1; a.map2d_shift = 0;
        const dim3 mgrid((a.batch_inner / ppm) * ((a.nbatch / a.batch_inner) / ppm));
        if (kernel_name) *kernel_name = "gemm_bf16_macro_kernel";
        const int v = a.m == 64 ? 0 : (a.m == 32 ? 1 : (k16 ? 3 : 2));
        if (a.a_type == LIBXSMM_DATATYPE_F16) {       // packed f16 or f32 C (element-wise f16 stores: the single-problem kernels)
          if (form == 2) goto no_macro;
          using k16fn = void (*)(GemmArgs);
          static const k16fn f16_table[2][2] = {{gemm_bf16_macro_kernel<0, 64, false, 4, 4, 4, 0, true>, gemm_bf16_macro_kernel<0, 32, false, 4, 4, 4, 0, true>},
                                                {gemm_bf16_macro_kernel<1, 64, false, 4, 4, 4, 0, true>, gemm_bf16_macro_kernel<1, 32, false, 4, 4, 4, 0, true>}};
          static const bool f16_ok = []() {
            for (int f = 0; f < 2; ++f)
              for (int w = 0; w < 2; ++w) { if (hipFuncSetAttribute((const void*)f16_table[f][w], hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return false; }
            return true; }();
          if (!f16_ok) { (void)hipGetLastError(); goto no_macro; }
          if (kernel_name) *kernel_name = "gemm_f16_macro_kernel";
          hipLaunchKernelGGL(f16_table[form][v], mgrid, dim3(256), 131072, st, a);
          return (int)hipGetLastError();
        }
#ifdef LIBXSMM_HIP_EXPERIMENTS      // make -C libxsmm_amd/csrc EXPERIMENTS=1: the ablations / wave layouts of profiles/r03_bf16_macro_ablation.txt (tools/bb_ablate.sh)
        static const int abl = []() { const char* e = getenv("LIBXSMM_HIP_BB_ABL"); return e ? atoi(e) : 0; }();      // timing experiments (wrong results)
        static const int shape = []() { const char* e = getenv("LIBXSMM_HIP_BM_SHAPE"); return e ? atoi(e) : 0; }();   // experiments: 824 = 8 waves of 2 x 4 tiles, 842 = 4 x 2
        if ((abl || shape) && form == 1 && v == 0) {
#define BM_VAR_(NW_, TI_, TJ_, V_) { static const bool ok = hipFuncSetAttribute((const void*)gemm_bf16_macro_kernel<1, 64, false, NW_, TI_, TJ_, V_>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) == hipSuccess; \
            if (ok) { hipLaunchKernelGGL((gemm_bf16_macro_kernel<1, 64, false, NW_, TI_, TJ_, V_>), mgrid, dim3(64 * NW_), 131072, st, a); return (int)hipGetLastError(); } }
          if (shape == 824 && abl == 0) BM_VAR_(8, 2, 4, 0)
          else if (shape == 842 && abl == 0) BM_VAR_(8, 4, 2, 0)
          else if (shape == 824 && abl == 3) BM_VAR_(8, 2, 4, 3)
          else if (shape == 0 && abl == 1) BM_VAR_(4, 4, 4, 1)
          else if (shape == 0 && abl == 2) BM_VAR_(4, 4, 4, 2)
          else if (shape == 0 && abl == 3) BM_VAR_(4, 4, 4, 3)
#undef BM_VAR_
        }
#endif
        hipLaunchKernelGGL(table[form][v], mgrid, dim3(256), 131072, st, a);
        return (int)hipGetLastError();
      }
      (void)hipGetLastError();
    }
  }
  no_macro:
  // 16 x 16 problems with a plain epilogue: four problems per wave
  if (p16_ok(a)) {
    const bool bf16 = a.a_type == LIBXSMM_DATATYPE_BF16;
    a.tiles_m = a.tiles_n = 1; a.map2d_shift = 0;
    // round 4: A through a wave-private LDS image, one 16-byte request per lane (gemm_small_kernels.hip); needs 16-byte aligned A rows
    { int taken = 0;
      const int e16 = launch_gemm_p16w(a, stream_nt(a, bf16 ? 2 : 4, a.c_type == LIBXSMM_DATATYPE_F32 ? 4 : 2), stream, kernel_name, &taken);
      if (taken) return e16; }
    // small launches are bound by their own latency: keep one problem per wave there (4x the waves), four per wave from 16384 problems on
    const bool four = a.nbatch >= 16384u;
    grid = dim3((unsigned int)((a.nbatch + (four ? 15u : 3u)) / (four ? 16u : 4u)));
    if (kernel_name) *kernel_name = bf16 ? "gemm_bf16_p16_kernel" : "gemm_f32_p16_kernel";
    if (bf16) { if (four) hipLaunchKernelGGL((gemm_p16_kernel<4, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_p16_kernel<1, true>), grid, dim3(256), 0, st, a); }
    else { if (four) hipLaunchKernelGGL((gemm_p16_kernel<4, false>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_p16_kernel<1, false>), grid, dim3(256), 0, st, a); }
    return (int)hipGetLastError();
  }
  // 2-D batches of exact f32 32^3 / 64^3 problems (K any multiple of 32), NN, strided: the workgroup-cooperative blocked kernel
  if (a.batch_inner && (pl.path == P_F32_1x1 || pl.path == P_F32_2x2) && f32_blocked_ok(a)) {
    const int mb = a.m / 32, ppw = 4 / mb;
    const unsigned int ni = a.batch_inner, nj = a.nbatch / a.batch_inner;
    a.tiles_m = a.tiles_n = mb; a.map2d_shift = 0;
    grid = dim3((ni / ppw) * (nj / ppw));
    if (mb == 1) { if (kernel_name) *kernel_name = "gemm_f32_blocked_kernel<1>"; hipLaunchKernelGGL((gemm_f32_blocked_kernel<1>), grid, dim3(256), 0, st, a); }
    else { if (kernel_name) *kernel_name = "gemm_f32_blocked_kernel<2>"; hipLaunchKernelGGL((gemm_f32_blocked_kernel<2>), grid, dim3(256), 0, st, a); }
    return (int)hipGetLastError();
  }
  // f32 shapes that are not whole 32 / 64 tiles, plain epilogue: persistent workgroups, operands staged in registers, 16 x 16 MFMA tiles
  // (round 5: shapes of several WHOLE 16-tiles that are not whole 32-tiles -- 48^3, 32 x 48 ... -- ran a wave per 16-tile, nine waves per 48^3 problem that each fetched
  //  their own panels: 0.49 of the HBM roofline.  The one-problem-per-workgroup kernel takes them like 40^3 and 56^3: 48^3 0.49 -> 0.73, batch 4096 0.41 -> 0.56, beta = 1
  //  0.53 -> 0.70 (profiles/r05_t16_ragged.jsonl); LIBXSMM_HIP_T16_RAGGED=0: the wave-per-tile kernel)
  constexpr bool t16_ragged = true;
  // (whole 32-tiles that are several per problem -- 96^3, 128^3 -- stay a wave per tile: on gemm_wgp_f32_kernel 96^3 0.47 -> 0.49, 128^3 0.47 -> 0.37, r05_t16_ragged.jsonl:
  //  the f32 matrix pipe is as busy as the memory there)
  if (((pl.path == P_F32_1x1 || pl.path == P_F32_2x2) && !pl.exact) || (t16_ragged && pl.path == P_F32_T16 && (a.m > 16 || a.n > 16))) {
    RaggedCfg rc; int waves = 1, np = 1, rounds = 0; unsigned int lds_bytes = 0; bool single = false;
    if (f32_ragged_plan(a, rc, waves, np, rounds, lds_bytes, single)) {
      a.tiles_m = a.tiles_n = 1; a.map2d_shift = 0;
      if (kernel_name) *kernel_name = "gemm_f32_ragged_kernel";
      if (waves == 1) {                                      // rounds == ragged_rounds_cap(...): the LDS plan is sized for exactly this instantiation
        if (rounds == 4) launch_ragged<1, 1, 4>(a, rc, lds_bytes, single, st);
        else if (rounds == 8 && np == 1) launch_ragged<1, 1, 8>(a, rc, lds_bytes, single, st);
        else if (rounds == 8) launch_ragged<1, 2, 8>(a, rc, lds_bytes, single, st);
        else if (rounds == 12) launch_ragged<1, 2, 12>(a, rc, lds_bytes, single, st);
        else launch_ragged<1, 2, 17>(a, rc, lds_bytes, single, st);
      } else {
        if (rounds == 7) launch_ragged<4, 2, 7>(a, rc, lds_bytes, single, st);
        else if (rounds == 10 && np == 2) launch_ragged<4, 2, 10>(a, rc, lds_bytes, single, st);
        else if (rounds == 10) launch_ragged<4, 4, 10>(a, rc, lds_bytes, single, st);
        else if (rounds == 14 && np <= 4) launch_ragged<4, 4, 14>(a, rc, lds_bytes, single, st);
        else if (rounds == 14) launch_ragged<4, 6, 14>(a, rc, lds_bytes, single, st);
        else if (np <= 4) launch_ragged<4, 4, 24>(a, rc, lds_bytes, single, st);
        else launch_ragged<4, 6, 24>(a, rc, lds_bytes, single, st);
      }
      return (int)hipGetLastError();
    }
    // round 5: what that plan does not hold (72^3 with beta = 1: three images next to the operands; deep K with wide n ...) ran on the wave-per-tile kernel at 0.19 of
    // the roofline.  When rows of A and columns of B are whole 16-byte pieces: the blocks by LDS-DMA, K in chunks, one problem per workgroup (gemm_wgp_f32_kernels.hip:
    // 0.43 - 0.48 there).  Where both apply the register-staged kernel is the faster one (72^3 0.51 against 0.41, 40^3 0.66 against 0.56: these shapes are as much
    // matrix-pipe as memory bound, profiles/r05_wgp_f32.jsonl) and keeps its shapes.
    { int taken = 0; const int e = launch_gemm_wgp_f32(a, stream, kernel_name, &taken); if (taken) return e; }
  }
  // bf16 in the other operand forms (flat / transposed A, transposed / VNNI B, VNNI C): operands re-laid out on the way out of a wave-private LDS image (round 4)
  { int af = 0, bf = 0;
    if (bf16_forms_ok(a, af, bf)) {
      grid = wave_grid(32, 32);
      if (kernel_name) *kernel_name = "gemm_bf16_forms_kernel";
      if (af == AF_VNNI) launch_bf16_forms_a<AF_VNNI>(a, bf, grid, st); else if (af == AF_FLAT) launch_bf16_forms_a<AF_FLAT>(a, bf, grid, st); else launch_bf16_forms_a<AF_TRANS>(a, bf, grid, st);
      return (int)hipGetLastError();
    } }
  // IEEE halves take the bf16 fast paths with f32 accumulation and f16 or f32 C (round 6: any beta and the fused operators too -- the start value is rounded to f16 on
  // its way into the accumulators, tile_init<.., F16S>)
  const bool f16_fast = a.a_type == LIBXSMM_DATATYPE_F16 && a.b_type == LIBXSMM_DATATYPE_F16 && !a.comp_f16 && !a.vnni_c &&
    (a.c_type == LIBXSMM_DATATYPE_F16 || a.c_type == LIBXSMM_DATATYPE_F32);
  // 8-bit operands on the masked matrix-core kernel: any shape with whole k-quads, any alignment, any batch-reduce form.  Returns false for what it does not
  // take (more than 2^31 bytes inside one operand).
  auto launch_m8 = [&](bool big) -> bool {
    constexpr int tile_env = 0;      // experiments: 1 forces 32 x 32 tiles, 2 forces 64 x 64
    if (tile_env == 1) big = false; else if (tile_env == 2) big = true;
    const bool fp8 = a.a_type == LIBXSMM_DATATYPE_BF8 || a.a_type == LIBXSMM_DATATYPE_HF8;
    if (fp8 && a.c_type != LIBXSMM_DATATYPE_F32 && a.vnni_c) return false;
    if ((a.k & 3) || a.k <= 0) return false;
    const bool ua = a.a_type == LIBXSMM_DATATYPE_U8, ub = a.b_type == LIBXSMM_DATATYPE_U8;
    // packed operand blocks of several tiles: one problem per workgroup, whole problem in LDS (gemm_wgp8_kernels.hip, round 5); an error of that launch is left
    // for the hipGetLastError() behind the switch
    { int taken = 0; (void)launch_gemm_wgp8(a, fp8 ? (a.a_type == LIBXSMM_DATATYPE_HF8 ? 2 : 1) : 0, ua, ub, stream, kernel_name, &taken); if (taken) return true; }
    grid = big ? wave_grid(64, 64) : wave_grid(32, 32);
    if (kernel_name) *kernel_name = big ? "gemm_mfma_8bit_kernel<2,2>" : "gemm_mfma_8bit_kernel<1,1>";
    // strided forms whose blocks, rows and columns start on dwords: B through LDS (see the kernel); LIBXSMM_HIP_M8_LDS=0 keeps B in registers (measurement switch)
    constexpr bool lds_off = false;
    const unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
      (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0) | (unsigned long long)a.ldb;
    const bool bl = !lds_off && !a.list_a && a.br_mode != 1 && a.br_mode != 2 && (abits & 3ull) == 0ull &&
      (unsigned long long)a.n * (unsigned long long)a.ldb < (1ull << 31) && (unsigned long long)a.k * (unsigned long long)a.lda < (1ull << 31);
#define LAUNCH_M8K_(MT_, NT_, K_, UA_, UB_) do { if (bl) hipLaunchKernelGGL((gemm_mfma_8bit_kernel<MT_, NT_, K_, UA_, UB_, true>), grid, dim3(256), 0, st, a); \
                                                 else hipLaunchKernelGGL((gemm_mfma_8bit_kernel<MT_, NT_, K_, UA_, UB_, false>), grid, dim3(256), 0, st, a); } while (0)
#define LAUNCH_M8_(MT_, NT_) do { \
      if (fp8) { if (a.a_type == LIBXSMM_DATATYPE_HF8) LAUNCH_M8K_(MT_, NT_, 2, false, false); else LAUNCH_M8K_(MT_, NT_, 1, false, false); } \
      else if (!ua && !ub) LAUNCH_M8K_(MT_, NT_, 0, false, false); \
      else if (ua && !ub) LAUNCH_M8K_(MT_, NT_, 0, true, false); \
      else if (!ua && ub) LAUNCH_M8K_(MT_, NT_, 0, false, true); \
      else LAUNCH_M8K_(MT_, NT_, 0, true, true); } while (0)
    if (big) LAUNCH_M8_(2, 2); else LAUNCH_M8_(1, 1);
#undef LAUNCH_M8_
#undef LAUNCH_M8K_
    return true;
  };
  switch (pl.path) {
    case P_W8_1x1: case P_W8_2x2: {
      const bool big = pl.path == P_W8_2x2, va8 = (a.flags & LIBXSMM_GEMM_FLAG_VNNI_A) != 0;
      const int kind = a.a_type == LIBXSMM_DATATYPE_I8 ? 4 : ((a.a_type == LIBXSMM_DATATYPE_HF8 ? 1 : 0) + (va8 ? 0 : 2));
      // ragged / several-tile shapes with a packed weight block: one problem per workgroup out of LDS (gemm_wgp16_kernels.hip, round 5)
      // (8-bit FLOAT weights also as whole 64-tiles: 64^3 0.66 -> 0.74; int8 weights with row scales stay with the LDS-B kernel there: 0.70 against 0.68)
      if ((a.m % 64) || (a.n % 64) || (a.k % 32) || kind != 4) { int taken = 0; const int e = launch_gemm_wgp16_w8(a, kind, stream, kernel_name, &taken); if (taken) return e; }
      grid = big ? wave_grid(64, 64) : wave_grid(32, 32);
      const int tw = big ? 64 : 32;
      const unsigned long long bbits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? a.br_stride_b : 0) | (unsigned long long)((long long)a.ldb * 2);
      const bool exact = (a.m % tw) == 0 && (a.n % tw) == 0 && (a.k % 32) == 0 && (bbits & 15ull) == 0 && !a.list_a && a.br_mode != 1 && a.br_mode != 2 &&
        (kind >= 2 || ((((unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | (unsigned long long)(a.br_mode == 3 ? a.br_stride_a : 0)) & 1ull) == 0));
      const bool bl = !exact && !(a.k & 1) && ragged16_b_dwords(a);        // ragged shapes with B on dwords: B through LDS
      const bool xlds = true;      // whole tiles: B through LDS by 16-byte requests (round 5: adopted, 64^3 bf8 0.605 -> 0.658, scaled i8 0.628 -> 0.659: profiles/r05_ragged16_switches.jsonl)
#define LAUNCH_W8K_(MT_, NT_, K_) do { if (exact && xlds) hipLaunchKernelGGL((gemm_w8_bf16_kernel<MT_, NT_, K_, true, true>), grid, dim3(256), 0, st, a); \
                                       else if (exact) hipLaunchKernelGGL((gemm_w8_bf16_kernel<MT_, NT_, K_, true>), grid, dim3(256), 0, st, a); \
                                       else if (bl) hipLaunchKernelGGL((gemm_w8_bf16_kernel<MT_, NT_, K_, false, true>), grid, dim3(256), 0, st, a); \
                                       else hipLaunchKernelGGL((gemm_w8_bf16_kernel<MT_, NT_, K_, false>), grid, dim3(256), 0, st, a); } while (0)
#define LAUNCH_W8_(MT_, NT_) do { switch (kind) { case 0: LAUNCH_W8K_(MT_, NT_, 0); break; case 1: LAUNCH_W8K_(MT_, NT_, 1); break; case 2: LAUNCH_W8K_(MT_, NT_, 2); break; \
                                                  case 3: LAUNCH_W8K_(MT_, NT_, 3); break; default: LAUNCH_W8K_(MT_, NT_, 4); break; } } while (0)
      if (big) LAUNCH_W8_(2, 2); else LAUNCH_W8_(1, 1);
#undef LAUNCH_W8_
#undef LAUNCH_W8K_
      break;
    }
    case P_M8_1x1: case P_M8_2x2: {
      if (launch_m8(pl.path == P_M8_2x2)) break;
      if (kernel_name) *kernel_name = "gemm_generic_kernel";
      const long long gblocks = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * (long long)a.nbatch;
      hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned int)gblocks), dim3(64, 4), 0, st, a);
      break;
    }
    case P_F32_T16: grid = wave_grid(16, 16); hipLaunchKernelGGL(gemm_mfma_f32_t16_kernel, grid, dim3(256), 0, st, a); break;
    case P_F32_1x1:
      grid = wave_grid(32, 32);
      if (pl.exact && operands_aligned16(a, 4) && f32_dma_mode() >= 2 && a.lda < (1 << 22) && a.ldb < (1 << 22)) {
        const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B;
        if (kernel_name) *kernel_name = "gemm_f32_dma_kernel<1,1>";
        if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_dma_kernel<1, 1, false, false>), grid, dim3(256), 0, st, a);
        else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_dma_kernel<1, 1, true, false>), grid, dim3(256), 0, st, a);
        else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_dma_kernel<1, 1, false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_f32_dma_kernel<1, 1, true, true>), grid, dim3(256), 0, st, a);
      }
      else if (pl.exact && operands_aligned16(a, 4) && f32_lean_ok(a)) {
        const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B;
        LeanF32Args la;
        la.a = a.a; la.b = a.b; la.c = a.c; la.bs_a = a.bs_a; la.bs_b = a.bs_b; la.bs_c = a.bs_c;
        la.brs_a = a.br_mode == 3 ? a.br_stride_a : 0; la.brs_b = a.br_mode == 3 ? a.br_stride_b : 0;
        la.nbatch = a.nbatch; la.kchunks = (unsigned int)a.k >> 5; la.nchunks = (unsigned int)a.br_count * la.kchunks;
        la.lda = (unsigned int)a.lda; la.ldb = (unsigned int)a.ldb; la.ldc = (unsigned int)a.ldc;
        if (kernel_name) *kernel_name = "gemm_f32_stream_kernel_lean";
        // cache policy (see the kernel).  Streaming hint of the calling thread (libxsmm_hip_set_streaming_hint): 2 = operands are read once
        // from HBM -> nt loads; 1 = operands are re-read / cache resident -> never nt; 0 (default) = decide by size: a launch that moves
        // more than the Infinity Cache holds cannot have resident operands.  LIBXSMM_HIP_F32_POLICY=0|1|2 forces a kernel variant.
        constexpr int pol_env = -1;
        const unsigned long long footprint = (unsigned long long)a.nbatch * (a.br_count * (unsigned long long)(a.k) * 256ull + 4096ull);
        const bool c16 = ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)((long long)a.ldc * 4)) & 15ull) == 0ull);
        int pol = stream_nt(a, 4, 4) ? 1 : 0;
        (void)footprint;
        if (pol == 1 && c16) pol = 3;                        // nt loads AND whole 16-byte stores
        if (pol_env >= 0 && pol_env <= 3) pol = pol_env;
        if ((pol == 0 || pol == 3) && !c16) pol = pol == 0 ? 2 : 1;
#define LAUNCH_LEAN__(TA_, TB_, S_, P_) hipLaunchKernelGGL((gemm_f32_stream_kernel_lean<TA_, TB_, S_, P_>), grid, dim3(256), 0, st, la)
#define LAUNCH_LEAN_S_(TA_, TB_, S_) do { if (pol == 0) LAUNCH_LEAN__(TA_, TB_, S_, 0); else if (pol == 1) LAUNCH_LEAN__(TA_, TB_, S_, 1); else if (pol == 3) LAUNCH_LEAN__(TA_, TB_, S_, 3); else LAUNCH_LEAN__(TA_, TB_, S_, 2); } while (0)
#define LAUNCH_LEAN_(TA_, TB_) do { if (la.nchunks == 1) LAUNCH_LEAN_S_(TA_, TB_, true); else LAUNCH_LEAN_S_(TA_, TB_, false); } while (0)
        if (!ta && !tb) LAUNCH_LEAN_(false, false); else if (ta && !tb) LAUNCH_LEAN_(true, false); else if (!ta && tb) LAUNCH_LEAN_(false, true); else LAUNCH_LEAN_(true, true);
#undef LAUNCH_LEAN_S_
#undef LAUNCH_LEAN__
#undef LAUNCH_LEAN_
      }
      else if (pl.exact && operands_aligned16(a, 4) && a.lda < (1 << 22) && a.ldb < (1 << 22) && a.ldc < (1 << 22)) {
        const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B;
        if (kernel_name) *kernel_name = "gemm_f32_stream_kernel";
        if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_stream_kernel<false, false>), grid, dim3(256), 0, st, a);
        else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_stream_kernel<true, false>), grid, dim3(256), 0, st, a);
        else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_stream_kernel<false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_f32_stream_kernel<true, true>), grid, dim3(256), 0, st, a);
      }
      else if (pl.exact && operands_aligned16(a, 4)) launch_f32<1, 1, GM_STAGED>(a, grid, st);
      else if (pl.exact) launch_f32<1, 1, GM_EXACT>(a, grid, st);
      else if (f32_blob_ok(a)) { if (kernel_name) *kernel_name = "gemm_f32_blob_kernel"; hipLaunchKernelGGL(gemm_f32_blob_kernel, grid, dim3(256), 0, st, a); }
      else launch_f32<1, 1, GM_MASKED>(a, grid, st);
      break;
    case P_F32_2x2:
      grid = wave_grid(64, 64);
      if (pl.exact && a.m == 64 && a.n == 64 && a.bs_b == 0 && f32_wg64_ok(a) && operands_aligned16(a, 4)) {       // B shared by the batch: persistent workgroups, B's operands in registers
        int taken = 0;
        const int e64 = launch_gemm_f32_wg64_sharedb(a, stream_nt(a, 4, 4), stream, kernel_name, &taken);
        if (taken) return e64;
      }
      if (pl.exact && a.m == 64 && a.n == 64 && f32_wg64_ok(a) && operands_aligned16(a, 4)) {
        a.map2d_shift = 0;                                  // one problem per workgroup: the super-tile dealing counts four problems per workgroup
        grid = dim3(a.nbatch);
        if (kernel_name) *kernel_name = "gemm_f32_wg64_kernel";
        if (stream_nt(a, 4, 4)) hipLaunchKernelGGL((gemm_f32_wg64_kernel<2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_f32_wg64_kernel<0>), grid, dim3(256), 0, st, a);
        break;
      }
      if (pl.exact && operands_aligned16(a, 4) && f32_dma_mode() >= 1 && a.lda < (1 << 22) && a.ldb < (1 << 22)) {
        const bool ta = a.flags & LIBXSMM_GEMM_FLAG_TRANS_A, tb = a.flags & LIBXSMM_GEMM_FLAG_TRANS_B;
        if (kernel_name) *kernel_name = "gemm_f32_dma_kernel<2,2>";
        if (!ta && !tb && stream_nt(a, 4, 4)) hipLaunchKernelGGL((gemm_f32_dma_kernel<2, 2, false, false, 2>), grid, dim3(256), 0, st, a);
        else if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_dma_kernel<2, 2, false, false>), grid, dim3(256), 0, st, a);
        else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_dma_kernel<2, 2, true, false>), grid, dim3(256), 0, st, a);
        else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_dma_kernel<2, 2, false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_f32_dma_kernel<2, 2, true, true>), grid, dim3(256), 0, st, a);
      }
      else if (pl.exact && operands_aligned16(a, 4)) launch_f32<2, 2, GM_STAGED>(a, grid, st);
      else if (pl.exact) launch_f32<2, 2, GM_EXACT>(a, grid, st);
      else launch_f32<2, 2, GM_MASKED>(a, grid, st);
      break;
    case P_BF16_1x1:
      // gemm_wgp.hpp (round 5): ragged shapes, and whole 32-tiles that are several tiles per problem (96^3 as nine waves that each fetched their own panels: 0.39)
      { int taken = 0; const int e = launch_gemm_wgp16(a, stream, kernel_name, &taken); if (taken) return e; }
      grid = wave_grid(32, 32);
      if (a.a_type == LIBXSMM_DATATYPE_F16 && f16_fast && pl.exact && bf16_stream_ok(a)) {       // the bf16 streaming kernel on IEEE halves
        if (kernel_name) *kernel_name = "gemm_f16_stream_kernel<1,1>";
        if (stream_nt(a, 2, typesize_c(a))) hipLaunchKernelGGL((gemm_bf16_stream_kernel<1, 1, 2, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_stream_kernel<1, 1, 0, true>), grid, dim3(256), 0, st, a);
        break;
      }
      if (a.a_type == LIBXSMM_DATATYPE_F16) {
        if (kernel_name) *kernel_name = "gemm_mfma_f16_kernel<1,1>";
        if (pl.exact) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, true, true>), grid, dim3(256), 0, st, a);
        else if (ragged16_b_dwords(a) && ragged16_bounded(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, false, true, true, true>), grid, dim3(256), 0, st, a);
        else if (ragged16_b_dwords(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, false, true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, false, true>), grid, dim3(256), 0, st, a);
        break;
      }
      if (pl.exact && bf16_stream_ok(a)) {
        if (kernel_name) *kernel_name = "gemm_bf16_stream_kernel<1,1>";
        if (stream_nt(a, 2, typesize_c(a))) hipLaunchKernelGGL((gemm_bf16_stream_kernel<1, 1, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_stream_kernel<1, 1, 0>), grid, dim3(256), 0, st, a);
      }
      else if (pl.exact) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, true>), grid, dim3(256), 0, st, a);
      else if (ragged16_b_dwords(a) && ragged16_bounded(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, false, false, true, true>), grid, dim3(256), 0, st, a);
      else if (ragged16_b_dwords(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, false, false, true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((gemm_mfma_bf16_kernel<1, 1, false>), grid, dim3(256), 0, st, a);
      break;
    case P_BF16_2x2:
      // (whole 64-tiles stay with the 64^3-per-workgroup kernel: 0.735 against 0.753 at 65 536 problems but 0.61 against 0.53 at 4096, profiles/r05_wgp_pair.jsonl)
      if (!pl.exact) { int taken = 0; const int e = launch_gemm_wgp16(a, stream, kernel_name, &taken); if (taken) return e; }       // gemm_wgp.hpp (round 5)
      grid = wave_grid(64, 64);
      if (pl.exact && a.m == 64 && a.n == 64 && !a.batch_inner && (a.a_type != LIBXSMM_DATATYPE_F16 || f16_fast)) {       // gemm_w64_kernels.hip (round 6): 64^3, one problem per wave
        int taken = 0; const int e = launch_gemm_16bit_w64(a, stream_nt(a, 2, typesize_c(a)), stream, kernel_name, &taken); if (taken) return e;
      }
      if (a.a_type == LIBXSMM_DATATYPE_F16 && f16_fast && pl.exact && a.m == 64 && a.n == 64 && !a.batch_inner && bf16_wg64_ok(a)) {
        a.map2d_shift = 0;
        grid = dim3(a.nbatch);
        if (kernel_name) *kernel_name = "gemm_f16_wg64_kernel";
        hipLaunchKernelGGL((gemm_bf16_wg64_kernel<0, true>), grid, dim3(256), 0, st, a);
        break;
      }
      if (a.a_type == LIBXSMM_DATATYPE_F16 && f16_fast && pl.exact && bf16_stream_ok(a)) {
        if (kernel_name) *kernel_name = "gemm_f16_stream_kernel<2,2>";
        hipLaunchKernelGGL((gemm_bf16_stream_kernel<2, 2, 0, true>), grid, dim3(256), 0, st, a);
        break;
      }
      if (a.a_type == LIBXSMM_DATATYPE_F16) {
        if (kernel_name) *kernel_name = "gemm_mfma_f16_kernel<2,2>";
        if (pl.exact) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, true, true>), grid, dim3(256), 0, st, a);
        else if (ragged16_b_dwords(a) && ragged16_bounded(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, false, true, true, true>), grid, dim3(256), 0, st, a);
        else if (ragged16_b_dwords(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, false, true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, false, true>), grid, dim3(256), 0, st, a);
        break;
      }
      if (pl.exact && a.m == 64 && a.n == 64 && !a.batch_inner && bf16_wg64_ok(a)) {
        a.map2d_shift = 0;
        grid = dim3(a.nbatch);
        if (kernel_name) *kernel_name = "gemm_bf16_wg64_kernel";
        // cacheable loads whatever the size: non-temporal loads measured slower on this kernel (batch 65536 from HBM: 0.71 vs 0.75; batch 4096: 0.59 vs 0.61)
        hipLaunchKernelGGL((gemm_bf16_wg64_kernel<0>), grid, dim3(256), 0, st, a);
        break;
      }
      if (pl.exact && bf16_stream_ok(a)) {
        if (kernel_name) *kernel_name = "gemm_bf16_stream_kernel<2,2>";
        // measured (batch 65536, operands from HBM): nt on the B stream takes the 32^3 kernel from 0.72 to 0.84 of the HBM roofline but the
        // 64^3 kernel from 0.68 to 0.60 -- so only the <1,1> form asks for it
        hipLaunchKernelGGL((gemm_bf16_stream_kernel<2, 2, 0>), grid, dim3(256), 0, st, a);
      }
      else if (pl.exact) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, true>), grid, dim3(256), 0, st, a);
      else if (ragged16_b_dwords(a) && ragged16_bounded(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, false, false, true, true>), grid, dim3(256), 0, st, a);
      else if (ragged16_b_dwords(a)) hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, false, false, true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((gemm_mfma_bf16_kernel<2, 2, false>), grid, dim3(256), 0, st, a);
      break;
    case P_MXMX_1x1: case P_MXMX_2x2: {
      // operands are read as dwords, scales as bytes: any alignment of the leading dimensions works; bases dword aligned; offsets < 4 GiB
      const unsigned long long bits = (unsigned long long)(size_t)a.a | (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_a | (unsigned long long)a.bs_b |
        (unsigned long long)(a.br_mode == 3 ? (a.br_stride_a | a.br_stride_b) : 0);
      const bool ok = !a.list_a && (bits & 3ull) == 0 && (long long)a.lda * a.k < (1ll << 31) && (long long)a.ldb * a.k < (1ll << 31);
      if (ok) {
        if (a.a_type == LIBXSMM_DATATYPE_MXHF6) {           // E2M3 on the matrix cores (the plan sends E3M2 to the exact kernel, see plan_gemm)
          constexpr bool small6 = false;
          const bool big6 = pl.path == P_MXMX_2x2 && !small6;
          const long long lim = (1ll << 31) / 3;
          if ((long long)a.lda * (a.k / 4) < lim && (long long)a.ldb * (a.k / 4) < lim) {
            constexpr int gather6 = 0;
            a.tune = gather6;
            grid = big6 ? wave_grid(64, 64) : wave_grid(32, 32);
            if (kernel_name) *kernel_name = big6 ? "gemm_mx6_stream_kernel<2,2>" : "gemm_mx6_stream_kernel<1,1>";
            if (big6) hipLaunchKernelGGL((gemm_mx_stream_kernel<2, 2, 2>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((gemm_mx_stream_kernel<1, 1, 2>), grid, dim3(256), 0, st, a);
            break;
          }
          if (kernel_name) *kernel_name = "gemm_generic_kernel";
          const long long gb6 = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * (long long)a.nbatch;
          hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned int)gb6), dim3(64, 4), 0, st, a);
          break;
        }
        const int fmt = a.a_type == LIBXSMM_DATATYPE_MXFP4X2 ? 4 : (a.a_type == LIBXSMM_DATATYPE_MXBF8 ? 1 : 0);
        const bool big = pl.path == P_MXMX_2x2;
        grid = big ? wave_grid(64, 64) : wave_grid(32, 32);
#define LAUNCH_MX_(MT_, NT_) do { \
          if (fmt == 4) hipLaunchKernelGGL((gemm_mx_stream_kernel<MT_, NT_, 4>), grid, dim3(256), 0, st, a); \
          else if (fmt == 1) hipLaunchKernelGGL((gemm_mx_stream_kernel<MT_, NT_, 1>), grid, dim3(256), 0, st, a); \
          else hipLaunchKernelGGL((gemm_mx_stream_kernel<MT_, NT_, 0>), grid, dim3(256), 0, st, a); } while (0)
        if (big) LAUNCH_MX_(2, 2); else LAUNCH_MX_(1, 1);
#undef LAUNCH_MX_
        break;
      }
      if (kernel_name) *kernel_name = "gemm_generic_kernel";
      const long long gblocks = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * (long long)a.nbatch;
      hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned int)gblocks), dim3(64, 4), 0, st, a);
      break;
    }
    case P_MX4_1x1: case P_MX4_2x2: {
      // B columns 16-byte aligned (LDS-DMA); A and the scales are read byte-wise (any alignment); offsets inside a tile < 4 GiB
      const unsigned long long bits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? a.br_stride_b : 0) | (unsigned long long)((long long)a.ldb * 2);
      const bool ok = !a.list_a && a.br_mode != 1 && a.br_mode != 2 && (bits & 15ull) == 0 && (long long)a.lda * a.k < (1ll << 31) && (long long)a.ldb * a.n * 2 < (1ll << 31);
      if (ok) {
        if (pl.path == P_MX4_2x2) { grid = wave_grid(64, 64); hipLaunchKernelGGL((gemm_mxfp4_stream_kernel<2, 2>), grid, dim3(256), 0, st, a); }
        else { grid = wave_grid(32, 32); hipLaunchKernelGGL((gemm_mxfp4_stream_kernel<1, 1>), grid, dim3(256), 0, st, a); }
        break;
      }
      if (kernel_name) *kernel_name = "gemm_generic_kernel";
      const long long gblocks = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * (long long)a.nbatch;
      hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned int)gblocks), dim3(64, 4), 0, st, a);
      break;
    }
    case P_FP8_1x1: case P_FP8_2x2: {
      const unsigned long long bits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? a.br_stride_b : 0) | (unsigned long long)a.ldb;
      const unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | (unsigned long long)(a.br_mode == 3 ? a.br_stride_a : 0);
      const bool ok = !a.list_a && a.br_mode != 1 && a.br_mode != 2 && (bits & 15ull) == 0 && (abits & 3ull) == 0 && (long long)a.lda * a.k < (1ll << 31) && (long long)a.ldb * a.n < (1ll << 31);
      if (ok) {
        const bool hf8 = a.a_type == LIBXSMM_DATATYPE_HF8, big = pl.path == P_FP8_2x2;
        // whole 32-tiles, several per problem (96^3: nine): one problem per workgroup out of LDS instead of nine waves that each fetch their own panels (gemm_wgp8_kernels.hip)
        // (also whole 64-tiles -- 64^3 was one wave per problem: bf8 0.64 -> 0.76, i8 0.65 -> 0.75, profiles/r05_wgp_pair.jsonl; more than twelve tiles are not taken there)
        if (a.m > 32 && a.n > 32 && (a.m / 32) * (a.n / 32) <= 12) { int taken = 0; (void)launch_gemm_wgp8(a, hf8 ? 2 : 1, false, false, stream, kernel_name, &taken); if (taken) break; }      // (128^3: 0.66 here against 0.43 there)
        grid = big ? wave_grid(64, 64) : wave_grid(32, 32);
        if (a.c_type != LIBXSMM_DATATYPE_F32) {            // C in the operands' type
          if (a.vnni_c) goto fp8_generic;
          if (kernel_name) *kernel_name = big ? "gemm_fp8c8_stream_kernel<2,2>" : "gemm_fp8c8_stream_kernel<1,1>";
          if (big) { if (hf8) hipLaunchKernelGGL((gemm_fp8_stream_kernel<2, 2, true, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_fp8_stream_kernel<2, 2, false, true>), grid, dim3(256), 0, st, a); }
          else { if (hf8) hipLaunchKernelGGL((gemm_fp8_stream_kernel<1, 1, true, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_fp8_stream_kernel<1, 1, false, true>), grid, dim3(256), 0, st, a); }
          break;
        }
        if (big) { if (hf8) hipLaunchKernelGGL((gemm_fp8_stream_kernel<2, 2, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_fp8_stream_kernel<2, 2, false>), grid, dim3(256), 0, st, a); }
        else { if (hf8) hipLaunchKernelGGL((gemm_fp8_stream_kernel<1, 1, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_fp8_stream_kernel<1, 1, false>), grid, dim3(256), 0, st, a); }
        break;
      }
      fp8_generic:
      if (launch_m8(pl.path == P_FP8_2x2)) break;            // unaligned operands, pointer / offset lists: the masked matrix-core kernel
      if (kernel_name) *kernel_name = "gemm_generic_kernel";
      const long long gblocks = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * (long long)a.nbatch;
      hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned int)gblocks), dim3(64, 4), 0, st, a);
      break;
    }
    case P_MX4I8_1x1: case P_MX4I8_2x2: {
      const unsigned long long bits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? a.br_stride_b : 0) | (unsigned long long)a.ldb;
      const unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | (unsigned long long)(a.br_mode == 3 ? a.br_stride_a : 0) |
        (unsigned long long)(size_t)a.b_scf | (unsigned long long)a.bs_bscf;
      const bool ok = !a.list_a && a.br_mode != 1 && a.br_mode != 2 && (bits & 15ull) == 0 && (abits & 3ull) == 0 && (long long)a.lda * a.k < (1ll << 31) && (long long)a.ldb * a.n < (1ll << 31) &&
        (a.c_type == LIBXSMM_DATATYPE_F32 || a.c_type == LIBXSMM_DATATYPE_BF16) && a.a_scf && a.b_scf;
      if (ok) {
        // waves that walk several consecutive tiles with the next chunk's operands in flight: at least 8 K waves per launch, at most 8 tiles each
        constexpr int pipe = 8;
        a.tiles_m = a.m / 32; a.tiles_n = a.n / 32; a.map2d_shift = 0;
        const unsigned long long tiles = (unsigned long long)a.tiles_m * a.tiles_n * a.nbatch;
        const bool c_ok = (a.flags & LIBXSMM_GEMM_FLAG_BETA_0) == 0 || ((((unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c | (unsigned long long)a.bs_c2) & 15ull) == 0 &&
          ((unsigned long long)a.ldc * (a.c_type == LIBXSMM_DATATYPE_F32 ? 4ull : 2ull)) % 16ull == 0 && !a.list_c);
        if (pipe > 0 && c_ok && tiles < (1ull << 31) && (a.k >> 6) * a.br_count < (1ull << 31)) {
          const unsigned int per_wave = (unsigned int)std::min<unsigned long long>((unsigned long long)pipe, std::max<unsigned long long>(1ull, tiles / 8192ull));
          const unsigned long long waves = (tiles + per_wave - 1) / per_wave;
          // two chunks in flight per wave at three waves per SIMD; four / six at two waves per SIMD measured the same or worse (the kernel is bound by its
          // ~300 vector instructions per tile -- code expansion, convert / scale / add of 2 x 16 block sums -- not by the operand round trip any more)
          hipLaunchKernelGGL((gemm_mx4i8_pipe_kernel<2>), dim3((unsigned int)((waves + 3) / 4)), dim3(256), 0, st, a, per_wave, (unsigned int)tiles);
          if (kernel_name) *kernel_name = "gemm_mx4i8_pipe_kernel";
          break;
        }
        grid = wave_grid(32, 32);
        hipLaunchKernelGGL((gemm_mx4i8_stream_kernel<1, 1>), grid, dim3(256), 0, st, a);
        break;
      }
      if (kernel_name) *kernel_name = "gemm_generic_kernel";
      const long long gblocks2 = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * (long long)a.nbatch;
      hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned int)gblocks2), dim3(64, 4), 0, st, a);
      break;
    }
    case P_I8_1x1: case P_I8_2x2: {
      // same alignment contract as the bf16 streaming kernel (element size 1): B columns 16-byte aligned, A dword aligned
      const unsigned long long bits = (unsigned long long)(size_t)a.b | (unsigned long long)a.bs_b | (unsigned long long)(a.br_mode == 3 ? a.br_stride_b : 0) | (unsigned long long)a.ldb;
      const unsigned long long abits = (unsigned long long)(size_t)a.a | (unsigned long long)a.bs_a | (unsigned long long)(a.br_mode == 3 ? a.br_stride_a : 0) | (unsigned long long)(size_t)a.c | (unsigned long long)a.bs_c;
      const bool ok = !a.list_a && a.br_mode != 1 && a.br_mode != 2 && (bits & 15ull) == 0 && (abits & 3ull) == 0 && (long long)a.lda * a.k < (1ll << 31) && (long long)a.ldb * a.n < (1ll << 31);
      const bool i4 = a.a_type == LIBXSMM_DATATYPE_I4X2 || a.a_type == LIBXSMM_DATATYPE_U4X2;
      if (ok && i4) {
        const bool big = pl.path == P_I8_2x2;
        grid = big ? wave_grid(64, 64) : wave_grid(32, 32);
        if (kernel_name) *kernel_name = big ? "gemm_i4_stream_kernel<2,2>" : "gemm_i4_stream_kernel<1,1>";
        if (big) hipLaunchKernelGGL((gemm_i8_stream_kernel<2, 2, false, true, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_i8_stream_kernel<1, 1, false, true, 1>), grid, dim3(256), 0, st, a);
        break;
      }
      const int lowbit = a.a_type == LIBXSMM_DATATYPE_I2X4 ? 2 : (a.a_type == LIBXSMM_DATATYPE_I1X8 ? 3 : 0);
      if (ok && lowbit) {
        const bool ub = a.b_type == LIBXSMM_DATATYPE_U8, big = pl.path == P_I8_2x2;
        grid = big ? wave_grid(64, 64) : wave_grid(32, 32);
        if (kernel_name) *kernel_name = lowbit == 2 ? (big ? "gemm_i2_stream_kernel<2,2>" : "gemm_i2_stream_kernel<1,1>") : (big ? "gemm_i1_stream_kernel<2,2>" : "gemm_i1_stream_kernel<1,1>");
#define LAUNCH_LB_(MT_, NT_) do { \
          if (lowbit == 2) { if (ub) hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, false, true, 2>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, false, false, 2>), grid, dim3(256), 0, st, a); } \
          else { if (ub) hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, false, true, 3>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, false, false, 3>), grid, dim3(256), 0, st, a); } } while (0)
        if (big) LAUNCH_LB_(2, 2); else LAUNCH_LB_(1, 1);
#undef LAUNCH_LB_
        break;
      }
      if (ok && !i4 && !lowbit) {
        const bool ua = a.a_type == LIBXSMM_DATATYPE_U8, ub = a.b_type == LIBXSMM_DATATYPE_U8, big = pl.path == P_I8_2x2;
        if (a.m > 32 && a.n > 32 && (a.m / 32) * (a.n / 32) <= 12) { int taken = 0; (void)launch_gemm_wgp8(a, 0, ua, ub, stream, kernel_name, &taken); if (taken) break; }      // (as for the 8-bit floats above)
        grid = big ? wave_grid(64, 64) : wave_grid(32, 32);
#define LAUNCH_I8_(MT_, NT_) do { \
          if (!ua && !ub) hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, false, false>), grid, dim3(256), 0, st, a); \
          else if (ua && !ub) hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, true, false>), grid, dim3(256), 0, st, a); \
          else if (!ua && ub) hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, false, true>), grid, dim3(256), 0, st, a); \
          else hipLaunchKernelGGL((gemm_i8_stream_kernel<MT_, NT_, true, true>), grid, dim3(256), 0, st, a); } while (0)
        if (big) LAUNCH_I8_(2, 2); else LAUNCH_I8_(1, 1);
#undef LAUNCH_I8_
        break;
      }
      if (!i4 && !lowbit && launch_m8(pl.path == P_I8_2x2)) break;       // unaligned operands, pointer / offset lists: the masked matrix-core kernel
      if (kernel_name) *kernel_name = "gemm_generic_kernel";
    }
    // fallthrough
    default: {
      const long long blocks = (long long)((a.m + 63) / 64) * ((a.n + 3) / 4) * (long long)a.nbatch;
      hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned int)blocks), dim3(64, 4), 0, st, a);
    }
  }
  return (int)hipGetLastError();
}

}  // namespace xamd
#endif  // XAMD_GEMM_SHARD
