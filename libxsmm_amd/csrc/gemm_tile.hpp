// gemm_tile.hpp -- the 32 x 32 tile prologue / epilogue shared by the dense matrix-core kernels of every GEMM translation unit (round 5: cut out of gemm_kernels.hip
// so that kernel families can live in translation units of their own): small conversions, C / bias loads, TileCtx, tile_init, tile_store (+ the buffer-addressed
// form), the 16-bit MFMA step, the wave -> (problem, tile) decomposition.  Semantics of the fused epilogue: [ref: src/generator_gemm_reference_impl.c:294-372].
#pragma once
#include "gemm_device.hpp"
#include "bf16_cvt.hpp"

// a*b+c below means two roundings unless fma()/MFMA is spelled out: parity with the reference's C loops depends on it
#pragma clang fp contract(off)

namespace xamd {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short x) { return __uint_as_float((unsigned int)x << 16); }
// RNE with denormals-are-zero and NaN quieting [ref: src/libxsmm_math.c:684-704]
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7f800000u) == 0u) u &= 0x80000000u;
  if ((u & 0x7f800000u) == 0x7f800000u) { if (u & 0x007fffffu) u |= 0x00400000u; }
  else u += 0x00007fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// sigmoid(x) = (tanh(x/2) + 1) / 2 [ref: mateltwise ref :18-20], evaluated as 1 / (1 + e^-x) with the hardware
// exp/rcp: ~1e-6 relative, inside the reference's own 7e-4 bound for fused sigmoid (gemm_kernel.c:5396) and a
// small fraction of the code of an inlined tanhf (the epilogue is instantiated 16x per tile).
__device__ __forceinline__ float act_apply(int act, float x) {
  if (act == 1 || act == 2) return (x <= 0.0f) ? 0.0f : x;
  if (act == 3) return __frcp_rn(1.0f + __expf(-x));
  return x;
}

template <int ACT> __device__ __forceinline__ float act_fixed(float x) {
  if (ACT == 1 || ACT == 2) return (x <= 0.0f) ? 0.0f : x;
  if (ACT == 3) return __frcp_rn(1.0f + __expf(-x));
  return x;
}

// 8-bit floats [ref: src/libxsmm_math.c:546-585]: BF8 (E5M2) is the upper byte of an IEEE half; HF8 (E4M3, bias 7, no infinities)
__device__ __forceinline__ float bf8_to_f32(unsigned char x) { return (float)__builtin_bit_cast(_Float16, (unsigned short)((unsigned short)x << 8)); }
__device__ __forceinline__ float hf8_to_f32(unsigned char in) {
  const unsigned int s = (unsigned int)(in & 0x80u) << 24, e = (in & 0x78u) >> 3;
  unsigned int m = in & 0x07u, e_norm = e + 120u;
  if (e == 0u && m != 0u) { unsigned int lz = 2u; lz = (m > 1u) ? 1u : lz; lz = (m > 3u) ? 0u : lz; e_norm -= lz; m = (m << (lz + 1u)) & 7u; }
  else if (e == 0u && m == 0u) e_norm = 0u;
  else if (e == 15u && m == 7u) { e_norm = 255u; m = 4u; }
  return __uint_as_float((e_norm << 23) | (m << 20) | s);
}
__device__ __forceinline__ float load_as_f32(gcptr base, long long idx, int type) {
  if (type == LIBXSMM_DATATYPE_F32) return ((GM const float*)base)[idx];
  if (type == LIBXSMM_DATATYPE_BF32) return bf16_to_f32(f32_to_bf16_rne(((GM const float*)base)[idx]));      // f32 storage, bf16 precision [ref: gemm ref :1366,:1384-1389]
  if (type == LIBXSMM_DATATYPE_BF8) return bf8_to_f32(((GM const unsigned char*)base)[idx]);
  if (type == LIBXSMM_DATATYPE_HF8) return hf8_to_f32(((GM const unsigned char*)base)[idx]);
  return bf16_to_f32(((GM const unsigned short*)base)[idx]);
}

// C and bias operands are f32 or bf16 only (keeps the 8-bit decoders out of every tile prologue)
__device__ __forceinline__ float load_c_f32(gcptr base, long long idx, int type) {
  if (type == LIBXSMM_DATATYPE_F16) return (float)((GM const _Float16*)base)[idx];
  return (type == LIBXSMM_DATATYPE_F32) ? ((GM const float*)base)[idx] : bf16_to_f32(((GM const unsigned short*)base)[idx]);
}

__device__ __forceinline__ bool aligned16(const void* p, long long ld_elems) {
  return ((((unsigned long long)(size_t)p) & 15ull) == 0ull) && ((ld_elems & 3) == 0);
}

// Epilogue shared by all MFMA kernels.  `acc` is in transposed-product layout: lane&31 = i (row of
// C), register r of half h = column j_local(r,h) = (r&3) + 8*(r>>2) + 4*h.
struct TileCtx { int i; int j0; int h; bool ivalid; };
__device__ __forceinline__ int jl_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// CF32: the C (and bias) datatype is known to be f32 at compile time (f32 kernels) -- drops the bf16 paths
// NOBIAS: the caller adds the column bias itself (gemm_wgp16_kernel: from an LDS image behind its first barrier) -- beta * C only
// F16S (IEEE-half GEMMs): the start value is rounded to a half on its way in -- the reference's F16 loop adds f16(start) to the finished sum, whatever C's type, and the
// fused path hands it bias + C as an f32 image [ref: gemm ref :2112-2117 with :296-317].  Identity when the start is one f16 value (C or the bias alone).
template <bool EXACT, bool CF32, bool NOBIAS = false, bool HOIST = false, bool F16S = false>
__device__ __forceinline__ void tile_init(f32x16& acc, const GemmArgs& p, const BatchPtrs& q, const TileCtx& t) {
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  const int c_type = CF32 ? (int)LIBXSMM_DATATYPE_F32 : p.c_type;
  const bool colbias = !NOBIAS && p.colbias;
  if (beta0 && !colbias) {          // the streaming case: nothing to read
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    return;
  }
  float bias = 0.0f;
  if (colbias && (EXACT || t.ivalid)) bias = load_c_f32(q.d, t.i, c_type);
  // HOIST (the workgroup-per-problem kernels): the datatype of C is decided ONCE, outside the loop over the tile's sixteen elements -- a per-element switch puts every load
  // into a basic block of its own and the sixteen loads become sixteen memory round trips one after the other (bf16 72^3 with beta = 1: 0.28 of the roofline against
  // 0.66 with beta = 0).  Not for the wave-per-tile kernels: sixteen addresses live at once cost them a third of their occupancy (gemm_bf16_stream_kernel<1,1>: 76 -> 144
  // registers), also when beta = 0.
  if constexpr (HOIST) {
    float start[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) start[r] = 0.0f;
    if (!beta0) {
      if (c_type == LIBXSMM_DATATYPE_F32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int j = t.j0 + jl_of(r, t.h); if (EXACT || (t.ivalid && j < p.n)) start[r] = ((GM const float*)q.c)[(long long)j * p.ldc + t.i]; }
      } else if (c_type == LIBXSMM_DATATYPE_F16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int j = t.j0 + jl_of(r, t.h); if (EXACT || (t.ivalid && j < p.n)) start[r] = (float)((GM const _Float16*)q.c)[(long long)j * p.ldc + t.i]; }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int j = t.j0 + jl_of(r, t.h); if (EXACT || (t.ivalid && j < p.n)) start[r] = bf16_to_f32(((GM const unsigned short*)q.c)[(long long)j * p.ldc + t.i]); }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float v = colbias ? (beta0 ? bias : bias + start[r]) : start[r]; acc[r] = F16S ? (float)(_Float16)v : v; }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = t.j0 + jl_of(r, t.h);
      float start = 0.0f;
      if (!beta0 && (EXACT || (t.ivalid && j < p.n))) start = load_c_f32(q.c, (long long)j * p.ldc + t.i, c_type);
      const float v = colbias ? (beta0 ? bias : bias + start) : start;
      acc[r] = F16S ? (float)(_Float16)v : v;
    }
  }
}

// The same for a C (and bias) of the operands' 8-bit float type (BF8 / HF8 GEMMs with a result of their own type): beta * C and the fused column bias come in through the
// type [ref: gemm ref :296-317 -- the bias operand has C's type].
template <bool EXACT>
__device__ __forceinline__ void tile_init_c8(f32x16& acc, const GemmArgs& p, const BatchPtrs& q, const TileCtx& t, bool hf8) {
  const bool beta0 = (p.flags & LIBXSMM_GEMM_FLAG_BETA_0) != 0;
  float bias = 0.0f;
  if (p.colbias && (EXACT || t.ivalid)) { const unsigned char x = ((GM const unsigned char*)q.d)[t.i]; bias = hf8 ? hf8_to_f32(x) : bf8_to_f32(x); }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = t.j0 + jl_of(r, t.h);
    float v = 0.0f;
    if (!beta0 && (EXACT || (t.ivalid && j < p.n))) { const unsigned char x = ((GM const unsigned char*)q.c)[(long long)j * p.ldc + t.i]; v = hf8 ? hf8_to_f32(x) : bf8_to_f32(x); }
    acc[r] = p.colbias ? (beta0 ? bias : bias + v) : v;
  }
}

// Hardware RNE conversion of two f32 to a packed bf16 pair (v_cvt_pk_bf16_f32): the reference's rounding for everything but f32 denormals -- bf16_cvt.hpp has the exact
// forms every C store goes through since round 6; this bare one is for values that cannot be denormal (operands being re-laid).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi) { return bf16_pk_hw(lo, hi); }
// the same for IEEE halves (RNE, the conversion the reference's f32 -> f16 helper performs [ref: src/libxsmm_math.c libxsmm_convert_f32_to_f16])
typedef _Float16 hwf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt_pk_f16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hwf16x2));
}
// one 32 x 32 x 16 step on 16-bit operands: bf16 or (F16) IEEE halves -- same operand layout, same rate
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <bool F16> __device__ __forceinline__ f32x16 mfma_16bit(const u32x4& b, const u32x4& a, const f32x16& acc) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, b), __builtin_bit_cast(f16x8_t, a), acc, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

// Tile store.  ACT is the fused activation (0 none, 1 ReLU, 2 ReLU + bitmask, 3 sigmoid), fixed at compile time so the
// streaming variants carry no activation code.  bf16 output of exact tiles goes out as packed dwords: registers
// (2g, 2g+1) hold rows (j, j+1) of column i; one v_cvt_pk_bf16_f32 packs them, one DPP quad swap fetches the
// neighbouring lane's pair and one v_perm_b32 (lane-parity dependent selector) forms (i, i+1) of row j in even
// lanes and of row j+1 in odd lanes: 3 VALU + 1 dword store per two values.
// ReLU bitmask of one tile (bit i % 8 of byte i / 8 + j * mask_ld / 8 [ref: mateltwise ref :150-157, :2142]): a wave ballot per column
template <bool EXACT>
__device__ __forceinline__ void tile_relu_mask(const f32x16& acc, const GemmArgs& p, const BatchPtrs& q, const TileCtx& t) {
  if (!q.mask) return;
  const int lane = threadIdx.x & 63;
  const long long mask_ld = ((p.ldc + 15) / 16) * 16;
  static_for<16>([&](auto rc) {
    constexpr int r = rc.value;
    const int j = t.j0 + jl_of(r, t.h);
    const bool ok = EXACT || (t.ivalid && j < p.n);
    const unsigned long long pos = __ballot(ok && !(acc[r] <= 0.0f));
    const unsigned long long val = __ballot(ok);
    if ((lane & 7) == 0 && ok) {
      GM unsigned char* byte = q.mask + t.i / 8 + (long long)j * (mask_ld / 8);
      const unsigned char vm = (unsigned char)((val >> lane) & 0xffu), nb = (unsigned char)((pos >> lane) & 0xffu);
      *byte = EXACT ? nb : (unsigned char)((*byte & ~vm) | (nb & vm));
    }
  });
}
// the fused activation of a tile whose store is the kernel's own (8-bit float results): bitmask first (it is taken from the sums), then the activation in place
template <bool EXACT>
__device__ __forceinline__ void tile_activate(f32x16& acc, const GemmArgs& p, const BatchPtrs& q, const TileCtx& t) {
  if (p.act == 0) return;                                         // wave-uniform
  if (p.act == 2) tile_relu_mask<EXACT>(acc, p, q, t);
  if (p.act == 3) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = act_fixed<3>(acc[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = act_fixed<1>(acc[r]);
  }
}

template <bool EXACT, bool CF32, int ACT, bool NT>
__device__ __forceinline__ void tile_store_impl(const f32x16& acc, const GemmArgs& p, const BatchPtrs& q, const TileCtx& t) {
  const int lane = threadIdx.x & 63;
  const bool out_f32 = CF32 || (p.c_type == LIBXSMM_DATATYPE_F32);
  // bf16 fast path needs an even ldc and a 4-byte aligned C
  // (round 4: also for ragged tiles with an even m -- every row pair is whole -- with the store masked by row and column)
  const bool pack2 = (EXACT || (p.m & 1) == 0) && !out_f32 && ((p.ldc & 1) == 0) && ((((unsigned long long)(size_t)q.c) & 3ull) == 0ull);
  if (ACT == 2) tile_relu_mask<EXACT>(acc, p, q, t);
  const bool c_f16 = p.c_type == LIBXSMM_DATATYPE_F16;            // wave-uniform: the 16-bit output is bf16 or IEEE half
  if (!CF32 && pack2) {
    const bool odd = (lane & 1) != 0;
    const unsigned int sel = odd ? 0x03020706u : 0x05040100u;
    GM unsigned short* base = (GM unsigned short*)q.c + (long long)(t.j0 + 4 * t.h + (odd ? 1 : 0)) * p.ldc + (t.i & ~1);
    float y[16]; unsigned int wp[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = act_fixed<ACT>(acc[r]);
    if (c_f16) {
#pragma unroll
      for (int g = 0; g < 8; ++g) wp[g] = cvt_pk_f16(y[2 * g], y[2 * g + 1]);
    } else bf16_pk_exact_n<8>(y, wp);                     // (the reference's conversion exactly: denormal sums are flushed -- bf16_cvt.hpp)
    static_for<8>([&](auto gc) {
      constexpr int g = gc.value, r0 = 2 * g, jr = (r0 & 3) + 8 * (r0 >> 2);
      const unsigned int w = wp[g];
      const unsigned int n = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
      if (EXACT || (t.ivalid && t.j0 + 4 * t.h + (odd ? 1 : 0) + jr < p.n))
        st_stream<NT>((GM unsigned int*)(base + (long long)jr * p.ldc), (unsigned int)__builtin_amdgcn_perm(n, w, sel));
    });
    return;
  }
  static_for<16>([&](auto rc) {
    constexpr int r = rc.value;
    const int j = t.j0 + jl_of(r, t.h);
    const bool ok = EXACT || (t.ivalid && j < p.n);
    const float y = act_fixed<ACT>(acc[r]);
    if (out_f32) { if (ok) st_stream<NT>((GM float*)q.c + (long long)j * p.ldc + t.i, y); }
    else if (ok) st_stream<NT>((GM unsigned short*)q.c + (long long)j * p.ldc + t.i, c_f16 ? __builtin_bit_cast(unsigned short, (_Float16)y) : f32_to_bf16_rne(y));
  });
}
// wave-uniform dispatch on the activation
template <bool EXACT, bool CF32, bool NT = true>
__device__ __forceinline__ void tile_store(const f32x16& acc, const GemmArgs& p, const BatchPtrs& q, const TileCtx& t) {
  if (p.act == 0) tile_store_impl<EXACT, CF32, 0, NT>(acc, p, q, t);
  else if (p.act == 1) tile_store_impl<EXACT, CF32, 1, NT>(acc, p, q, t);
  else if (p.act == 2) tile_store_impl<EXACT, CF32, 2, NT>(acc, p, q, t);
  else tile_store_impl<EXACT, CF32, 3, NT>(acc, p, q, t);
}

// The same store through a descriptor that carries the exact extent of the C block (prepared with BND, see gemm_mfma_bf16_kernel; off by default): one lane offset
// per tile, every column offset scalar; a column beyond n lies beyond the extent and its store is dropped by the address unit, rows beyond m are masked lanes.
// Plain results only (no bitmask, no VNNI C): f32, or 16-bit as packed row pairs when m and ldc are even and C starts on a dword, else element-wise.
template <int ACT>
__device__ __forceinline__ void tile_store_buf_impl(const f32x16& acc, const GemmArgs& p, const __amdgpu_buffer_rsrc_t& rc, bool c_dword, const TileCtx& t) {
  if (!t.ivalid) return;
  const unsigned int ldc = (unsigned int)p.ldc, lane = threadIdx.x & 63u;
  if (p.c_type == LIBXSMM_DATATYPE_F32) {
    const unsigned int voff = ((4u * (unsigned int)t.h) * ldc + (unsigned int)t.i) * 4u;
    static_for<16>([&](auto rc_) { constexpr int r = rc_.value;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(act_fixed<ACT>(acc[r])), rc, (int)voff, (int)(((unsigned int)t.j0 + (unsigned int)((r & 3) + 8 * (r >> 2))) * ldc * 4u), 0); });
    return;
  }
  const bool c_f16 = p.c_type == LIBXSMM_DATATYPE_F16;
  if (c_dword && !(p.m & 1) && !(ldc & 1u)) {
    const bool odd = (lane & 1u) != 0;
    const unsigned int sel = odd ? 0x03020706u : 0x05040100u;
    const unsigned int voff = ((4u * (unsigned int)t.h + (odd ? 1u : 0u)) * ldc + ((unsigned int)t.i & ~1u)) * 2u;
    float y[16]; unsigned int wp[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = act_fixed<ACT>(acc[r]);
    if (c_f16) {
#pragma unroll
      for (int g = 0; g < 8; ++g) wp[g] = cvt_pk_f16(y[2 * g], y[2 * g + 1]);
    } else bf16_pk_exact_n<8>(y, wp);
    static_for<8>([&](auto gc) { constexpr int g = gc.value, r0 = 2 * g, jr = (r0 & 3) + 8 * (r0 >> 2);
      const unsigned int w = wp[g];
      const unsigned int n = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
      __builtin_amdgcn_raw_buffer_store_b32((unsigned int)__builtin_amdgcn_perm(n, w, sel), rc, (int)voff, (int)(((unsigned int)t.j0 + (unsigned int)jr) * ldc * 2u), 0); });
    return;
  }
  const unsigned int voff = ((4u * (unsigned int)t.h) * ldc + (unsigned int)t.i) * 2u;
  static_for<16>([&](auto rc_) { constexpr int r = rc_.value;
    const float y = act_fixed<ACT>(acc[r]);
    __builtin_amdgcn_raw_buffer_store_b16(c_f16 ? __builtin_bit_cast(unsigned short, (_Float16)y) : f32_to_bf16_rne(y), rc, (int)voff, (int)(((unsigned int)t.j0 + (unsigned int)((r & 3) + 8 * (r >> 2))) * ldc * 2u), 0); });
}
__device__ __forceinline__ void tile_store_buf(const f32x16& acc, const GemmArgs& p, const __amdgpu_buffer_rsrc_t& rc, bool c_dword, const TileCtx& t) {
  if (p.act == 0) tile_store_buf_impl<0>(acc, p, rc, c_dword, t);
  else if (p.act == 1) tile_store_buf_impl<1>(acc, p, rc, c_dword, t);
  else tile_store_buf_impl<3>(acc, p, rc, c_dword, t);
}

// wave -> (batch element, tile) decomposition shared by the MFMA kernels
struct WaveJob { unsigned int bidx; int i0, j0; bool active; };
__device__ __forceinline__ WaveJob wave_job(const GemmArgs& p, int tile_m, int tile_n) {
  // The wave index is uniform but not provably so to the compiler: readfirstlane moves the whole tile/batch
  // address computation to the scalar unit.  Index math is 32-bit and division-free in the common case of
  // one tile per problem (launch_gemm guarantees tiles * nbatch < 2^31).
  const unsigned int wid = logical_block(p) * (blockDim.x >> 6) + (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int per_gemm = (unsigned int)(p.tiles_m * p.tiles_n);
  WaveJob w;
  w.active = wid < per_gemm * p.nbatch;
  if (per_gemm == 1) { w.bidx = wid; w.i0 = 0; w.j0 = 0; }
  else {
    w.bidx = wid / per_gemm;
    const unsigned int t = wid - w.bidx * per_gemm;
    const unsigned int tn = t / (unsigned int)p.tiles_m;
    w.i0 = (int)(t - tn * (unsigned int)p.tiles_m) * tile_m; w.j0 = (int)tn * tile_n;
  }
  return w;
}

// 8-bit WEIGHTS as the bf16 values the reference multiplies with (gemm_w8_bf16_kernel, gemm_wgp16_kernel<.., AK>): KIND 0 / 1: BF8 / HF8 in VNNI-2 pairs, 2 / 3: flat, 4: int8 x row scale
typedef float f32x2w __attribute__((ext_vector_type(2)));
template <int KIND> __device__ __forceinline__ unsigned int w8_pair_to_bf16(unsigned int two_bytes, float scf) {      // byte 0 = even k, byte 1 = odd k  ->  packed bf16 pair
  if constexpr (KIND == 4) {
    const float f0 = (float)(int)(signed char)(two_bytes & 0xffu) * scf, f1 = (float)(int)(signed char)((two_bytes >> 8) & 0xffu) * scf;
    return cvt_pk_bf16(f0, f1);
  } else {
    const f32x2w f = (KIND & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)two_bytes, false) : __builtin_amdgcn_cvt_pk_f32_bf8((int)two_bytes, false);
    return (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);        // exact: both formats have at most 3 significand bits
  }
}

}  // namespace xamd
