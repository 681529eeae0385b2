// gemm_8bit.hpp -- pieces shared by the 8-bit matrix-core kernels of gemm_kernels.hip (wave per tile) and gemm_wgp16_kernels.hip (workgroup per problem), round 5:
// the products of one 32-deep chunk with the signedness corrections of the integer forms (the matrix core is signed-only: an unsigned operand is fed as u ^ 0x80 and
// 128 * sum(other operand) is recovered from v_dot4 sums), the reference's two-step f32 -> E5M2 / E4M3 rounding on the hardware conversion, correctly rounded mul / add.
#pragma once
#include "gemm_tile.hpp"
#include "lowp.hpp"

#pragma clang fp contract(off)

namespace xamd {

template <typename T> __device__ __forceinline__ T mul_rn(T a, T b) { return a * b; }
template <typename T> __device__ __forceinline__ T add_rn(T a, T b) { return a + b; }

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned int f32x2_to_fp8_ref(float x0, float x1, bool hf8) {      // byte 0: x0, byte 1: x1
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
  const float f0 = (float)h0, f1 = (float)h1;
  unsigned int r = (unsigned int)(hf8 ? __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false) : __builtin_amdgcn_cvt_pk_bf8_f32(f0, f1, 0, false)) & 0xffffu;
  if (__builtin_expect((x0 != x0) || (x1 != x1), 0)) {
    const unsigned short b0 = __builtin_bit_cast(unsigned short, h0), b1 = __builtin_bit_cast(unsigned short, h1);
    r = hf8 ? ((unsigned int)lowp::f16_to_hf8_rne(b0) | ((unsigned int)lowp::f16_to_hf8_rne(b1) << 8)) : ((unsigned int)lowp::f16_to_bf8_rne(b0) | ((unsigned int)lowp::f16_to_bf8_rne(b1) << 8));
  }
  return r;
}
__device__ __forceinline__ unsigned char f32_to_fp8_ref(float x, bool hf8) { return (unsigned char)(f32x2_to_fp8_ref(x, x, hf8) & 0xffu); }

template <int MT, int NT, int KIND, bool UA, bool UB>
__device__ __forceinline__ void m8_products(const unsigned int (&aw)[MT][4], const unsigned int (&bw)[NT][4], i32x16 (&iacc)[KIND == 0 ? MT : 1][KIND == 0 ? NT : 1],
                                            f32x16 (&facc)[KIND == 0 ? 1 : MT][KIND == 0 ? 1 : NT], int (&sum_a)[MT], int (&sum_b)[NT]) {
  constexpr bool INT = KIND == 0, HF8 = KIND == 2;
  if constexpr (INT) {
    i32x4 af[MT], bf[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = i32x4{(int)aw[mt][0], (int)aw[mt][1], (int)aw[mt][2], (int)aw[mt][3]};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[nt] = i32x4{(int)bw[nt][0], (int)bw[nt][1], (int)bw[nt][2], (int)bw[nt][3]};
    static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
      iacc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[nt], af[mt], iacc[mt][nt], 0, 0, 0); });
    if constexpr (UA) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum_b[nt] = __builtin_amdgcn_sdot4((int)bw[nt][e], 0x01010101, sum_b[nt], false);
    }
    if constexpr (UB) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum_a[mt] = __builtin_amdgcn_sdot4((int)aw[mt][e], 0x01010101, sum_a[mt], false);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      static_for<MT * NT>([&](auto idx) { constexpr int mt = idx.value / NT, nt = idx.value % NT;
        const long a8 = (long)(((unsigned long long)aw[mt][2 * s + 1] << 32) | aw[mt][2 * s]), b8 = (long)(((unsigned long long)bw[nt][2 * s + 1] << 32) | bw[nt][2 * s]);
        if (HF8) facc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b8, a8, facc[mt][nt], 0, 0, 0);
        else facc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(b8, a8, facc[mt][nt], 0, 0, 0); });
  }
}
// BL (strided forms with dword-aligned blocks and columns: launch_gemm): the wave's B panel of a chunk -- 32 bytes of each of its columns, a whole column apart in
// memory -- by LDS-DMA a dword per lane (eight lanes = one column, eight columns per instruction, no registers) and back as ONE ds_read_b128 (integers) / two
// ds_read_b64 (8-bit floats: k-quads 4 s + 2 h + {0, 1}) per column tile, the 64 lanes reading the 2 KiB image end to end; A buffer-addressed (32-bit offsets).

}  // namespace xamd
