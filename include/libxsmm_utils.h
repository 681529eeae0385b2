/*
 * libxsmm_utils.h -- the helper surface the reference's sample drivers compile and link against
 * ("plus helpers the sample drivers link against"; Appendix C: the reference's own drivers are the
 * integration test-suite).  With this header and libxsmm_amd.so the reference's UNMODIFIED driver sources
 * (samples/xgemm/gemm_kernel.c, samples/xgemm_sparse/spmm_kernel.c, samples/xgemm_norm_packed/\*.c,
 * samples/xgemm_sparse_Ainregs/pyfr_driver_asp_reg.c, samples/eltwise/\*.c) build with gcc and run on the GPU:
 * oracle/Makefile target `drivers`, tests/test_reference_drivers_gpu.py.
 *
 * What is here: the C library headers the reference's umbrella header pulls in, small generic macros
 * [ref: include/libxsmm_macros.h, include/libxsmm_math.h:17-60], 16/8-bit float helpers
 * [ref: include/libxsmm_math.h:186-214, include/utils/libxsmm_lpflt_quant.h:44-59] and the external RNG state
 * [ref: include/libxsmm_math.h:216-231].  Everything is written against the documented behaviour, not copied.
 */
#ifndef LIBXSMM_UTILS_H
#define LIBXSMM_UTILS_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <inttypes.h>
#include <assert.h>
#include <float.h>
#include <math.h>
#include "libxsmm.h"

/* ---- generic macros --------------------------------------------------------------------- */
#define LIBXSMM_ABS(A) (0 <= (A) ? (A) : -(A))
#define LIBXSMM_NEQ(A, B) ((A) != (B))
#define LIBXSMM_FABS(A) fabs(A)
#define LIBXSMM_FABSF(A) fabsf(A)
#define LIBXSMM_POWF(A, B) powf(A, B)
#define LIBXSMM_LOGF(A) logf(A)
#define LIBXSMM_FLOORF(A) floorf(A)
#define LIBXSMM_ROUNDF(A) roundf(A)
#define LIBXSMM_FREXPF(A, B) frexpf(A, B)
#define LIBXSMM_EXPF(A) expf(A)
#define LIBXSMM_TANHF(A) tanhf(A)
#define LIBXSMM_SQRTF(A) sqrtf(A)
#define LIBXSMM_ERFF(A) erff(A)
#define LIBXSMM_CAST_BLASINT(VALUE) ((libxsmm_blasint)(VALUE))
#define LIBXSMM_CAST_UINT(VALUE) ((unsigned int)(VALUE))
#define LIBXSMM_CAST_USHORT(VALUE) ((unsigned short)(VALUE))
#define LIBXSMM_EOR(ENUM_TYPE, ENUM, FLAG) ((ENUM_TYPE)(((int)(ENUM)) | ((int)(FLAG))))
#define LIBXSMM_EXPECT(EXPR) do { if (!(EXPR)) { /* evaluated, not enforced */ } } while (0)
#define LIBXSMM_SNPRINTF(S, N, ...) snprintf(S, N, __VA_ARGS__)
#define LIBXSMM_PUTENV(A) putenv(A)
#define LIBXSMM_PRAGMA_SIMD
#define LIBXSMM_OMP_VAR(A) (void)(A)
#define LIBXSMM_ELIDE_RESULT(TYPE, EXPR) do { const TYPE libxsmm_elide_result_ = (EXPR); (void)libxsmm_elide_result_; } while (0)
#define LIBXSMM_ROUND(A) round(A)
#define LIBXSMM_DELTA(T0, T1) ((T0) < (T1) ? ((T1) - (T0)) : ((T0) - (T1)))
#define LIBXSMM_CONST_VOID_PTR(A) ((const void*)(A))
#if defined(_OPENMP)
# define LIBXSMM_OMP_MASKED _Pragma("omp master")
#else
# define LIBXSMM_OMP_MASKED
#endif

/* multi-dimensional views over flat buffers (C99 variably modified types): NDIMS is a literal 1..6,
 * DECL lists the initial pointer followed by the NDIMS-1 inner bounds, ACCESS the NDIMS indices followed by the same bounds */
#if !defined(LIBXSMM_ASSERT)
# include <assert.h>
# define LIBXSMM_ASSERT(EXPR) assert(EXPR)
# define LIBXSMM_ASSERT_MSG(EXPR, MSG) assert((EXPR) && *(MSG))
#endif
/* linear (row-major) index of I0..In-1 under the bounds S1..Sn-1, and untyped / typed element addresses computed from it
 * [ref: include/libxsmm_macros.h:751-819]; arrays declared by LIBXSMM_VLA_DECL are real C99 VLAs here, so their name carries no postfix */
#define LIBXSMM_VLA
#define LIBXSMM_VLA_POSTFIX
#define LIBXSMM_INDEX1(NDIMS, ...) LIBXSMM_CONCATENATE(LIBXSMM_INDEX1_, NDIMS)(__VA_ARGS__)
#define LIBXSMM_INDEX1_1(I0) ((size_t)(I0))
#define LIBXSMM_INDEX1_2(I0, I1, S1) (LIBXSMM_INDEX1_1(I0) * (size_t)(S1) + (size_t)(I1))
#define LIBXSMM_INDEX1_3(I0, I1, I2, S1, S2) (LIBXSMM_INDEX1_2(I0, I1, S1) * (size_t)(S2) + (size_t)(I2))
#define LIBXSMM_INDEX1_4(I0, I1, I2, I3, S1, S2, S3) (LIBXSMM_INDEX1_3(I0, I1, I2, S1, S2) * (size_t)(S3) + (size_t)(I3))
#define LIBXSMM_INDEX1_5(I0, I1, I2, I3, I4, S1, S2, S3, S4) (LIBXSMM_INDEX1_4(I0, I1, I2, I3, S1, S2, S3) * (size_t)(S4) + (size_t)(I4))
#define LIBXSMM_INDEX1_6(I0, I1, I2, I3, I4, I5, S1, S2, S3, S4, S5) (LIBXSMM_INDEX1_5(I0, I1, I2, I3, I4, S1, S2, S3, S4) * (size_t)(S5) + (size_t)(I5))
#define LIBXSMM_ACCESS_RO(NDIMS, TYPESIZE, ARRAY, ...) ((const void*)((const char*)(ARRAY) + (size_t)(TYPESIZE) * LIBXSMM_INDEX1(NDIMS, __VA_ARGS__)))
#define LIBXSMM_ACCESS_RW(NDIMS, TYPESIZE, ARRAY, ...) ((void*)((char*)(ARRAY) + (size_t)(TYPESIZE) * LIBXSMM_INDEX1(NDIMS, __VA_ARGS__)))
#define LIBXSMM_ACCESS(NDIMS, TYPE, ARRAY, ...) ((TYPE*)(ARRAY) + LIBXSMM_INDEX1(NDIMS, __VA_ARGS__))
#define LIBXSMM_VLA_DECL(NDIMS, ELEMENT_TYPE, ARRAY_VAR, ...) LIBXSMM_CONCATENATE(LIBXSMM_VLA_DECL_, NDIMS)(ELEMENT_TYPE, ARRAY_VAR, __VA_ARGS__)
#define LIBXSMM_VLA_DECL_1(T, V, INIT) T* V = (T*)(INIT)
#define LIBXSMM_VLA_DECL_2(T, V, INIT, S1) T (*V)[S1] = (T (*)[S1])(INIT)
#define LIBXSMM_VLA_DECL_3(T, V, INIT, S1, S2) T (*V)[S1][S2] = (T (*)[S1][S2])(INIT)
#define LIBXSMM_VLA_DECL_4(T, V, INIT, S1, S2, S3) T (*V)[S1][S2][S3] = (T (*)[S1][S2][S3])(INIT)
#define LIBXSMM_VLA_DECL_5(T, V, INIT, S1, S2, S3, S4) T (*V)[S1][S2][S3][S4] = (T (*)[S1][S2][S3][S4])(INIT)
#define LIBXSMM_VLA_DECL_6(T, V, INIT, S1, S2, S3, S4, S5) T (*V)[S1][S2][S3][S4][S5] = (T (*)[S1][S2][S3][S4][S5])(INIT)
#define LIBXSMM_VLA_ACCESS(NDIMS, ARRAY, ...) LIBXSMM_CONCATENATE(LIBXSMM_VLA_ACCESS_, NDIMS)(ARRAY, __VA_ARGS__)
#define LIBXSMM_VLA_ACCESS_1(A, I0) ((A)[I0])
#define LIBXSMM_VLA_ACCESS_2(A, I0, I1, S1) ((A)[I0][I1])
#define LIBXSMM_VLA_ACCESS_3(A, I0, I1, I2, S1, S2) ((A)[I0][I1][I2])
#define LIBXSMM_VLA_ACCESS_4(A, I0, I1, I2, I3, S1, S2, S3) ((A)[I0][I1][I2][I3])
#define LIBXSMM_VLA_ACCESS_5(A, I0, I1, I2, I3, I4, S1, S2, S3, S4) ((A)[I0][I1][I2][I3][I4])
#define LIBXSMM_VLA_ACCESS_6(A, I0, I1, I2, I3, I4, I5, S1, S2, S3, S4, S5) ((A)[I0][I1][I2][I3][I4][I5])

/* Fill an NROWS x NCOLS column-major matrix (leading dimension LD) with reproducible values: SEED != 0 gives
 * SCALE*(SEED+1)*(1 + col*NROWS + row) and SEED in the padding rows, SEED == 0 a shuffled ramp inside [-SCALE, +SCALE]. */
LIBXSMM_API double libxsmm_hip_matinit_value(double seed, double scale, libxsmm_blasint row, libxsmm_blasint col,
  libxsmm_blasint nrows, libxsmm_blasint ncols, libxsmm_blasint ld);
#define LIBXSMM_MATINIT(TYPE, SEED, DST, NROWS, NCOLS, LD, SCALE) do { \
  libxsmm_blasint libxsmm_mi_c_, libxsmm_mi_r_; const libxsmm_blasint libxsmm_mi_ld_ = (libxsmm_blasint)(LD); \
  for (libxsmm_mi_c_ = 0; libxsmm_mi_c_ < (libxsmm_blasint)(NCOLS); ++libxsmm_mi_c_) \
    for (libxsmm_mi_r_ = 0; libxsmm_mi_r_ < libxsmm_mi_ld_; ++libxsmm_mi_r_) \
      ((TYPE*)(DST))[(size_t)libxsmm_mi_c_ * (size_t)libxsmm_mi_ld_ + (size_t)libxsmm_mi_r_] = (TYPE)libxsmm_hip_matinit_value( \
        (double)(SEED), (double)(SCALE), libxsmm_mi_r_, libxsmm_mi_c_, (libxsmm_blasint)(NROWS), (libxsmm_blasint)(NCOLS), libxsmm_mi_ld_); \
} while (0)
#define LIBXSMM_MATINIT_SEQ(TYPE, SEED, DST, NROWS, NCOLS, LD, SCALE) LIBXSMM_MATINIT(TYPE, SEED, DST, NROWS, NCOLS, LD, SCALE)
#define LIBXSMM_MATINIT_OMP(TYPE, SEED, DST, NROWS, NCOLS, LD, SCALE) LIBXSMM_MATINIT(TYPE, SEED, DST, NROWS, NCOLS, LD, SCALE)

/* ---- target ids the drivers compare against [ref: include/libxsmm_cpuid.h:23-59]; this back end reports
 * LIBXSMM_X86_GENERIC (see libxsmm.h), so none of their ISA-specific branches is taken ---------------------- */
#define LIBXSMM_X86_SSE3              1003
#define LIBXSMM_X86_SSE42             1004
#define LIBXSMM_X86_AVX               1005
#define LIBXSMM_X86_AVX2              1006
#define LIBXSMM_X86_AVX2_ADL          1007
#define LIBXSMM_X86_AVX2_SRF          1008
#define LIBXSMM_X86_AVX512_VL128_SKX  1041
#define LIBXSMM_X86_AVX512_VL256_SKX  1051
#define LIBXSMM_X86_AVX512_VL256_CLX  1052
#define LIBXSMM_X86_AVX512_VL256_CPX  1053
#define LIBXSMM_X86_AVX512_SKX        1101
#define LIBXSMM_X86_AVX512_CLX        1102
#define LIBXSMM_X86_AVX512_CPX        1103
#define LIBXSMM_X86_AVX512_GNR        1105
#define LIBXSMM_X86_AVX512_DMR        1106
#define LIBXSMM_X86_AVX512_ACE1       1107
#define LIBXSMM_X86_ALLFEAT           1999
#define LIBXSMM_AARCH64_V81           2001
#define LIBXSMM_AARCH64_ALLFEAT       2999
#define LIBXSMM_RV64_ALLFEAT          3999

/* ---- 16-bit and 8-bit floats ------------------------------------------------------------- */
typedef union libxsmm_float16_ushort { libxsmm_float16 f; unsigned short u; } libxsmm_float16_ushort;
typedef union libxsmm_bfloat8_f16 { libxsmm_bfloat8 i[2]; libxsmm_float16 hf; } libxsmm_bfloat8_f16;

/** IEEE half <-> f32: RNE, f32 denormals flushed first, overflow -> infinity, NaNs quieted. */
LIBXSMM_API libxsmm_float16 libxsmm_convert_f32_to_f16(float in);
LIBXSMM_API float libxsmm_convert_f16_to_f32(libxsmm_float16 in);
/** BF8 (E5M2) = RNE of the half's upper byte; HF8 (E4M3, bias 7, no infinities: overflow and specials -> NaN 0x7f). */
LIBXSMM_API libxsmm_bfloat8 libxsmm_convert_f32_to_bf8_rne(float in);
LIBXSMM_API libxsmm_bfloat8 libxsmm_convert_f32_to_bf8_stochastic(float in, unsigned int seed);
LIBXSMM_API libxsmm_hfloat8 libxsmm_convert_f16_to_hf8_rne(libxsmm_float16 in);
LIBXSMM_API libxsmm_hfloat8 libxsmm_convert_f32_to_hf8_rne(float in);
LIBXSMM_API float libxsmm_convert_bf8_to_f32(libxsmm_bfloat8 in);
LIBXSMM_API float libxsmm_convert_hf8_to_f32(libxsmm_hfloat8 in);
/** array forms */
LIBXSMM_API void libxsmm_rne_convert_fp32_f16(const float* in, libxsmm_float16* out, size_t length);
LIBXSMM_API void libxsmm_convert_f16_f32(const libxsmm_float16* in, float* out, size_t length);
LIBXSMM_API void libxsmm_rne_convert_fp32_bf8(const float* in, libxsmm_bfloat8* out, size_t length);
LIBXSMM_API void libxsmm_convert_bf8_f32(const libxsmm_bfloat8* in, float* out, size_t length);
LIBXSMM_API void libxsmm_rne_convert_fp32_hf8(const float* in, libxsmm_hfloat8* out, size_t length);
LIBXSMM_API void libxsmm_convert_hf8_f32(const libxsmm_hfloat8* in, float* out, size_t length);
LIBXSMM_API void libxsmm_stochastic_convert_fp32_bf8(const float* in, libxsmm_bfloat8* out, unsigned int length,
  void* rng_state, unsigned int start_seed_idx);

/* ---- external RNG state: 16 xoshiro128+ lanes, 4 x 16 words [ref: include/libxsmm_math.h:216-231] ------------ */
LIBXSMM_API unsigned int* libxsmm_rng_create_extstate(unsigned int seed);
LIBXSMM_API unsigned int libxsmm_rng_get_extstate_size(void);
LIBXSMM_API void libxsmm_rng_destroy_extstate(unsigned int* stateptr);

/* ---- small math helpers the quantisation samples use [ref: include/utils/libxsmm_math.h:40-54, include/libxsmm_math.h:257-267] ------ */
/** 2^x for 8-bit exponents, exact in single precision (a power of two, or 0 / infinity beyond the f32 range). */
LIBXSMM_API float libxsmm_sexp2_u8(unsigned char x);
LIBXSMM_API float libxsmm_sexp2_i8(signed char x);
LIBXSMM_API float libxsmm_sexp2_i8i(int x);
/** round to the nearest integer, ties to even (the current rounding mode of the host is the default one) */
LIBXSMM_API double libxsmm_nearbyint(double x);
LIBXSMM_API float libxsmm_nearbyintf(float x);
LIBXSMM_API double libxsmm_dsqrt(double x);
LIBXSMM_API float libxsmm_ssqrt(float x);

/* ---- strings [ref: include/libxsmm_memory.h:102-103] ------------------------------------- */
LIBXSMM_API const char* libxsmm_stristrn(const char a[], const char b[], size_t maxlen);
LIBXSMM_API const char* libxsmm_stristr(const char a[], const char b[]);

#endif /* LIBXSMM_UTILS_H */
