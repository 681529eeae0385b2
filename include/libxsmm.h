/*
 * libxsmm.h -- public C API of the MI355X-native TPP backend (libxsmm_amd).
 *
 * This single header is the drop-in boundary: it declares the same symbols, enum
 * values and struct layouts a caller of libxsmm/libxsmm 2.0 compiles against for
 * the dispatch/param hot path, so that existing C host code recompiles unchanged
 * and links against libxsmm_amd.so instead of libxsmm.a.  What is different is what
 * a dispatched handle *does*: it launches a hand-written CDNA4 (gfx950) HIP kernel
 * instead of jumping into JIT-emitted x86 code.
 *
 * It was written from the reference's documented interface, not copied from it.
 * For every block the reference location it replaces is cited as
 *   [ref: <path under /root/reference>:<lines>].
 *
 * Pointer contract (the one semantic addition of a GPU backend): every data pointer
 * stored in a libxsmm_*_param must be device-accessible (hipMalloc, hipMallocManaged,
 * hipHostMalloc or memory obtained from libxsmm_aligned_malloc of THIS library, which
 * returns pinned device-visible host memory).  See libxsmm_hip.h for streams, batched
 * launches and the multi-GPU sharder.
 */
#ifndef LIBXSMM_H
#define LIBXSMM_H

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
# define LIBXSMM_EXTERN_C extern "C"
# define LIBXSMM_EXTERN extern "C"
#else
# define LIBXSMM_EXTERN_C
# define LIBXSMM_EXTERN extern
#endif
#if !defined(LIBXSMM_API)
# define LIBXSMM_API LIBXSMM_EXTERN_C __attribute__((visibility("default")))
#endif
#define LIBXSMM_APIEXT LIBXSMM_API
#define LIBXSMM_INLINE static inline
#define LIBXSMM_ARGDEF(ARG, DEFAULT) ARG

/* ---- version / configuration constants ------------------------------------------ */
#define LIBXSMM_VERSION_MAJOR 2
#define LIBXSMM_VERSION_MINOR 0
#define LIBXSMM_BACKEND_HIP_GFX950 1
#define LIBXSMM_ALIGNMENT 64
#define LIBXSMM_ALPHA 1
#define LIBXSMM_BETA 1
#define LIBXSMM_FLAGS 0
#define LIBXSMM_PREFETCH_NONE 0
/* [ref: include/libxsmm_typedefs.h:143-145] opaque storage handed to *_descriptor_init */
#define LIBXSMM_DESCRIPTOR_MAXSIZE 96

/* ---- small utility macros callers of the reference rely on ----------------------- */
#define LIBXSMM_MIN(A, B) ((A) < (B) ? (A) : (B))
#define LIBXSMM_MAX(A, B) ((A) < (B) ? (B) : (A))
#define LIBXSMM_UPDIV(N, D) (((N) + (D) - 1) / (D))
#define LIBXSMM_LO2(N, NPOT) ((N) & ~((NPOT) - 1))          /* round down / up to a multiple of a power of two [ref: include/libxsmm_macros.h:630-631] */
#define LIBXSMM_UP2(N, NPOT) LIBXSMM_LO2((N) + ((NPOT) - 1), NPOT)
#define LIBXSMM_UP(N, D) (LIBXSMM_UPDIV(N, D) * (D))
#define LIBXSMM_UNUSED(X) (void)(X)
#define LIBXSMM_CONCATENATE_(A, B) A##B
#define LIBXSMM_CONCATENATE(A, B) LIBXSMM_CONCATENATE_(A, B)

/* ---- scalar typedefs  [ref: include/libxsmm_typedefs.h:155-196] ------------------- */
typedef int                libxsmm_blasint;      /* LP64 build of the reference */
typedef unsigned int       libxsmm_bitfield;
typedef unsigned long long libxsmm_timer_tickint;
typedef unsigned short     libxsmm_bfloat16;
typedef unsigned short     libxsmm_float16;
typedef unsigned char      libxsmm_bfloat8;
typedef unsigned char      libxsmm_hfloat8;
typedef union libxsmm_float_uint   { float f; unsigned int u; } libxsmm_float_uint;
typedef union libxsmm_bfloat16_f32 { libxsmm_bfloat16 i[2]; float f; } libxsmm_bfloat16_f32;

/* Same size as the reference's blob; 8-byte aligned because this library's descriptors hold 64-bit strides. */
#if defined(__GNUC__)
typedef struct libxsmm_descriptor_blob { char data[LIBXSMM_DESCRIPTOR_MAXSIZE]; } __attribute__((aligned(8))) libxsmm_descriptor_blob;
#else
typedef struct libxsmm_descriptor_blob { union { char data[LIBXSMM_DESCRIPTOR_MAXSIZE]; long long align_[LIBXSMM_DESCRIPTOR_MAXSIZE / 8]; }; } libxsmm_descriptor_blob;
#endif
typedef struct libxsmm_gemm_descriptor  libxsmm_gemm_descriptor;   /* opaque */
typedef struct libxsmm_meltw_descriptor libxsmm_meltw_descriptor;  /* opaque */

/* ---- element types  [ref: include/libxsmm_typedefs.h:218-246] ---------------------
 * Kept as an X-table so the library can also derive name/size tables from it. */
#define LIBXSMM_DATATYPE_TABLE(X) \
  X(F64, 8) X(F32, 4) X(BF16, 2) X(F16, 2) X(BF8, 1) X(HF8, 1) X(I64, 8) X(U64, 8) \
  X(I32, 4) X(U32, 4) X(I16, 2) X(U16, 2) X(I8, 1) X(U8, 1) X(MXBF8, 1) X(MXHF8, 1) \
  X(MXBF6, 1) X(MXHF6, 1) X(I4X2, 1) X(U4X2, 1) X(MXFP4X2, 1) X(NVFP4X2, 1) X(I2X4, 1) \
  X(I1X8, 1) X(BF32, 4) X(IMPLICIT, 0) X(UNSUPPORTED, 0)
typedef enum libxsmm_datatype {
#define LIBXSMM_X_(NAME, SIZE) LIBXSMM_DATATYPE_##NAME,
  LIBXSMM_DATATYPE_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  LIBXSMM_DATATYPE_COUNT_
} libxsmm_datatype;
LIBXSMM_API unsigned char libxsmm_typesize(libxsmm_datatype datatype);
#define LIBXSMM_TYPESIZE(ENUM) ((int)libxsmm_typesize((libxsmm_datatype)(ENUM)))
/* LIBXSMM_DATATYPE(float) etc. [ref: include/libxsmm_typedefs.h:137] */
#define LIBXSMM_DATATYPE_double LIBXSMM_DATATYPE_F64
#define LIBXSMM_DATATYPE_float  LIBXSMM_DATATYPE_F32
#define LIBXSMM_DATATYPE_int    LIBXSMM_DATATYPE_I32
#define LIBXSMM_DATATYPE_short  LIBXSMM_DATATYPE_I16
#define LIBXSMM_DATATYPE_char   LIBXSMM_DATATYPE_I8
#define LIBXSMM_DATATYPE(TYPE) LIBXSMM_CONCATENATE(LIBXSMM_DATATYPE_, TYPE)

/* ---- element-wise (TPP) operations  [ref: include/libxsmm_typedefs.h:248-444] ----- */
typedef enum libxsmm_meltw_operation {
  LIBXSMM_MELTW_OPERATION_NONE = 0, LIBXSMM_MELTW_OPERATION_UNARY = 1,
  LIBXSMM_MELTW_OPERATION_BINARY = 2, LIBXSMM_MELTW_OPERATION_TERNARY = 3
} libxsmm_meltw_operation;

#define LIBXSMM_MELTW_UNARY_FLAG_TABLE(X) \
  X(NONE, 0) X(BITMASK_2BYTEMULT, 1) X(BCAST_ROW, 2) X(BCAST_COL, 4) X(BCAST_SCALAR, 8) \
  X(REDUCE_COLS, 16) X(REDUCE_ROWS, 32) X(REDUCE_INIT_ACC, 64) X(IDX_SIZE_4BYTES, 128) \
  X(IDX_SIZE_8BYTES, 256) X(REDUCE_INF_ACC, 512) X(REDUCE_NO_PREFETCH, 1024) \
  X(REDUCE_RECORD_ARGOP, 2048) X(STOCHASTIC_ROUND, 4096) X(GS_ROWS, 16) X(GS_COLS, 32) \
  X(GS_OFFS, 8192) X(NTS_HINT, 16384) X(NO_SCF_QUANT, 1024) X(SIGN_SAT_QUANT, 16)
typedef enum libxsmm_meltw_unary_flags {
#define LIBXSMM_X_(NAME, VALUE) LIBXSMM_MELTW_FLAG_UNARY_##NAME = VALUE,
  LIBXSMM_MELTW_UNARY_FLAG_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  LIBXSMM_MELTW_FLAG_UNARY_LAST_ = 32768
} libxsmm_meltw_unary_flags;

#define LIBXSMM_MELTW_UNARY_TABLE(X) \
  X(NONE, 0) X(IDENTITY, 1) X(XOR, 2) X(X2, 3) X(SQRT, 4) X(RELU, 5) X(RELU_INV, 6) X(TANH, 7) \
  X(TANH_INV, 8) X(SIGMOID, 9) X(SIGMOID_INV, 10) X(GELU, 11) X(GELU_INV, 12) X(NEGATE, 13) \
  X(INC, 14) X(RECIPROCAL, 15) X(RECIPROCAL_SQRT, 16) X(EXP, 17) X(REDUCE_X_OP_ADD, 18) \
  X(REDUCE_X2_OP_ADD, 19) X(REDUCE_X_X2_OP_ADD, 20) X(REDUCE_X_OP_MAX, 21) X(REDUCE_X_OP_MUL, 22) \
  X(REDUCE_X_OP_ADD_NCNC_FORMAT, 23) X(REDUCE_TO_SCALAR_OP_ADD, 24) X(DROPOUT, 25) \
  X(DROPOUT_INV, 26) X(REPLICATE_COL_VAR, 27) X(TRANSFORM_NORM_TO_VNNI2, 28) \
  X(TRANSFORM_NORM_TO_NORMT, 29) X(TRANSFORM_VNNI2_TO_VNNI2T, 30) X(TRANSFORM_NORM_TO_VNNI2T, 31) \
  X(TRANSFORM_NORM_TO_VNNI2_PAD, 32) X(UNZIP, 33) X(LEAKY_RELU, 34) X(LEAKY_RELU_INV, 35) \
  X(ELU, 36) X(ELU_INV, 37) X(STOCHASTIC_ROUND, 38) X(TRANSFORM_PADM_MOD2, 39) \
  X(TRANSFORM_PADN_MOD2, 40) X(TRANSFORM_PADNM_MOD2, 41) X(QUANT, 42) X(DEQUANT, 43) \
  X(REDUCE_COLS_IDX_OP_ADD, 44) X(DECOMPRESS_SPARSE_FACTOR_1, 45) X(DECOMPRESS_SPARSE_FACTOR_2, 46) \
  X(DECOMPRESS_SPARSE_FACTOR_4, 47) X(DECOMPRESS_SPARSE_FACTOR_8, 48) \
  X(DECOMPRESS_SPARSE_FACTOR_16, 49) X(DECOMPRESS_SPARSE_FACTOR_32, 50) X(GATHER, 51) \
  X(SCATTER, 52) X(REDUCE_COLS_IDX_OP_MAX, 53) X(TRANSFORM_NORM_TO_VNNI4, 54) \
  X(TRANSFORM_VNNI4_TO_VNNI4T, 55) X(TRANSFORM_NORM_TO_VNNI4T, 56) X(TRANSFORM_NORM_TO_VNNI4_PAD, 57) \
  X(TRANSFORM_PADM_MOD4, 58) X(TRANSFORM_PADN_MOD4, 59) X(TRANSFORM_PADNM_MOD4, 60) \
  X(TRANSFORM_VNNI4_TO_NORM, 61) X(TRANSFORM_VNNI4_TO_VNNI2, 62) X(DUMP, 63) \
  X(DECOMP_FP32_TO_BF16X2, 64) X(DECOMP_FP32_TO_BF16X3, 65) X(TRANSFORM_VNNI4T_TO_NORM, 66) \
  X(TRANSFORM_VNNI2T_TO_NORM, 67) X(REDUCE_COLS_IDX_OP_MIN, 68) X(REDUCE_X_OP_MIN, 69) \
  X(REDUCE_X_OP_ABSMAX, 70) X(TRANSFORM_NORM_TO_VNNI8, 71) X(TRANSFORM_VNNI8_TO_VNNI8T, 72) \
  X(TRANSFORM_NORM_TO_VNNI8T, 73) X(TRANSFORM_NORM_TO_VNNI8_PAD, 74) \
  X(TRANSFORM_VNNI8T_TO_NORM, 75) X(TRANSFORM_VNNI8_TO_NORM, 76)
typedef enum libxsmm_meltw_unary_type {
#define LIBXSMM_X_(NAME, VALUE) LIBXSMM_MELTW_TYPE_UNARY_##NAME = VALUE,
  LIBXSMM_MELTW_UNARY_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  LIBXSMM_MELTW_TYPE_UNARY_COUNT_ = 77
} libxsmm_meltw_unary_type;

#define LIBXSMM_MELTW_BINARY_FLAG_TABLE(X) \
  X(NONE, 0) X(BCAST_ROW_IN_0, 1) X(BCAST_ROW_IN_1, 2) X(BCAST_COL_IN_0, 4) X(BCAST_COL_IN_1, 8) \
  X(BCAST_SCALAR_IN_0, 16) X(BCAST_SCALAR_IN_1, 32) X(STOCHASTIC_ROUND, 64) \
  X(BITMASK_2BYTEMULT, 128) X(NTS_HINT, 256)
typedef enum libxsmm_meltw_binary_flags {
#define LIBXSMM_X_(NAME, VALUE) LIBXSMM_MELTW_FLAG_BINARY_##NAME = VALUE,
  LIBXSMM_MELTW_BINARY_FLAG_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  LIBXSMM_MELTW_FLAG_BINARY_LAST_ = 512
} libxsmm_meltw_binary_flags;

#define LIBXSMM_MELTW_BINARY_TABLE(X) \
  X(NONE, 0) X(ADD, 1) X(MUL, 2) X(SUB, 3) X(DIV, 4) X(MULADD, 5) X(MATMUL, 6) \
  X(MUL_AND_REDUCE_TO_SCALAR_OP_ADD, 7) X(PACK, 8) X(MAX, 9) X(MIN, 10) X(BRGEMM, 11) \
  X(BRGEMM_B_TRANS, 12) X(BRGEMM_A_TRANS, 13) X(BRGEMM_A_TRANS_B_TRANS, 14) X(BRGEMM_A_VNNI, 15) \
  X(BRGEMM_A_VNNI_B_TRANS, 16) X(BRGEMM_A_VNNI_TRANS, 17) X(BRGEMM_A_VNNI_TRANS_B_TRANS, 18) \
  X(MATMUL_B_TRANS, 19) X(MATMUL_A_TRANS, 20) X(MATMUL_A_TRANS_B_TRANS, 21) X(MATMUL_A_VNNI, 22) \
  X(MATMUL_A_VNNI_B_TRANS, 23) X(MATMUL_A_VNNI_TRANS, 24) X(MATMUL_A_VNNI_TRANS_B_TRANS, 25) \
  X(ZIP, 26) X(CMP_OP_GT, 27) X(CMP_OP_GE, 28) X(CMP_OP_LT, 29) X(CMP_OP_LE, 30) \
  X(CMP_OP_EQ, 31) X(CMP_OP_NE, 32)
typedef enum libxsmm_meltw_binary_type {
#define LIBXSMM_X_(NAME, VALUE) LIBXSMM_MELTW_TYPE_BINARY_##NAME = VALUE,
  LIBXSMM_MELTW_BINARY_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  LIBXSMM_MELTW_TYPE_BINARY_COUNT_ = 33
} libxsmm_meltw_binary_type;

#define LIBXSMM_MELTW_TERNARY_FLAG_TABLE(X) \
  X(NONE, 0) X(BCAST_ROW_IN_0, 1) X(BCAST_ROW_IN_1, 2) X(BCAST_ROW_IN_2, 4) X(BCAST_COL_IN_0, 8) \
  X(BCAST_COL_IN_1, 16) X(BCAST_COL_IN_2, 32) X(BCAST_SCALAR_IN_0, 64) X(BCAST_SCALAR_IN_1, 128) \
  X(BCAST_SCALAR_IN_2, 256) X(REUSE_IN_2_AS_OUT, 512) X(BITMASK_2BYTEMULT, 1024) \
  X(STOCHASTIC_ROUND, 2048)
typedef enum libxsmm_meltw_ternary_flags {
#define LIBXSMM_X_(NAME, VALUE) LIBXSMM_MELTW_FLAG_TERNARY_##NAME = VALUE,
  LIBXSMM_MELTW_TERNARY_FLAG_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  LIBXSMM_MELTW_FLAG_TERNARY_LAST_ = 4096
} libxsmm_meltw_ternary_flags;

#define LIBXSMM_MELTW_TERNARY_TABLE(X) \
  X(NONE, 0) X(MULADD, 1) X(MATMUL, 2) X(SELECT, 3) X(NMULADD, 4) X(BRGEMM, 5) X(BRGEMM_B_TRANS, 6) \
  X(BRGEMM_A_TRANS, 7) X(BRGEMM_A_TRANS_B_TRANS, 8) X(BRGEMM_A_VNNI, 9) X(BRGEMM_A_VNNI_B_TRANS, 10) \
  X(BRGEMM_A_VNNI_TRANS, 11) X(BRGEMM_A_VNNI_TRANS_B_TRANS, 12) X(MATMUL_B_TRANS, 13) \
  X(MATMUL_A_TRANS, 14) X(MATMUL_A_TRANS_B_TRANS, 15) X(MATMUL_A_VNNI, 16) \
  X(MATMUL_A_VNNI_B_TRANS, 17) X(MATMUL_A_VNNI_TRANS, 18) X(MATMUL_A_VNNI_TRANS_B_TRANS, 19)
typedef enum libxsmm_meltw_ternary_type {
#define LIBXSMM_X_(NAME, VALUE) LIBXSMM_MELTW_TYPE_TERNARY_##NAME = VALUE,
  LIBXSMM_MELTW_TERNARY_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  LIBXSMM_MELTW_TYPE_TERNARY_COUNT_ = 20
} libxsmm_meltw_ternary_type;

/* ---- GEMM flags  [ref: include/libxsmm_typedefs.h:446-551] ----------------------- */
#define LIBXSMM_GEMM_FLAG_TABLE(X) \
  X(NONE, 0) X(TRANS_A, 1) X(TRANS_B, 2) X(TRANS_AB, 3) X(BETA_0, 4) X(ALIGN_A, 8) X(ALIGN_C, 16) \
  X(ALIGN_C_NTS_HINT, 32 | 16) X(NO_RESET_TILECONFIG, 64) X(NO_SETUP_TILECONFIG, 128) \
  X(VNNI_A, 256) X(VNNI_B, 512) X(VNNI_C, 1024) X(USE_XGEMM_ABI, 2048) X(USE_XGEMM_EXT_ABI, 4096) \
  X(DESC_ISBIG, 8192) X(BATCH_REDUCE_ADDRESS, 8192) X(BATCH_REDUCE_OFFSET, 16384) \
  X(BATCH_REDUCE_STRIDE, 32768) X(USE_COL_VEC_SCF, 65536) X(USE_COL_VEC_ZPT, 131072) \
  X(INTLV_A_FORMAT, 262144) X(DECOMPRESS_A_VIA_BITMASK, 524288) X(USE_MxK_ZPT, 1048576) \
  X(USE_MxK_SCF, 2097152) X(INVALID, 4194304)
typedef enum libxsmm_gemm_flags {
#define LIBXSMM_X_(NAME, VALUE) LIBXSMM_GEMM_FLAG_##NAME = (VALUE),
  LIBXSMM_GEMM_FLAG_TABLE(LIBXSMM_X_)
#undef LIBXSMM_X_
  /* combined convenience flags of the reference */
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BETA_0 = 4 | 48,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BATCH_REDUCE_ADDRESS = 8192 | 48,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BETA_0_BATCH_REDUCE_ADDRESS = 4 | 48 | 8192,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BATCH_REDUCE_OFFSET = 16384 | 48,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BETA_0_BATCH_REDUCE_OFFSET = 4 | 48 | 16384,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BATCH_REDUCE_STRIDE = 32768 | 48,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BETA_0_BATCH_REDUCE_STRIDE = 4 | 48 | 32768
} libxsmm_gemm_flags;
typedef enum libxsmm_basic_gemm_flags {
  LIBXSMM_BASIC_GEMM_FLAG_NONE = 0, LIBXSMM_BASIC_GEMM_FLAG_TRANS_A = 1, LIBXSMM_BASIC_GEMM_FLAG_TRANS_B = 2,
  LIBXSMM_BASIC_GEMM_FLAG_TRANS_AB = 3, LIBXSMM_BASIC_GEMM_FLAG_BETA_0 = 4, LIBXSMM_BASIC_GEMM_FLAG_ALIGN_A = 8,
  LIBXSMM_BASIC_GEMM_FLAG_ALIGN_C = 16, LIBXSMM_BASIC_GEMM_FLAG_ALIGN_C_NTS_HINT = 1024 | 16,
  LIBXSMM_BASIC_GEMM_FLAG_INVALID = 524288
} libxsmm_basic_gemm_flags;
/* 'N'/'n' means "not transposed"; anything else transposes [ref: include/libxsmm_macros.h:279-282] */
#define LIBXSMM_GEMM_FLAGS(TRANSA, TRANSB) ((libxsmm_bitfield)( \
  (('n' == (TRANSA) || 'N' == (TRANSA)) ? 0 : LIBXSMM_GEMM_FLAG_TRANS_A) | \
  (('n' == (TRANSB) || 'N' == (TRANSB)) ? 0 : LIBXSMM_GEMM_FLAG_TRANS_B)))

/* transposes given by address: a NULL request falls back to what DEFAULT says; all other bits of DEFAULT are kept */
#define LIBXSMM_GEMM_PFLAGS(TRANSA, TRANSB, DEFAULT) ((int)(LIBXSMM_GEMM_FLAGS( \
  (0 != ((const void*)(TRANSA)) ? *((const char*)(TRANSA)) : ((0 != (LIBXSMM_GEMM_FLAG_TRANS_A & (DEFAULT))) ? 't' : 'n')), \
  (0 != ((const void*)(TRANSB)) ? *((const char*)(TRANSB)) : ((0 != (LIBXSMM_GEMM_FLAG_TRANS_B & (DEFAULT))) ? 't' : 'n'))) \
  | ((DEFAULT) & ~(LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B))))
#define LIBXSMM_GEMM_VNNI_FLAGS(TRANSA, TRANSB, VNNIA, VNNIB) ((libxsmm_bitfield)(LIBXSMM_GEMM_FLAGS(TRANSA, TRANSB) | \
  (('n' == (VNNIA) || 'N' == (VNNIA)) ? 0 : LIBXSMM_GEMM_FLAG_VNNI_A) | (('n' == (VNNIB) || 'N' == (VNNIB)) ? 0 : LIBXSMM_GEMM_FLAG_VNNI_B)))

typedef enum libxsmm_gemm_prefetch_type {
  LIBXSMM_GEMM_PREFETCH_NONE = 0, LIBXSMM_GEMM_PREFETCH_AL2 = 1, LIBXSMM_GEMM_PREFETCH_BL2 = 2
} libxsmm_gemm_prefetch_type;
typedef enum libxsmm_gemm_batch_reduce_type {
  LIBXSMM_GEMM_BATCH_REDUCE_NONE = 0, LIBXSMM_GEMM_BATCH_REDUCE_ADDRESS = 1,
  LIBXSMM_GEMM_BATCH_REDUCE_OFFSET = 2, LIBXSMM_GEMM_BATCH_REDUCE_STRIDE = 4
} libxsmm_gemm_batch_reduce_type;
typedef enum libxsmm_kernel_kind {
  LIBXSMM_KERNEL_KIND_MATMUL = 0, LIBXSMM_KERNEL_KIND_MELTW = 1, LIBXSMM_KERNEL_KIND_MEQN = 2,
  LIBXSMM_KERNEL_KIND_USER = 3, LIBXSMM_KERNEL_UNREGISTERED = 4
} libxsmm_kernel_kind;

/* ---- run-time argument structs  [ref: include/libxsmm_typedefs.h:570-725] ---------
 * Slot usage on this backend is documented in DESIGN.md "param slots".            */
typedef struct libxsmm_matrix_arg { void *primary, *secondary, *tertiary, *quaternary, *quinary, *senary; } libxsmm_matrix_arg;
typedef struct libxsmm_matrix_op_arg { void *primary, *secondary, *tertiary, *quaternary; } libxsmm_matrix_op_arg;

typedef struct libxsmm_meltw_unary_shape {
  libxsmm_blasint m, n, ldi, ldo;
  libxsmm_datatype in0_type, out_type, comp_type;
} libxsmm_meltw_unary_shape;
typedef struct libxsmm_meltw_binary_shape {
  libxsmm_blasint m, n, ldi, ldi2, ldo;
  libxsmm_datatype in0_type, in1_type, out_type, comp_type;
} libxsmm_meltw_binary_shape;
typedef struct libxsmm_meltw_ternary_shape {
  libxsmm_blasint m, n, ldi, ldi2, ldi3, ldo;
  libxsmm_datatype in0_type, in1_type, in2_type, out_type, comp_type;
} libxsmm_meltw_ternary_shape;

typedef struct libxsmm_meltw_unary_param   { libxsmm_matrix_op_arg op; libxsmm_matrix_arg in, out; } libxsmm_meltw_unary_param;
typedef struct libxsmm_meltw_binary_param  { libxsmm_matrix_op_arg op; libxsmm_matrix_arg in0, in1, out; } libxsmm_meltw_binary_param;
typedef struct libxsmm_meltw_ternary_param { libxsmm_matrix_op_arg op; libxsmm_matrix_arg in0, in1, in2, out; } libxsmm_meltw_ternary_param;

typedef void (*libxsmm_meltwfunction_unary)(const libxsmm_meltw_unary_param*);
typedef void (*libxsmm_meltwfunction_binary)(const libxsmm_meltw_binary_param*);
typedef void (*libxsmm_meltwfunction_ternary)(const libxsmm_meltw_ternary_param*);
typedef union libxsmm_xmeltwfunction {
  void (*xmeltw)(const void*);
  libxsmm_meltwfunction_unary meltw_unary;
  libxsmm_meltwfunction_binary meltw_binary;
  libxsmm_meltwfunction_ternary meltw_ternary;
} libxsmm_xmeltwfunction;

typedef struct libxsmm_gemm_param     { libxsmm_matrix_op_arg op; libxsmm_matrix_arg a, b, c; } libxsmm_gemm_param;
typedef struct libxsmm_gemm_ext_param { libxsmm_matrix_op_arg op; libxsmm_matrix_arg a, b, c, d, ap, bp, cp; } libxsmm_gemm_ext_param;

typedef struct libxsmm_gemm_shape {
  libxsmm_blasint m, n, k, lda, ldb, ldc;
  libxsmm_datatype a_in_type, b_in_type, out_type, comp_type;
} libxsmm_gemm_shape;
typedef struct libxsmm_gemm_batch_reduce_config {
  libxsmm_gemm_batch_reduce_type br_type;
  libxsmm_blasint br_stride_a_hint, br_stride_b_hint;   /* BYTES (stride mode) */
  unsigned char br_unroll_hint;
} libxsmm_gemm_batch_reduce_config;
typedef struct libxsmm_spgemm_config { libxsmm_blasint packed_width, bk, bn; } libxsmm_spgemm_config;
typedef struct libxsmm_gemm_ext_unary_argops {
  libxsmm_blasint ldap; libxsmm_meltw_unary_type ap_unary_type; libxsmm_bitfield ap_unary_flags; libxsmm_blasint store_ap;
  libxsmm_blasint ldbp; libxsmm_meltw_unary_type bp_unary_type; libxsmm_bitfield bp_unary_flags; libxsmm_blasint store_bp;
  libxsmm_blasint ldcp; libxsmm_meltw_unary_type cp_unary_type; libxsmm_bitfield cp_unary_flags; libxsmm_blasint store_cp;
} libxsmm_gemm_ext_unary_argops;
typedef struct libxsmm_gemm_ext_binary_postops {
  libxsmm_blasint ldd; libxsmm_datatype d_in_type; libxsmm_meltw_binary_type d_binary_type; libxsmm_bitfield d_binary_flags;
} libxsmm_gemm_ext_binary_postops;
typedef struct libxsmm_tilecfg_state { unsigned char tileconfig[64]; } libxsmm_tilecfg_state;

typedef void (*libxsmm_dmmfunction)(const double* a, const double* b, double* c);
typedef void (*libxsmm_smmfunction)(const float* a, const float* b, float* c);
typedef void (*libxsmm_gemmfunction)(const libxsmm_gemm_param*);
typedef void (*libxsmm_gemmfunction_ext)(const libxsmm_gemm_ext_param*);
typedef void (*libxsmm_tilecfgfunction)(const libxsmm_tilecfg_state*);
typedef union libxsmm_xmmfunction {
  const void* ptr_const; void* ptr;
  void (*xmm)(const void*, const void*, void*);
  void (*xgemm)(const void*);
  libxsmm_dmmfunction dmm; libxsmm_smmfunction smm;
  libxsmm_gemmfunction gemm; libxsmm_gemmfunction_ext gemm_ext;
  libxsmm_tilecfgfunction tilecfg;
} libxsmm_xmmfunction;

typedef struct libxsmm_mmkernel_info {
  libxsmm_datatype iprecision, oprecision;
  libxsmm_gemm_prefetch_type prefetch;
  unsigned int lda, ldb, ldc, m, n, k;
  int flags;
} libxsmm_mmkernel_info;
typedef struct libxsmm_meltwkernel_info { unsigned int ldi, ldo, m, n, datatype, flags, operation; } libxsmm_meltwkernel_info;
typedef struct libxsmm_kernel_info {
  libxsmm_kernel_kind kind;
  unsigned int nflops;
  size_t code_size;
  unsigned int is_reference_kernel;   /* always 0 here: there is no CPU fallback */
} libxsmm_kernel_info;
typedef struct libxsmm_registry_info { size_t capacity, size, nbytes, nstatic, ncache; } libxsmm_registry_info;

/* ---- target "architecture" ids  [ref: include/libxsmm_cpuid.h:23-39] --------------
 * This backend reports LIBXSMM_X86_GENERIC (the reference's id for LIBXSMM_TARGET=generic), below LIBXSMM_X86_AVX512_SPR, so that callers do
 * not hoist AMX tile configuration. */
#define LIBXSMM_TARGET_ARCH_UNKNOWN 0
#define LIBXSMM_TARGET_ARCH_GENERIC 1
#define LIBXSMM_X86_GENERIC         1002
#define LIBXSMM_X86_AVX512_SPR      1104
#define LIBXSMM_HIP_GFX950          950

/* ---- library life cycle & global state  [ref: include/libxsmm.h:63-100] ----------- */
LIBXSMM_EXTERN unsigned int libxsmm_ninit;
LIBXSMM_EXTERN int libxsmm_verbosity;
LIBXSMM_EXTERN int libxsmm_target_archid;
LIBXSMM_EXTERN int libxsmm_stdio_handle;
LIBXSMM_EXTERN int libxsmm_se;
LIBXSMM_API void libxsmm_init(void);
LIBXSMM_API void libxsmm_finalize(void);
LIBXSMM_API int libxsmm_get_target_archid(void);
LIBXSMM_API void libxsmm_set_target_archid(int id);
LIBXSMM_API const char* libxsmm_get_target_arch(void);
LIBXSMM_API void libxsmm_set_target_arch(const char* arch);
LIBXSMM_API const char* libxsmm_get_typename(libxsmm_datatype datatype);
LIBXSMM_API int libxsmm_get_verbosity(void);
LIBXSMM_API void libxsmm_set_verbosity(int level);
LIBXSMM_API int libxsmm_cpuid(void* info);
LIBXSMM_API int libxsmm_cpuid_dot_pack_factor(libxsmm_datatype datatype);   /* bf16: 2, i8: 4 */
LIBXSMM_API int libxsmm_cpuid_vlen(int id);                                /* bytes: 64 */
LIBXSMM_API int libxsmm_cpuid_vlen32(int id);                              /* 32-bit lanes: 16 (rows per DROPOUT draw) [ref: include/libxsmm_cpuid.h:123] */

/* ---- kernel introspection / lifetime  [ref: include/libxsmm.h:94-104,226-229] ----- */
LIBXSMM_API int libxsmm_get_mmkernel_info(libxsmm_xmmfunction kernel, libxsmm_mmkernel_info* info);
LIBXSMM_API int libxsmm_get_meltwkernel_info(libxsmm_xmeltwfunction kernel, libxsmm_meltwkernel_info* info);
LIBXSMM_API int libxsmm_get_kernel_info(const void* kernel, libxsmm_kernel_info* info);
LIBXSMM_API int libxsmm_get_registry_info(libxsmm_registry_info* info);
LIBXSMM_API void libxsmm_release_kernel(const void* kernel);

/* ---- shape/config constructors  [ref: src/libxsmm_generator.c:323-460] ------------ */
LIBXSMM_API libxsmm_gemm_shape libxsmm_create_gemm_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  libxsmm_datatype a_in_type, libxsmm_datatype b_in_type, libxsmm_datatype out_type, libxsmm_datatype comp_type);
LIBXSMM_API libxsmm_gemm_batch_reduce_config libxsmm_create_gemm_batch_reduce_config(libxsmm_gemm_batch_reduce_type br_type,
  libxsmm_blasint br_stride_a_hint, libxsmm_blasint br_stride_b_hint, unsigned char br_unroll_hint);
LIBXSMM_API libxsmm_gemm_ext_unary_argops libxsmm_create_gemm_ext_unary_argops(
  libxsmm_blasint ldap, libxsmm_meltw_unary_type ap_unary_type, libxsmm_bitfield ap_unary_flags, libxsmm_blasint store_ap,
  libxsmm_blasint ldbp, libxsmm_meltw_unary_type bp_unary_type, libxsmm_bitfield bp_unary_flags, libxsmm_blasint store_bp,
  libxsmm_blasint ldcp, libxsmm_meltw_unary_type cp_unary_type, libxsmm_bitfield cp_unary_flags, libxsmm_blasint store_cp);
LIBXSMM_API libxsmm_gemm_ext_binary_postops libxsmm_create_gemm_ext_binary_postops(libxsmm_blasint ldd,
  libxsmm_datatype d_in_type, libxsmm_meltw_binary_type d_binary_type, libxsmm_bitfield d_binary_flags);
LIBXSMM_API libxsmm_meltw_unary_shape libxsmm_create_meltw_unary_shape(libxsmm_blasint m, libxsmm_blasint n,
  libxsmm_blasint ldi, libxsmm_blasint ldo, libxsmm_datatype in0_type, libxsmm_datatype out_type, libxsmm_datatype comp_type);
LIBXSMM_API libxsmm_meltw_binary_shape libxsmm_create_meltw_binary_shape(libxsmm_blasint m, libxsmm_blasint n,
  libxsmm_blasint ldi, libxsmm_blasint ldi2, libxsmm_blasint ldo,
  libxsmm_datatype in0_type, libxsmm_datatype in1_type, libxsmm_datatype out_type, libxsmm_datatype comp_type);
LIBXSMM_API libxsmm_meltw_ternary_shape libxsmm_create_meltw_ternary_shape(libxsmm_blasint m, libxsmm_blasint n,
  libxsmm_blasint ldi, libxsmm_blasint ldi2, libxsmm_blasint ldi3, libxsmm_blasint ldo,
  libxsmm_datatype in0_type, libxsmm_datatype in1_type, libxsmm_datatype in2_type, libxsmm_datatype out_type, libxsmm_datatype comp_type);

/* ---- descriptor initialisers (low-level)  [ref: include/libxsmm_generator.h:101-170] */
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_datatype a_type, libxsmm_datatype b_type, libxsmm_datatype comp_type, libxsmm_datatype c_type,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init_gemm(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_shape gemm_shape, libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init_brgemm(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_shape gemm_shape, libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags,
  libxsmm_gemm_batch_reduce_config brgemm_config);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init_brgemm_ext(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_shape gemm_shape, libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags,
  libxsmm_gemm_batch_reduce_config brgemm_config, libxsmm_gemm_ext_unary_argops unary_argops,
  libxsmm_gemm_ext_binary_postops binary_postops);
LIBXSMM_API libxsmm_meltw_descriptor* libxsmm_meltw_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_datatype in_type, libxsmm_datatype out_type, libxsmm_blasint m, libxsmm_blasint n,
  libxsmm_blasint ldi, libxsmm_blasint ldo, unsigned short flags, unsigned short param, unsigned char operation);
LIBXSMM_API libxsmm_meltw_descriptor* libxsmm_meltw_descriptor_init2(libxsmm_descriptor_blob* blob,
  libxsmm_datatype in0_type, libxsmm_datatype in1_type, libxsmm_datatype in2_type, libxsmm_datatype comp_type,
  libxsmm_datatype out_type, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ldi, libxsmm_blasint ldo,
  libxsmm_blasint ldi2, libxsmm_blasint ldi3, unsigned short flags, unsigned short param, unsigned char operation);

/* ---- dispatch: descriptor -> cached handle  [ref: include/libxsmm.h:125-147] ------
 * NULL == unsupported / illegal combination; never aborts; mute unless LIBXSMM_VERBOSE. */
LIBXSMM_API libxsmm_xmmfunction libxsmm_xmmdispatch(const libxsmm_gemm_descriptor* descriptor);
LIBXSMM_API libxsmm_gemmfunction libxsmm_dispatch_gemm(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags);
LIBXSMM_API libxsmm_gemmfunction libxsmm_dispatch_brgemm(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_gemm_batch_reduce_config brgemm_config);
LIBXSMM_API libxsmm_gemmfunction_ext libxsmm_dispatch_brgemm_ext(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_gemm_batch_reduce_config brgemm_config,
  libxsmm_gemm_ext_unary_argops unary_argops, libxsmm_gemm_ext_binary_postops binary_postops);
LIBXSMM_API libxsmm_tilecfgfunction libxsmm_dispatch_tilecfg_gemm(libxsmm_gemm_shape gemm_shape, libxsmm_bitfield gemm_flags);
LIBXSMM_API libxsmm_xmeltwfunction libxsmm_dispatch_meltw(const libxsmm_meltw_descriptor* descriptor);
LIBXSMM_API libxsmm_meltwfunction_unary libxsmm_dispatch_meltw_unary(libxsmm_meltw_unary_type unary_type,
  libxsmm_meltw_unary_shape unary_shape, libxsmm_bitfield unary_flags);
LIBXSMM_API libxsmm_meltwfunction_binary libxsmm_dispatch_meltw_binary(libxsmm_meltw_binary_type binary_type,
  libxsmm_meltw_binary_shape binary_shape, libxsmm_bitfield binary_flags);
LIBXSMM_API libxsmm_meltwfunction_ternary libxsmm_dispatch_meltw_ternary(libxsmm_meltw_ternary_type ternary_type,
  libxsmm_meltw_ternary_shape ternary_shape, libxsmm_bitfield ternary_flags);

/* ---- BLAS-style small GEMM (column-major; alpha is taken as 1, beta as 0 or 1, NULL n/k/ld* default as in BLAS-less LIBXSMM)
 * [ref: src/libxsmm_main.c:3933-3949, src/libxsmm_main.h:215-240] ------------------- */
LIBXSMM_API void libxsmm_dgemm(const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const double* alpha, const double* a, const libxsmm_blasint* lda, const double* b, const libxsmm_blasint* ldb,
  const double* beta, double* c, const libxsmm_blasint* ldc);
LIBXSMM_API void libxsmm_sgemm(const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const float* alpha, const float* a, const libxsmm_blasint* lda, const float* b, const libxsmm_blasint* ldb,
  const float* beta, float* c, const libxsmm_blasint* ldc);

/* ---- matrix equations: trees of TPPs evaluated as one kernel handle
 * [ref: include/libxsmm.h:149-162; include/libxsmm_typedefs.h:586-591,617-657,683-694; src/libxsmm_matrixeqn.c;
 *  semantics = src/generator_matequation_reference_impl.c:105-227: the tree is evaluated bottom-up, every op node is the
 *  TPP of its type with comp/out type = the op's dtype, intermediate shape per libxsmm_matrixeqn.c:869-936, the root
 *  writes `out_shape`].  Ops are pushed in pre-order (an op, then its operands left to right). ------------------- */
typedef struct libxsmm_meqn_arg_shape { libxsmm_blasint m, n, ld; libxsmm_datatype type; } libxsmm_meqn_arg_shape;
typedef enum libxsmm_matrix_arg_type { LIBXSMM_MATRIX_ARG_TYPE_SINGULAR = 0, LIBXSMM_MATRIX_ARG_TYPE_SET = 1 } libxsmm_matrix_arg_type;
typedef enum libxsmm_matrix_arg_set_type {
  LIBXSMM_MATRIX_ARG_SET_TYPE_NONE = 0, LIBXSMM_MATRIX_ARG_SET_TYPE_ABS_ADDRESS = 1,
  LIBXSMM_MATRIX_ARG_SET_TYPE_OFFSET_BASE = 2, LIBXSMM_MATRIX_ARG_SET_TYPE_STRIDE_BASE = 3
} libxsmm_matrix_arg_set_type;
typedef struct libxsmm_matrix_arg_attributes {
  libxsmm_matrix_arg_type type; libxsmm_matrix_arg_set_type set_type; libxsmm_blasint set_cardinality_hint, set_stride_hint;
} libxsmm_matrix_arg_attributes;
typedef struct libxsmm_meqn_op_metadata { libxsmm_blasint eqn_idx, op_arg_pos; } libxsmm_meqn_op_metadata;
typedef struct libxsmm_meqn_arg_metadata { libxsmm_blasint eqn_idx, in_arg_pos; } libxsmm_meqn_arg_metadata;
typedef struct libxsmm_meqn_param {
  const libxsmm_matrix_op_arg* ops_args;   /* per-op state (e.g. alpha of LEAKY_RELU), indexed by op_arg_pos */
  const libxsmm_matrix_arg* inputs;        /* inputs[in_arg_pos].primary: device-accessible */
  libxsmm_matrix_arg output;
} libxsmm_meqn_param;
typedef void (*libxsmm_meqn_function)(const libxsmm_meqn_param*);

LIBXSMM_API libxsmm_blasint libxsmm_meqn_create(void);
LIBXSMM_API libxsmm_meqn_arg_shape libxsmm_create_meqn_arg_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ld, libxsmm_datatype type);
LIBXSMM_API libxsmm_matrix_arg_attributes libxsmm_create_matrix_arg_attributes(libxsmm_matrix_arg_type type, libxsmm_matrix_arg_set_type set_type,
  libxsmm_blasint set_cardinality_hint, libxsmm_blasint set_stride_hint);
LIBXSMM_API libxsmm_meqn_arg_metadata libxsmm_create_meqn_arg_metadata(libxsmm_blasint eqn_idx, libxsmm_blasint in_arg_pos);
LIBXSMM_API libxsmm_meqn_op_metadata libxsmm_create_meqn_op_metadata(libxsmm_blasint eqn_idx, libxsmm_blasint op_arg_pos);
LIBXSMM_API int libxsmm_meqn_push_back_arg(libxsmm_meqn_arg_metadata arg_metadata, libxsmm_meqn_arg_shape arg_shape, libxsmm_matrix_arg_attributes arg_attr);
LIBXSMM_API int libxsmm_meqn_push_back_unary_op(libxsmm_meqn_op_metadata op_metadata, libxsmm_meltw_unary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags);
LIBXSMM_API int libxsmm_meqn_push_back_binary_op(libxsmm_meqn_op_metadata op_metadata, libxsmm_meltw_binary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags);
LIBXSMM_API int libxsmm_meqn_push_back_ternary_op(libxsmm_meqn_op_metadata op_metadata, libxsmm_meltw_ternary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags);
LIBXSMM_API void libxsmm_meqn_tree_print(libxsmm_blasint idx);
LIBXSMM_API void libxsmm_meqn_rpn_print(libxsmm_blasint idx);
/** NULL if the tree is incomplete or contains an op outside this backend (LIBXSMM_VERBOSE >= 1 names the node).  MATMUL / BRGEMM nodes run the
 * dense GEMM kernels (BRGEMM: A and B are STRIDE_BASE argument sets, the block count is read from ops_args[op_arg_pos].tertiary; the ternary
 * forms need REUSE_IN_2_AS_OUT and accumulate into their third operand in place); a GATHER node directly above an argument reads its index
 * list from inputs[pos].secondary; DUMP writes to ops_args[op_arg_pos].primary. */
LIBXSMM_API libxsmm_meqn_function libxsmm_dispatch_meqn(libxsmm_blasint idx, libxsmm_meqn_arg_shape out_shape);

/* ---- packed / sparse creators (caller-owned, release with libxsmm_release_kernel)
 * [ref: include/libxsmm.h:164-223; src/libxsmm_main.c:3553-3883] --------------------- */
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_spgemm_csr(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_blasint packed_width,
  const unsigned int* row_ptr, const unsigned int* column_idx, const void* values);
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_spgemm_csc(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_blasint packed_width,
  const unsigned int* column_ptr, const unsigned int* row_idx, const void* values);
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_spgemm_bcsc(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_spgemm_config spgemm_config);
LIBXSMM_API libxsmm_tilecfgfunction libxsmm_create_tilecfg_packed_spgemm_bcsc(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_spgemm_config spgemm_config);
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_spgemm_csr_areg(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_blasint max_N,
  const unsigned int* row_ptr, const unsigned int* column_idx, const double* values);

/* ---- FsSpMDM frontend  [ref: include/libxsmm_fsspmdm.h:16-45] --------------------- */
typedef struct libxsmm_fsspmdm libxsmm_fsspmdm;
#define libxsmm_dfsspmdm libxsmm_fsspmdm
#define libxsmm_sfsspmdm libxsmm_fsspmdm
LIBXSMM_API libxsmm_fsspmdm* libxsmm_fsspmdm_create(libxsmm_datatype datatype,
  libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  const void* alpha, const void* beta, const void* a_dense, int c_is_nt, libxsmm_timer_tickint (*timer_tick)(void));
LIBXSMM_API libxsmm_dfsspmdm* libxsmm_dfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, double alpha, double beta, const double* a_dense,
  int c_is_nt, libxsmm_timer_tickint (*timer_tick)(void));
LIBXSMM_API libxsmm_sfsspmdm* libxsmm_sfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, float alpha, float beta, const float* a_dense,
  int c_is_nt, libxsmm_timer_tickint (*timer_tick)(void));
LIBXSMM_API void libxsmm_fsspmdm_execute(const libxsmm_fsspmdm* handle, const void* B, void* C);
LIBXSMM_API void libxsmm_dfsspmdm_execute(const libxsmm_dfsspmdm* handle, const double* B, double* C);
LIBXSMM_API void libxsmm_sfsspmdm_execute(const libxsmm_sfsspmdm* handle, const float* B, float* C);
LIBXSMM_API void libxsmm_fsspmdm_destroy(libxsmm_fsspmdm* handle);
LIBXSMM_API void libxsmm_dfsspmdm_destroy(libxsmm_dfsspmdm* handle);
LIBXSMM_API void libxsmm_sfsspmdm_destroy(libxsmm_sfsspmdm* handle);

/* ---- helpers the reference's drivers link against (host side, CPU) ----------------
 * [ref: include/libxsmm_malloc.h, libxsmm_math.h, utils/libxsmm_timer.h, libxsmm_rng.h] */
LIBXSMM_API void* libxsmm_aligned_malloc(size_t size, size_t alignment);   /* pinned, device-visible */
LIBXSMM_API void* libxsmm_malloc(size_t size);
LIBXSMM_API void  libxsmm_free(const void* memory);
LIBXSMM_API libxsmm_timer_tickint libxsmm_timer_tick(void);
LIBXSMM_API double libxsmm_timer_duration(libxsmm_timer_tickint tick0, libxsmm_timer_tickint tick1);
LIBXSMM_API void libxsmm_rng_set_seed(unsigned int seed);
LIBXSMM_API double libxsmm_rng_f64(void);
LIBXSMM_API unsigned int libxsmm_rng_u32(unsigned int n);
LIBXSMM_API void libxsmm_rng_seq(void* data, size_t nbytes);
LIBXSMM_API void libxsmm_rng_f32_seq(float* rngs, libxsmm_blasint count);
LIBXSMM_API float libxsmm_convert_bf16_to_f32(libxsmm_bfloat16 in);
LIBXSMM_API libxsmm_bfloat16 libxsmm_convert_f32_to_bf16_rne(float in);
LIBXSMM_API libxsmm_bfloat16 libxsmm_convert_f32_to_bf16_truncate(float in);
LIBXSMM_API void libxsmm_rne_convert_fp32_bf16(const float* in, libxsmm_bfloat16* out, size_t length);
LIBXSMM_API void libxsmm_truncate_convert_f32_bf16(const float* in, libxsmm_bfloat16* out, size_t length);
LIBXSMM_API void libxsmm_convert_bf16_f32(const libxsmm_bfloat16* in, float* out, size_t length);

/* Matrix comparison in the style of libxsmm_matdiff [ref: include/libxsmm_math.h:60-110;
 * src/libxsmm_matdiff.h:141-142]: normf_rel is the metric the reference's tests bound. */
typedef struct libxsmm_matdiff_info {
  double norm1_abs, norm1_rel, normi_abs, normi_rel, normf_rel, linf_abs, linf_rel, l2_abs, l2_rel, rsq;
  double l1_ref, min_ref, max_ref, avg_ref, var_ref, l1_tst, min_tst, max_tst, avg_tst, var_tst;
  double v_ref, v_tst;
  libxsmm_blasint m, n, i, r;
} libxsmm_matdiff_info;
LIBXSMM_API int libxsmm_matdiff(libxsmm_matdiff_info* info, libxsmm_datatype datatype, libxsmm_blasint m, libxsmm_blasint n,
  const void* ref, const void* tst, const libxsmm_blasint* ldref, const libxsmm_blasint* ldtst);
LIBXSMM_API void libxsmm_matdiff_clear(libxsmm_matdiff_info* info);
/** Combines the result of one comparison into a running worst case [ref: src/libxsmm_math.c:386-446]. */
LIBXSMM_API void libxsmm_matdiff_reduce(libxsmm_matdiff_info* output, const libxsmm_matdiff_info* input);
LIBXSMM_API double libxsmm_matdiff_epsilon(const libxsmm_matdiff_info* input);

/**
 * Dense packed GEMMs (SOA layouts, packed width fastest); caller owned, release with libxsmm_release_kernel.
 * packed:  A [K][lda][P], B [N][ldb][P], C [N][ldc][P]:  C[n][m][p] (+)= sum_k A[k][m][p] * B[n][k][p]
 * ac_rm:   A [M][lda][P], B row-major [K][ldb] (not packed), C [M][ldc][P]:  C[m][n][p] (+)= sum_k A[m][k][p] * B[k][n]
 * bc_rm:   A row-major [M][lda] (not packed), B [K][ldb][P], C [M][ldc][P]:  C[m][n][p] (+)= sum_k A[m][k] * B[k][n][p]
 * F32 / F64.  [ref: include/libxsmm.h:190-214; src/libxsmm_main.c:3733-3840; gold loops in
 * samples/xgemm_packed/gemm_packed_kernel.c:35-72, samples/xgemm_norm_packed/dense_packedacrm.c:20-58, dense_packedbcrm.c:20-58]
 */
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_gemm(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_blasint packed_width);
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_gemm_ac_rm(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_blasint packed_width);
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_gemm_bc_rm(libxsmm_gemm_shape gemm_shape,
  libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags, libxsmm_blasint packed_width);

#include "libxsmm_hip.h"

#endif /* LIBXSMM_H */
