/*
 * oracle.h -- CPU restatement of the reference's algorithms for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it, and
 * only as the checker.  libxsmm_amd never links against it and has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle_pin.py checks every entry point of this file
 * (a) against the reference itself (oracle/_ref/libxsmm_ref.so, built by oracle/Makefile
 *     from the sources under /root/reference, when that tree is present) on seeded
 *     random inputs, and
 * (b) against the committed fixtures in tests/golden/ that were generated from the
 *     reference by tests/golden/make_golden.py.
 *
 * Each function cites the reference file:line whose behaviour it restates.  The code is
 * written with small accessor helpers instead of the reference's per-dtype loop nests.
 * Compile with -ffp-contract=off: the reference's C loops are built without FMA
 * contraction on the x86-64 baseline, and the order/rounding of every a*b and += below is
 * part of the contract.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/libxsmm.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* ---- dense GEMM / BRGEMM (+ fused ext epilogue) ------------------------------------- */
typedef struct oracle_gemm_desc {
  int m, n, k, lda, ldb, ldc;
  int a_type, b_type, c_type, comp_type;     /* libxsmm_datatype */
  unsigned int flags;                        /* libxsmm_gemm_flags incl. BATCH_REDUCE_* and *_ABI */
  long long br_stride_a, br_stride_b;        /* bytes (stride mode) */
  int colbias;                               /* 1: C = bcast_col(D) (+C) before the GEMM       */
  int act;                                   /* 0 none, 1 relu, 2 relu+bitmask, 3 sigmoid      */
} oracle_gemm_desc;

/* param is a libxsmm_gemm_param, or a libxsmm_gemm_ext_param if flags has USE_XGEMM_EXT_ABI.
 * [ref: src/generator_gemm_reference_impl.c:2817-2853 libxsmm_reference_gemm] */
void oracle_gemm(const void* param, const oracle_gemm_desc* desc);

/* Same contraction order as the MFMA kernels use, for bit-exact kernel debugging:
 * fmaf chain over k in natural order (what v_mfma_f32_32x32x2_f32 computes). */
void oracle_gemm_f32_fma(const void* param, const oracle_gemm_desc* desc);

/* ---- element-wise TPPs --------------------------------------------------------------- */
typedef struct oracle_meltw_desc {
  int m, n, ldi, ldo, ldi2, ldi3;
  int in0_type, in1_type, in2_type, comp_type, out_type;
  unsigned int flags;
  int type;        /* libxsmm_meltw_{unary,binary,ternary}_type */
  int operation;   /* libxsmm_meltw_operation */
} oracle_meltw_desc;

/* [ref: src/generator_mateltwise_reference_impl.c:2074 / 2505 / 2596 / 2663] */
/* DROPOUT: rows per draw of the generator = the 32-bit vector length of the CPU the reference runs on (default 16) */
void oracle_set_rng_width(int w);
void oracle_meltw_unary(const libxsmm_meltw_unary_param* param, const oracle_meltw_desc* desc);
void oracle_meltw_binary(const libxsmm_meltw_binary_param* param, const oracle_meltw_desc* desc);
void oracle_meltw_ternary(const libxsmm_meltw_ternary_param* param, const oracle_meltw_desc* desc);

/* ---- packed / sparse kernels (gold loops of the reference's drivers) -------------------
 * dtype is LIBXSMM_DATATYPE_F32 or _F64; beta0 != 0 means LIBXSMM_GEMM_FLAG_BETA_0.        */
/* C[m][n][p] (+)= sum_nz A.val[nz] * B[col[nz]][n][p]; rows without nz stay untouched.
 * [ref: samples/xgemm_norm_packed/asparse_packed_csr.c:113-130;
 *       src/generator_packed_spgemm_csr_asparse_avx_avx2_avx512.c:336-470] */
void oracle_packed_spgemm_csr_asparse(int dtype, int M, int N, int K, int P,
  const unsigned int* row_ptr, const unsigned int* col_idx, const void* a_vals,
  const void* B, int ldb, void* C, int ldc, int beta0);
/* C[m][n][p] (+)= sum_k A[m][k][p] * B_csc[k][n].
 * [ref: samples/xgemm_norm_packed/bsparse_packed_csc.c:133-150] */
void oracle_packed_spgemm_csc_bsparse(int dtype, int M, int N, int K, int P,
  const unsigned int* col_ptr, const unsigned int* row_idx, const void* b_vals,
  const void* A, int lda, void* C, int ldc, int beta0);
/* same with B in CSR (rows k). [ref: samples/xgemm_norm_packed/bsparse_packed_csr.c] */
void oracle_packed_spgemm_csr_bsparse(int dtype, int M, int N, int K, int P,
  const unsigned int* row_ptr, const unsigned int* col_idx, const void* b_vals,
  const void* A, int lda, void* C, int ldc, int beta0);
/* sparse C (ldc == 0) variant of the packed CSC kernel: the packed axis is reduced [ref: src/generator_packed_spgemm.c:81-94] */
void oracle_packed_spgemm_csc_csparse(int N, int K, int P, const unsigned int* col_ptr, const unsigned int* row_idx,
  const float* A, int lda, const float* B, int ldb, float* Cvals, int beta0);
/* Block-sparse B (BCSC), per M-block:  C[mb][n][m] = beta*C + sum_k A[mb][k][m]*B[k][n].
 * a_type/c_type in {F32, BF16}; bf16 A is VNNI-2 packed [K/2][M][2] when vnni_a != 0.
 * [ref: samples/xgemm_sparse/spmm_kernel.c:74-217 (gold), :219-375 (layouts)] */
void oracle_packed_spgemm_bcsc(int a_type, int c_type, int M, int N, int K, int m_blocks, int bk, int bn,
  int vnni_a, const void* A, const void* b_vals, const unsigned int* col_ptr, const unsigned int* row_idx,
  void* C, int beta0);
/* Row-major C[M x N] = A_csr * B + beta*C with alpha folded into the CSR values.
 * [ref: samples/xgemm_sparse_Ainregs/pyfr_driver_asp_reg.c:351-375; src/libxsmm_fsspmdm.c:196-236] */
void oracle_fsspmdm(int dtype, int M, int N, int K, const unsigned int* row_ptr, const unsigned int* col_idx,
  const void* a_vals, const void* B, int ldb, void* C, int ldc, int beta0);

/* Dense packed GEMMs [ref: samples/xgemm_norm_packed/dense_packedacrm.c:20-58, dense_packedbcrm.c:20-58,
 * samples/xgemm_packed/gemm_packed_kernel.c:35-72].  ac_rm: C[m][n][p] (+)= sum_k A[m][k][p] * B[k][n];
 * bc_rm: C[m][n][p] (+)= sum_k A[m][k] * B[k][n][p]; packed: C[n][m][p] (+)= sum_k A[k][m][p] * B[n][k][p]. */
void oracle_packed_gemm_ac_rm(int dtype, int M, int N, int K, int P, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int beta0);
void oracle_packed_gemm_bc_rm(int dtype, int M, int N, int K, int P, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int beta0);
void oracle_packed_gemm(int dtype, int M, int N, int K, int P, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int beta0);

/* ---- low-precision conversions  [ref: src/libxsmm_math.c:640-704] --------------------- */
unsigned short oracle_f32_to_bf16_rne(float x);   /* RNE with DAZ and NaN quieting */
unsigned short oracle_f32_to_bf16_trunc(float x);
float oracle_bf16_to_f32(unsigned short x);
float oracle_bf8_to_f32(unsigned char x);    /* E5M2  [ref: src/libxsmm_math.c:546-551] */
float oracle_hf8_to_f32(unsigned char x);    /* E4M3  [ref: src/libxsmm_math.c:553-585] */
float oracle_f16_to_f32(unsigned short h);
unsigned short oracle_f32_to_f16(float x);
unsigned char oracle_f32_to_bf8_rne(float x);
unsigned char oracle_f16_to_hf8_rne(unsigned short h);
unsigned char oracle_f32_to_hf8_rne(float x);

/* ---- comparison metric  [ref: src/libxsmm_matdiff.h:141-142 normf_rel] ---------------- */
double oracle_normf_rel(int dtype, long long count, const void* ref, const void* tst);

#if defined(__cplusplus)
}
#endif
#endif /* ORACLE_H */
