/*
 * oracle_sparse.c -- gold loops for the packed / sparse kernels (test-only).
 *
 * The reference has no C "reference implementation" for these kernels; what pins their
 * results are the gold loops inside its sample drivers.  Those are restated here:
 *   samples/xgemm_norm_packed/asparse_packed_csr.c:113-130   packed CSR, A sparse
 *   samples/xgemm_norm_packed/bsparse_packed_csc.c:133-150   packed CSC, B sparse
 *   samples/xgemm_norm_packed/bsparse_packed_csr.c           packed CSR, B sparse
 *   samples/xgemm_sparse/spmm_kernel.c:74-217                BCSC block-sparse B
 *   samples/xgemm_sparse_Ainregs/pyfr_driver_asp_reg.c:351-375   FsSpMDM
 * plus two behaviours only visible in the generators: BETA_0 zero-initialises the
 * accumulators, and an A-row without non-zeros leaves its C row untouched
 * (src/generator_packed_spgemm_csr_asparse_avx_avx2_avx512.c:347-357).
 *
 * The drivers' gold loops run over the *dense* K range (adding exact zeros); iterating
 * the stored non-zeros in ascending index order gives bit-identical sums, which is what
 * is done here.  Products are rounded before the add (compile with -ffp-contract=off).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define REAL_LOOP(T, BODY) do { typedef T real; BODY } while (0)

void oracle_packed_spgemm_csr_asparse(int dtype, int M, int N, int K, int P,
  const unsigned int* row_ptr, const unsigned int* col_idx, const void* a_vals,
  const void* B, int ldb, void* C, int ldc, int beta0)
{
  long long m, n, p; unsigned int z;
  (void)K;
#define BODY \
  const real* a = (const real*)a_vals; const real* b = (const real*)B; real* c = (real*)C; \
  for (m = 0; m < M; ++m) { \
    if (row_ptr[m + 1] == row_ptr[m]) continue;              /* empty row: C untouched */ \
    for (n = 0; n < N; ++n) for (p = 0; p < P; ++p) { \
      real acc = beta0 ? (real)0 : c[(m * ldc + n) * P + p]; \
      for (z = row_ptr[m]; z < row_ptr[m + 1]; ++z) { \
        const real prod = a[z] * b[((long long)col_idx[z] * ldb + n) * P + p]; \
        acc = acc + prod; \
      } \
      c[(m * ldc + n) * P + p] = acc; \
    } \
  }
  if (dtype == LIBXSMM_DATATYPE_F64) REAL_LOOP(double, BODY); else REAL_LOOP(float, BODY);
#undef BODY
}

void oracle_packed_spgemm_csc_bsparse(int dtype, int M, int N, int K, int P,
  const unsigned int* col_ptr, const unsigned int* row_idx, const void* b_vals,
  const void* A, int lda, void* C, int ldc, int beta0)
{
  long long m, n, p; unsigned int z;
  (void)K;
#define BODY \
  const real* a = (const real*)A; const real* b = (const real*)b_vals; real* c = (real*)C; \
  for (m = 0; m < M; ++m) for (n = 0; n < N; ++n) { \
    if (col_ptr[n + 1] == col_ptr[n] && !beta0) continue; \
    for (p = 0; p < P; ++p) { \
      real acc = beta0 ? (real)0 : c[(m * ldc + n) * P + p]; \
      for (z = col_ptr[n]; z < col_ptr[n + 1]; ++z) { \
        const real prod = a[(m * lda + (long long)row_idx[z]) * P + p] * b[z]; \
        acc = acc + prod; \
      } \
      c[(m * ldc + n) * P + p] = acc; \
    } \
  }
  if (dtype == LIBXSMM_DATATYPE_F64) REAL_LOOP(double, BODY); else REAL_LOOP(float, BODY);
#undef BODY
}

void oracle_packed_spgemm_csr_bsparse(int dtype, int M, int N, int K, int P,
  const unsigned int* row_ptr, const unsigned int* col_idx, const void* b_vals,
  const void* A, int lda, void* C, int ldc, int beta0)
{
  /* B given by rows k: for a fixed output column n the contributions arrive in ascending k,
   * the same order the dense gold loop uses. */
  long long m, n, p, k; unsigned int z;
#define BODY \
  const real* a = (const real*)A; const real* b = (const real*)b_vals; real* c = (real*)C; \
  for (m = 0; m < M; ++m) { \
    if (beta0) for (n = 0; n < N; ++n) for (p = 0; p < P; ++p) c[(m * ldc + n) * P + p] = (real)0; \
    for (k = 0; k < K; ++k) for (z = row_ptr[k]; z < row_ptr[k + 1]; ++z) { \
      n = col_idx[z]; \
      for (p = 0; p < P; ++p) { \
        const real prod = a[(m * lda + k) * P + p] * b[z]; \
        c[(m * ldc + n) * P + p] = c[(m * ldc + n) * P + p] + prod; \
      } \
    } \
  }
  if (dtype == LIBXSMM_DATATYPE_F64) REAL_LOOP(double, BODY); else REAL_LOOP(float, BODY);
#undef BODY
}

void oracle_packed_spgemm_bcsc(int a_type, int c_type, int M, int N, int K, int m_blocks, int bk, int bn,
  int vnni_a, const void* A, const void* b_vals, const unsigned int* col_ptr, const unsigned int* row_idx,
  void* C, int beta0)
{
  /* The driver's gold is a dense GEMM on the sparsified B [spmm_kernel.c:88-109 (f32), :113-151 (bf16), :153-217 (8-bit integers)];
   * zero blocks contribute exact zeros, so walking the stored blocks of block-column nb in
   * ascending block-row order reproduces it.  B block layout: vals[blk][dn][dk], k fastest.
   * 8-bit integers: a_type U8 means unsigned A x signed B, a_type I8 means signed A x unsigned B (the two combinations the
   * reference accepts [spmm_kernel.c:851-856]); A is VNNI-4 packed [K/4][M][4] when vnni_a [spmm_kernel.c:254-262]; C is int32. */
  long long mb, n, i; unsigned int blk; int dk;
  const int i8 = (a_type == LIBXSMM_DATATYPE_I8 || a_type == LIBXSMM_DATATYPE_U8);
  const int pack = !vnni_a ? 1 : (i8 ? 4 : (a_type == LIBXSMM_DATATYPE_BF16 ? 2 : 1));
  for (mb = 0; mb < m_blocks; ++mb) {
    for (n = 0; n < N; ++n) {
      const long long nb = n / bn, dn = n % bn;
      for (i = 0; i < M; ++i) {
        float acc = 0.0f; int iacc = 0;
        const long long cidx = mb * (long long)N * M + n * M + i;
        if (!beta0) {
          if (i8) iacc = ((const int*)C)[cidx];
          else acc = (c_type == LIBXSMM_DATATYPE_F32) ? ((const float*)C)[cidx] : oracle_bf16_to_f32(((const unsigned short*)C)[cidx]);
        }
        for (blk = col_ptr[nb]; blk < col_ptr[nb + 1]; ++blk) {
          const long long k0 = (long long)row_idx[blk] * bk;
          for (dk = 0; dk < bk; ++dk) {
            const long long k = k0 + dk;
            /* A per block: [K][M] col-major, or VNNI [K/pack][M][pack]  [spmm_kernel.c:244-262] */
            const long long aidx = mb * (long long)K * M + ((pack > 1) ? ((k / pack) * (M * pack) + i * pack + (k % pack)) : (k * M + i));
            const long long bidx = (long long)blk * bn * bk + dn * bk + dk;
            float av, bv, prod;
            if (i8) {
              const int ai = (a_type == LIBXSMM_DATATYPE_U8) ? (int)((const unsigned char*)A)[aidx] : (int)((const signed char*)A)[aidx];
              const int bi = (a_type == LIBXSMM_DATATYPE_U8) ? (int)((const signed char*)b_vals)[bidx] : (int)((const unsigned char*)b_vals)[bidx];
              iacc += ai * bi;
              continue;
            }
            if (a_type == LIBXSMM_DATATYPE_F32) { av = ((const float*)A)[aidx]; bv = ((const float*)b_vals)[bidx]; }
            else { av = oracle_bf16_to_f32(((const unsigned short*)A)[aidx]); bv = oracle_bf16_to_f32(((const unsigned short*)b_vals)[bidx]); }
            prod = av * bv; acc = acc + prod;
          }
        }
        if (i8) ((int*)C)[cidx] = iacc;
        else if (c_type == LIBXSMM_DATATYPE_F32) ((float*)C)[cidx] = acc; else ((unsigned short*)C)[cidx] = oracle_f32_to_bf16_rne(acc);
      }
    }
  }
}

/* Packed CSC with a SPARSE C (ldc == 0): C_val[z] (+)= sum_k sum_p A[k][row[z]][p] * B[k][n][p] for every stored entry z of column n.
 * The reference has no gold loop for it; this restates what its generator emits
 * [src/generator_packed_spgemm_csc_csparse_avx_avx2_avx512.c:17-195: per entry an accumulator over (k, packed chunks), horizontal add,
 *  beta handling on the scalar] with the accumulation in (k, p) order; pinned against the reference's JIT kernel within f32 summation-order tolerance. */
void oracle_packed_spgemm_csc_csparse(int N, int K, int P, const unsigned int* col_ptr, const unsigned int* row_idx,
  const float* A, int lda, const float* B, int ldb, float* Cvals, int beta0)
{
  int n, k; long long p; unsigned int z;
  for (n = 0; n < N; ++n) for (z = col_ptr[n]; z < col_ptr[n + 1]; ++z) {
    double acc = 0.0;
    for (k = 0; k < K; ++k) for (p = 0; p < P; ++p)
      acc += (double)A[((long long)k * lda + row_idx[z]) * P + p] * (double)B[((long long)k * ldb + n) * P + p];
    Cvals[z] = beta0 ? (float)acc : (float)((double)Cvals[z] + acc);
  }
}

void oracle_fsspmdm(int dtype, int M, int N, int K, const unsigned int* row_ptr, const unsigned int* col_idx,
  const void* a_vals, const void* B, int ldb, void* C, int ldc, int beta0)
{
  long long i, j; unsigned int z;
  (void)K;
#define BODY \
  const real* a = (const real*)a_vals; const real* b = (const real*)B; real* c = (real*)C; \
  for (j = 0; j < N; ++j) for (i = 0; i < M; ++i) { \
    real acc = beta0 ? (real)0 : c[i * ldc + j]; \
    for (z = row_ptr[i]; z < row_ptr[i + 1]; ++z) { \
      const real prod = a[z] * b[(long long)col_idx[z] * ldb + j]; \
      acc = acc + prod; \
    } \
    c[i * ldc + j] = acc; \
  }
  if (dtype == LIBXSMM_DATATYPE_F64) REAL_LOOP(double, BODY); else REAL_LOOP(float, BODY);
#undef BODY
}

/* ---- dense packed GEMMs (SOA layouts, P = packed width fastest) --------------------------------
 * [ref: samples/xgemm_norm_packed/dense_packedacrm.c:20-58 (matMulFusedAC), dense_packedbcrm.c:20-58 (matMulFusedBC),
 *  samples/xgemm_packed/gemm_packed_kernel.c:35-72].  k is the outermost loop of the gold code; the order of the
 *  additions into one C element is therefore k = 0..K-1, which is what is restated here per element. */
void oracle_packed_gemm_ac_rm(int dtype, int M, int N, int K, int P, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int beta0)
{
  long long m, n, k, p;
#define BODY \
  const real* a = (const real*)A; const real* b = (const real*)B; real* c = (real*)C; \
  for (m = 0; m < M; ++m) for (n = 0; n < N; ++n) for (p = 0; p < P; ++p) { \
    real acc = beta0 ? (real)0 : c[(m * ldc + n) * P + p]; \
    for (k = 0; k < K; ++k) { const real prod = a[(m * lda + k) * P + p] * b[k * ldb + n]; acc = acc + prod; } \
    c[(m * ldc + n) * P + p] = acc; \
  }
  if (dtype == LIBXSMM_DATATYPE_F64) REAL_LOOP(double, BODY); else REAL_LOOP(float, BODY);
#undef BODY
}
void oracle_packed_gemm_bc_rm(int dtype, int M, int N, int K, int P, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int beta0)
{
  long long m, n, k, p;
#define BODY \
  const real* a = (const real*)A; const real* b = (const real*)B; real* c = (real*)C; \
  for (m = 0; m < M; ++m) for (n = 0; n < N; ++n) for (p = 0; p < P; ++p) { \
    real acc = beta0 ? (real)0 : c[(m * ldc + n) * P + p]; \
    for (k = 0; k < K; ++k) { const real prod = a[m * lda + k] * b[(k * ldb + n) * P + p]; acc = acc + prod; } \
    c[(m * ldc + n) * P + p] = acc; \
  }
  if (dtype == LIBXSMM_DATATYPE_F64) REAL_LOOP(double, BODY); else REAL_LOOP(float, BODY);
#undef BODY
}
/* all three packed, column-major: C[n][m][p] (+)= sum_k A[k][m][p] * B[n][k][p] */
void oracle_packed_gemm(int dtype, int M, int N, int K, int P, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int beta0)
{
  long long m, n, k, p;
#define BODY \
  const real* a = (const real*)A; const real* b = (const real*)B; real* c = (real*)C; \
  for (m = 0; m < M; ++m) for (n = 0; n < N; ++n) for (p = 0; p < P; ++p) { \
    real acc = beta0 ? (real)0 : c[(n * ldc + m) * P + p]; \
    for (k = 0; k < K; ++k) { const real prod = a[(k * lda + m) * P + p] * b[(n * ldb + k) * P + p]; acc = acc + prod; } \
    c[(n * ldc + m) * P + p] = acc; \
  }
  if (dtype == LIBXSMM_DATATYPE_F64) REAL_LOOP(double, BODY); else REAL_LOOP(float, BODY);
#undef BODY
}
