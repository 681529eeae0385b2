/* c_driver.c -- a plain C99 caller of the LIBXSMM dispatch/param API, linked against libxsmm_amd.so.
 * It is written the way a user of the reference writes its drivers (cf. the flow of samples/xgemm/gemm_kernel.c:
 * shape -> dispatch -> param struct -> call -> compare with a gold loop) and uses nothing GPU specific: operands come
 * from libxsmm_aligned_malloc (device-visible pinned memory in this library), kernel calls are synchronous.
 *
 *   c_driver probe               prints the device count; dispatch must return NULL without a GPU (exit 0 if so)
 *   c_driver gemm M N K BR       BASELINE config #1 style single (BR)GEMM f32, gold = triple loop, exit 0 if error < 1e-5
 *   c_driver spmm P              packed CSR A-sparse 35x35 @ 15 %, gold loop as in asparse_packed_csr.c, exit 0 if error < 1e-5
 */
#include <libxsmm.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float frand(void) { return (float)((int)(libxsmm_rng_f64() * 10.0) - 5) / 10.0f; }   /* multiples of 0.1 */

static int run_gemm(int m, int n, int k, int br) {
  const libxsmm_gemm_shape shape = libxsmm_create_gemm_shape(m, n, k, m, k, m,
    LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
  const libxsmm_gemm_batch_reduce_config brcfg = libxsmm_create_gemm_batch_reduce_config(
    br > 1 ? LIBXSMM_GEMM_BATCH_REDUCE_STRIDE : LIBXSMM_GEMM_BATCH_REDUCE_NONE, (libxsmm_blasint)(sizeof(float) * m * k), (libxsmm_blasint)(sizeof(float) * k * n), 0);
  const libxsmm_gemmfunction kernel = libxsmm_dispatch_brgemm(shape, LIBXSMM_GEMM_FLAG_NONE /* beta = 1 */, LIBXSMM_GEMM_PREFETCH_NONE, brcfg);
  float *a, *b, *c, *gold;
  unsigned long long brcount = (unsigned long long)br;
  libxsmm_gemm_param param;
  double err = 0.0, ref = 0.0;
  int i, j, s, r;
  if (NULL == kernel) { fprintf(stderr, "dispatch returned NULL\n"); return 2; }
  a = (float*)libxsmm_aligned_malloc(sizeof(float) * m * k * br, 64);
  b = (float*)libxsmm_aligned_malloc(sizeof(float) * k * n * br, 64);
  c = (float*)libxsmm_aligned_malloc(sizeof(float) * m * n, 64);
  gold = (float*)malloc(sizeof(float) * m * n);
  if (!a || !b || !c || !gold) return 3;
  libxsmm_rng_set_seed(555);
  for (i = 0; i < m * k * br; ++i) a[i] = frand();
  for (i = 0; i < k * n * br; ++i) b[i] = frand();
  for (i = 0; i < m * n; ++i) gold[i] = c[i] = frand();
  for (r = 0; r < br; ++r) for (j = 0; j < n; ++j) for (s = 0; s < k; ++s) for (i = 0; i < m; ++i)
    gold[i + j * m] += a[r * m * k + i + s * m] * b[r * k * n + s + j * k];
  memset(&param, 0, sizeof(param));
  param.a.primary = a; param.b.primary = b; param.c.primary = c; param.op.tertiary = &brcount;
  kernel(&param);                                   /* synchronous: c is valid on return */
  for (i = 0; i < m * n; ++i) { err += ((double)c[i] - gold[i]) * ((double)c[i] - gold[i]); ref += (double)gold[i] * gold[i]; }
  err = sqrt(err / (ref > 0 ? ref : 1));
  {
    libxsmm_mmkernel_info info; libxsmm_xmmfunction x; x.gemm = kernel;
    if (EXIT_SUCCESS == libxsmm_get_mmkernel_info(x, &info)) printf("kernel m=%u n=%u k=%u lda=%u ldb=%u ldc=%u\n", info.m, info.n, info.k, info.lda, info.ldb, info.ldc);
  }
  printf("gemm %dx%dx%d br=%d: normf_rel = %.3g (target %s)\n", m, n, k, br, err, libxsmm_get_target_arch());
  libxsmm_free(a); libxsmm_free(b); libxsmm_free(c); free(gold);
  return err < 1e-5 ? 0 : 1;
}

static int run_spmm(int P) {
  enum { M = 35, K = 35, N = 8 };
  unsigned int rowptr[M + 1], colidx[M * K];
  float vals[M * K];
  unsigned int nnz = 0;
  int m, k, n, p;
  float *b, *c, *gold, *dvals;
  double err = 0.0, ref = 0.0;
  libxsmm_gemmfunction kernel;
  libxsmm_gemm_param param;
  const libxsmm_gemm_shape shape = libxsmm_create_gemm_shape(M, N, K, 0, N, N,
    LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
  libxsmm_rng_set_seed(555);
  for (m = 0; m < M; ++m) { rowptr[m] = nnz; for (k = 0; k < K; ++k) if (libxsmm_rng_f64() < 0.15) { colidx[nnz] = (unsigned int)k; vals[nnz] = frand() + 0.05f; ++nnz; } }
  rowptr[M] = nnz;
  kernel = libxsmm_create_packed_spgemm_csr(shape, LIBXSMM_GEMM_FLAG_BETA_0, LIBXSMM_GEMM_PREFETCH_NONE, P, rowptr, colidx, vals);
  if (NULL == kernel) { fprintf(stderr, "create_packed_spgemm_csr returned NULL\n"); return 2; }
  b = (float*)libxsmm_aligned_malloc(sizeof(float) * K * N * P, 64);
  c = (float*)libxsmm_aligned_malloc(sizeof(float) * M * N * P, 64);
  dvals = (float*)libxsmm_aligned_malloc(sizeof(float) * (nnz ? nnz : 1), 64);
  gold = (float*)calloc((size_t)M * N * P, sizeof(float));
  if (!b || !c || !gold || !dvals) return 3;
  memcpy(dvals, vals, sizeof(float) * nnz);
  for (n = 0; n < K * N * P; ++n) b[n] = frand();
  memset(c, 0xef, sizeof(float) * M * N * P);
  for (m = 0; m < M; ++m) for (k = (int)rowptr[m]; k < (int)rowptr[m + 1]; ++k) for (n = 0; n < N; ++n) for (p = 0; p < P; ++p)
    gold[(m * N + n) * P + p] += vals[k] * b[((int)colidx[k] * N + n) * P + p];
  memset(&param, 0, sizeof(param));
  param.a.primary = dvals; param.b.primary = b; param.c.primary = c;
  kernel(&param);
  for (m = 0; m < M; ++m) if (rowptr[m] != rowptr[m + 1]) for (n = 0; n < N * P; ++n) {
    const double d = (double)c[m * N * P + n] - gold[m * N * P + n];
    err += d * d; ref += (double)gold[m * N * P + n] * gold[m * N * P + n];
  }
  err = sqrt(err / (ref > 0 ? ref : 1));
  printf("packed spgemm csr 35x35 nnz=%u P=%d: normf_rel = %.3g\n", nnz, P, err);
  libxsmm_release_kernel((const void*)kernel);
  libxsmm_free(b); libxsmm_free(c); libxsmm_free(dvals); free(gold);
  return err < 1e-5 ? 0 : 1;
}

int main(int argc, char* argv[]) {
  int rc = 0;
  libxsmm_init();
  if (argc < 2 || 0 == strcmp(argv[1], "probe")) {
    const libxsmm_gemm_shape shape = libxsmm_create_gemm_shape(23, 23, 23, 23, 23, 23,
      LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
    const libxsmm_gemmfunction kernel = libxsmm_dispatch_gemm(shape, LIBXSMM_GEMM_FLAG_NONE, LIBXSMM_GEMM_PREFETCH_NONE);
    const int ndev = libxsmm_hip_device_count();
    printf("devices=%d handle=%s\n", ndev, NULL != kernel ? "non-NULL" : "NULL");
    rc = ((ndev > 0) == (NULL != kernel)) ? 0 : 1;      /* no device -> NULL, never a host fallback */
  }
  else if (0 == strcmp(argv[1], "gemm") && argc >= 6) rc = run_gemm(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
  else if (0 == strcmp(argv[1], "spmm") && argc >= 3) rc = run_spmm(atoi(argv[2]));
  else { fprintf(stderr, "usage: %s probe | gemm M N K BR | spmm P\n", argv[0]); rc = 64; }
  libxsmm_finalize();
  return rc;
}
