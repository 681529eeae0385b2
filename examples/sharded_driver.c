/* sharded_driver.c -- the multi-GPU path of SURVEY 8(e) from a plain C host: ONE process, ONE thread, no launcher, no Python.
 * The reference scales out through the caller's loop over independent problems [ref: samples/xgemm/gemm_kernel.c:4063-4066]; here the
 * batch axis is cut into one contiguous block per shard (libxsmm_hip_shard_range), every shard's A / B / C live on the shard's device, one
 * call launches all shards (libxsmm_hip_gemm[_ext]_batch_strided_sharded) and gathers C onto device 0 -- each source over its own link.
 *
 *   sharded_driver M BATCH NSHARDS [f32|bf16fused] [REPS]
 *   sharded_driver 35 P NSHARDS csr  [REPS]      packed CSR (A sparse, 35 x 35 @15 %, N = 9): the packed width P split (round 6: libxsmm_hip_create_packed_spgemm_csr_sharded)
 *   sharded_driver 64 MB NSHARDS bcsc [REPS]     bf16 BCSC 2:8 (64 x 256 x 64, bk = 32, bn = 16): the MB M-blocks split (libxsmm_hip_create_packed_spgemm_bcsc_sharded)
 *
 * Shard s runs on device s % device_count: on a one-GPU box the shards are VIRTUAL (one device, a stream and scratch of its own each), which is
 * what the parity test uses.  Gold: the whole batch in ONE unsharded launch on device 0; the gathered C must equal it bit for bit.
 * bf16fused = BASELINE config #5's kernel: bf16 VNNI-2 A, column bias + ReLU through libxsmm_dispatch_brgemm_ext.
 * Prints one JSON line.
 */
#include <libxsmm.h>
#include <libxsmm_hip.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_SHARDS 64

static void fill(void* host, size_t elems, int bf16) {
  size_t i;
  for (i = 0; i < elems; ++i) {                                           /* multiples of 0.1 like the reference's drivers; bf16 by truncation */
    const float v = (float)((int)(libxsmm_rng_f64() * 10.0) - 4) / 10.0f;
    if (bf16) { unsigned int bits; memcpy(&bits, &v, 4); ((unsigned short*)host)[i] = (unsigned short)(bits >> 16); }
    else ((float*)host)[i] = v;
  }
}

/* created (sparse) kernels: the creator's arguments go to the *_sharded creator, which builds one kernel per shard on the shard's device */
static int run_sparse(int bcsc, size_t axis, int nshards, int reps) {
  const int ndev = libxsmm_hip_device_count();
  enum { M = 35, K = 35, N = 9, BM = 64, BK = 256, BN = 64, bk = 32, bn = 16 };
  const size_t es = bcsc ? 2 : 4;
  unsigned int ptr[65], idx[35 * 35 + 16 * 8];
  float vals[35 * 35];
  unsigned int nnz = 0;
  libxsmm_hip_sharded_kernel* set;
  libxsmm_gemmfunction gold_kernel;
  libxsmm_gemm_param gp, p[MAX_SHARDS];
  unsigned long long nblk = BN / bn;
  size_t x_elems, c_elems, b_elems = 0, i;
  char *hx, *hb = NULL, *hgold, *hgot, *dx, *dv, *dgold, *dgot, *dcp = NULL, *dri = NULL;
  char *sx[MAX_SHARDS], *sv[MAX_SHARDS], *sc[MAX_SHARDS], *scp[MAX_SHARDS], *sri[MAX_SHARDS];
  int devices[MAX_SHARDS], n, s, r, rc = EXIT_SUCCESS, same;
  libxsmm_timer_tickint t0, t1;
  libxsmm_gemm_shape shape;
  libxsmm_spgemm_config cfg;
  libxsmm_rng_set_seed(777);
  if (bcsc) {                                                             /* 2 of every 8 K-blocks per block column */
    unsigned int nb, g;
    ptr[0] = 0;
    for (nb = 0; nb < BN / bn; ++nb) { for (g = 0; g < BK / bk; g += 8) { unsigned int a = g + nb % 8, b = g + (nb + 3) % 8; if (a > b) { const unsigned int t = a; a = b; b = t; } idx[nnz++] = a; if (b != a) idx[nnz++] = b; } ptr[nb + 1] = nnz; }
    shape = libxsmm_create_gemm_shape((libxsmm_blasint)axis, 0, BK, BK, 0, BN, LIBXSMM_DATATYPE_BF16, LIBXSMM_DATATYPE_BF16, LIBXSMM_DATATYPE_BF16, LIBXSMM_DATATYPE_F32);
    cfg.packed_width = BM; cfg.bk = bk; cfg.bn = bn;
    x_elems = axis * BK * BM; c_elems = axis * BN * BM; b_elems = (size_t)nnz * bn * bk;
  }
  else {
    int row, col;
    for (row = 0; row < M; ++row) { ptr[row] = nnz; for (col = 0; col < K; ++col) if (libxsmm_rng_f64() < 0.15) { idx[nnz] = (unsigned int)col; vals[nnz++] = (float)(libxsmm_rng_f64() - 0.5); } }
    ptr[M] = nnz;
    shape = libxsmm_create_gemm_shape(M, N, K, 0, N, N, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
    x_elems = (size_t)K * N * axis; c_elems = (size_t)M * N * axis;
  }
  hx = (char*)malloc(x_elems * es); hgold = (char*)malloc(c_elems * es); hgot = (char*)malloc(c_elems * es);
  if (bcsc) { hb = (char*)malloc(b_elems * es); fill(hb, b_elems, 1); }
  if (!hx || !hgold || !hgot || (bcsc && !hb)) return 3;
  fill(hx, x_elems, bcsc);
  for (s = 0; s < nshards; ++s) devices[s] = s % ndev;
  /* gold: the unsharded kernel on device 0 */
  libxsmm_hip_set_device(0);
  dx = (char*)libxsmm_hip_malloc(x_elems * es); dgold = (char*)libxsmm_hip_malloc(c_elems * es); dgot = (char*)libxsmm_hip_malloc(c_elems * es);
  dv = (char*)libxsmm_hip_malloc(bcsc ? b_elems * es : sizeof(vals));
  if (!dx || !dgold || !dgot || !dv) return 3;
  libxsmm_hip_memcpy_h2d(dx, hx, x_elems * es); libxsmm_hip_memcpy_h2d(dv, bcsc ? (const void*)hb : (const void*)vals, bcsc ? b_elems * es : sizeof(vals));
  libxsmm_hip_memset(dgold, 0xef, c_elems * es); libxsmm_hip_memset(dgot, 0xef, c_elems * es);
  memset(&gp, 0, sizeof(gp));
  if (bcsc) {
    dcp = (char*)libxsmm_hip_malloc(sizeof(ptr)); dri = (char*)libxsmm_hip_malloc(sizeof(idx));
    libxsmm_hip_memcpy_h2d(dcp, ptr, sizeof(ptr)); libxsmm_hip_memcpy_h2d(dri, idx, sizeof(idx));
    gold_kernel = libxsmm_create_packed_spgemm_bcsc(shape, LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_VNNI_A, LIBXSMM_GEMM_PREFETCH_NONE, cfg);
    gp.a.primary = dx; gp.b.primary = dv; gp.b.secondary = dcp; gp.b.tertiary = dri; gp.b.quaternary = &nblk; gp.c.primary = dgold;
  }
  else {
    gold_kernel = libxsmm_create_packed_spgemm_csr(shape, LIBXSMM_GEMM_FLAG_BETA_0, LIBXSMM_GEMM_PREFETCH_NONE, (libxsmm_blasint)axis, ptr, idx, vals);
    gp.a.primary = dv; gp.b.primary = dx; gp.c.primary = dgold;
  }
  if (NULL == gold_kernel) { fprintf(stderr, "the creator returned NULL\n"); return 2; }
  gold_kernel(&gp);
  libxsmm_hip_sync();
  /* the sharded set: same arguments, plus the shard count and the devices */
  set = bcsc ? libxsmm_hip_create_packed_spgemm_bcsc_sharded(shape, LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_VNNI_A, LIBXSMM_GEMM_PREFETCH_NONE, cfg, nshards, devices)
             : libxsmm_hip_create_packed_spgemm_csr_sharded(shape, LIBXSMM_GEMM_FLAG_BETA_0, LIBXSMM_GEMM_PREFETCH_NONE, (libxsmm_blasint)axis, ptr, idx, vals, nshards, devices);
  if (NULL == set) { fprintf(stderr, "the sharded creator returned NULL\n"); return 2; }
  n = libxsmm_hip_sharded_count(set);
  for (s = 0; s < n; ++s) {                                               /* every device holds only its own block, in the shard's own compact layout */
    int dev; size_t b, e, w, row;
    libxsmm_hip_sharded_range(set, s, &dev, &b, &e);
    w = e - b;
    libxsmm_hip_set_device(dev);
    memset(&p[s], 0, sizeof(p[s]));
    if (bcsc) {
      sx[s] = (char*)libxsmm_hip_malloc(w * BK * BM * es); sc[s] = (char*)libxsmm_hip_malloc(w * BN * BM * es); sv[s] = (char*)libxsmm_hip_malloc(b_elems * es);
      scp[s] = (char*)libxsmm_hip_malloc(sizeof(ptr)); sri[s] = (char*)libxsmm_hip_malloc(sizeof(idx));
      if (!sx[s] || !sc[s] || !sv[s] || !scp[s] || !sri[s]) return 3;
      libxsmm_hip_memcpy_h2d(sx[s], hx + b * BK * BM * es, w * BK * BM * es); libxsmm_hip_memcpy_h2d(sv[s], hb, b_elems * es);
      libxsmm_hip_memcpy_h2d(scp[s], ptr, sizeof(ptr)); libxsmm_hip_memcpy_h2d(sri[s], idx, sizeof(idx));
      p[s].a.primary = sx[s]; p[s].b.primary = sv[s]; p[s].b.secondary = scp[s]; p[s].b.tertiary = sri[s]; p[s].b.quaternary = &nblk; p[s].c.primary = sc[s];
    }
    else {                                                                /* B [K][N][P] -> the shard's [K][N][P_s] */
      char* tmp = (char*)malloc((size_t)K * N * w * es);
      if (!tmp) return 3;
      for (row = 0; row < (size_t)K * N; ++row) memcpy(tmp + row * w * es, hx + (row * axis + b) * es, w * es);
      sx[s] = (char*)libxsmm_hip_malloc((size_t)K * N * w * es); sc[s] = (char*)libxsmm_hip_malloc((size_t)M * N * w * es); sv[s] = (char*)libxsmm_hip_malloc(sizeof(vals));
      scp[s] = sri[s] = NULL;
      if (!sx[s] || !sc[s] || !sv[s]) return 3;
      libxsmm_hip_memcpy_h2d(sx[s], tmp, (size_t)K * N * w * es); libxsmm_hip_memcpy_h2d(sv[s], vals, sizeof(vals));
      libxsmm_hip_memset(sc[s], 0xef, (size_t)M * N * w * es);            /* rows of A without a non-zero leave their rows of C untouched, here as in the gold run */
      free(tmp);
      p[s].a.primary = sv[s]; p[s].b.primary = sx[s]; p[s].c.primary = sc[s];
    }
  }
  libxsmm_hip_set_device(0);
  t0 = libxsmm_timer_tick();
  for (r = 0; r < reps + 1 && EXIT_SUCCESS == rc; ++r) {
    if (1 == r) t0 = libxsmm_timer_tick();
    rc = libxsmm_hip_sharded_launch(set, p, 0, dgot, bcsc ? 0 : axis * es);        /* CSR: every shard's columns land in place inside [M][N][P] */
  }
  t1 = libxsmm_timer_tick();
  libxsmm_hip_memcpy_d2h(hgold, dgold, c_elems * es); libxsmm_hip_memcpy_d2h(hgot, dgot, c_elems * es);
  same = 0 == memcmp(hgold, hgot, c_elems * es);
  if (!same) { for (i = 0; i < c_elems * es && hgold[i] == hgot[i]; ++i) {} fprintf(stderr, "first differing byte: %lu of %lu (gold %02x %02x %02x %02x, sharded %02x %02x %02x %02x)\n", (unsigned long)i, (unsigned long)(c_elems * es),
      (unsigned char)hgold[i & ~(size_t)3], (unsigned char)hgold[(i & ~(size_t)3) + 1], (unsigned char)hgold[(i & ~(size_t)3) + 2], (unsigned char)hgold[(i & ~(size_t)3) + 3],
      (unsigned char)hgot[i & ~(size_t)3], (unsigned char)hgot[(i & ~(size_t)3) + 1], (unsigned char)hgot[(i & ~(size_t)3) + 2], (unsigned char)hgot[(i & ~(size_t)3) + 3]); }
  printf("{\"kernel\": \"%s\", \"axis\": %lu, \"shards\": %d, \"non_empty_shards\": %d, \"devices\": %d, \"reps\": %d, \"ms_per_sharded_launch_with_gather\": %.4f, "
         "\"bit_identical\": %s, \"rc\": %d, \"error\": %d, \"error_string\": \"%s\"}\n",
         bcsc ? "bcsc" : "csr", (unsigned long)axis, nshards, n, ndev, reps, libxsmm_timer_duration(t0, t1) * 1e3 / reps, same ? "true" : "false", rc,
         libxsmm_hip_get_last_error(), libxsmm_hip_get_last_error_string());
  for (s = 0; s < n; ++s) {
    int dev; libxsmm_hip_sharded_range(set, s, &dev, NULL, NULL); libxsmm_hip_set_device(dev);
    libxsmm_hip_free(sx[s]); libxsmm_hip_free(sc[s]); libxsmm_hip_free(sv[s]); if (scp[s]) libxsmm_hip_free(scp[s]); if (sri[s]) libxsmm_hip_free(sri[s]);
  }
  libxsmm_hip_set_device(0);
  libxsmm_hip_sharded_destroy(set);
  libxsmm_release_kernel((const void*)gold_kernel);
  libxsmm_hip_free(dx); libxsmm_hip_free(dv); libxsmm_hip_free(dgold); libxsmm_hip_free(dgot); if (dcp) libxsmm_hip_free(dcp); if (dri) libxsmm_hip_free(dri);
  free(hx); free(hgold); free(hgot); free(hb);
  return (same && EXIT_SUCCESS == rc && 0 == libxsmm_hip_get_last_error()) ? 0 : 1;
}

int main(int argc, char* argv[]) {
  const int m = argc > 1 ? atoi(argv[1]) : 32;
  const size_t batch = argc > 2 ? (size_t)atol(argv[2]) : 4096;
  const int nshards = argc > 3 ? atoi(argv[3]) : 2;
  const int fused = argc > 4 && 0 == strcmp(argv[4], "bf16fused");
  const int reps = argc > 5 ? atoi(argv[5]) : 3;
  const int ndev = libxsmm_hip_device_count();
  const size_t es = fused ? 2 : 4, blk = (size_t)m * m * es, total = blk * batch;
  const libxsmm_datatype dt = fused ? LIBXSMM_DATATYPE_BF16 : LIBXSMM_DATATYPE_F32;
  const libxsmm_gemm_shape shape = libxsmm_create_gemm_shape(m, m, m, m, m, m, dt, dt, dt, LIBXSMM_DATATYPE_F32);
  libxsmm_gemmfunction kernel = NULL; libxsmm_gemmfunction_ext kernel_ext = NULL;
  char *ha, *hb, *hd, *hgold, *hgot, *da0, *db0, *dd0, *dgold, *dgot;
  char *sa[MAX_SHARDS], *sb[MAX_SHARDS], *sc[MAX_SHARDS], *sd[MAX_SHARDS];
  libxsmm_gemm_param p[MAX_SHARDS]; libxsmm_gemm_ext_param pe[MAX_SHARDS];
  int devices[MAX_SHARDS];
  unsigned long long br = 1;
  libxsmm_timer_tickint t0, t1;
  double seconds;
  int s, r, rc = EXIT_SUCCESS, same;
  if (ndev <= 0) { fprintf(stderr, "no HIP device\n"); return 2; }
  if (nshards < 1 || nshards > MAX_SHARDS || (fused && 0 != (m % 2))) return 2;
  if (argc > 4 && (0 == strcmp(argv[4], "csr") || 0 == strcmp(argv[4], "bcsc"))) return run_sparse(0 == strcmp(argv[4], "bcsc"), batch, nshards, reps);
  if (fused) {
    const libxsmm_gemm_batch_reduce_config brc = libxsmm_create_gemm_batch_reduce_config(LIBXSMM_GEMM_BATCH_REDUCE_STRIDE, (libxsmm_blasint)blk, (libxsmm_blasint)blk, 0);
    const libxsmm_gemm_ext_unary_argops argops = libxsmm_create_gemm_ext_unary_argops(0, LIBXSMM_MELTW_TYPE_UNARY_NONE, LIBXSMM_MELTW_FLAG_UNARY_NONE, 0,
      0, LIBXSMM_MELTW_TYPE_UNARY_NONE, LIBXSMM_MELTW_FLAG_UNARY_NONE, 0, m, LIBXSMM_MELTW_TYPE_UNARY_RELU, LIBXSMM_MELTW_FLAG_UNARY_NONE, 0);
    const libxsmm_gemm_ext_binary_postops postops = libxsmm_create_gemm_ext_binary_postops(m, LIBXSMM_DATATYPE_BF16, LIBXSMM_MELTW_TYPE_BINARY_ADD, LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0);
    kernel_ext = libxsmm_dispatch_brgemm_ext(shape, LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_VNNI_A, LIBXSMM_GEMM_PREFETCH_NONE, brc, argops, postops);
  }
  else kernel = libxsmm_dispatch_gemm(shape, LIBXSMM_GEMM_FLAG_BETA_0, LIBXSMM_GEMM_PREFETCH_NONE);
  if (NULL == kernel && NULL == kernel_ext) { fprintf(stderr, "dispatch returned NULL\n"); return 2; }
  ha = (char*)malloc(total); hb = (char*)malloc(total); hd = (char*)malloc((size_t)m * es); hgold = (char*)malloc(total); hgot = (char*)malloc(total);
  if (!ha || !hb || !hd || !hgold || !hgot) return 3;
  libxsmm_rng_set_seed(555);
  fill(ha, total / es, fused); fill(hb, total / es, fused); fill(hd, (size_t)m, fused);
  /* gold: the whole batch, one launch, device 0 */
  libxsmm_hip_set_device(0);
  da0 = (char*)libxsmm_hip_malloc(total); db0 = (char*)libxsmm_hip_malloc(total); dd0 = (char*)libxsmm_hip_malloc((size_t)m * es);
  dgold = (char*)libxsmm_hip_malloc(total); dgot = (char*)libxsmm_hip_malloc(total);
  if (!da0 || !db0 || !dd0 || !dgold || !dgot) return 3;
  libxsmm_hip_memcpy_h2d(da0, ha, total); libxsmm_hip_memcpy_h2d(db0, hb, total); libxsmm_hip_memcpy_h2d(dd0, hd, (size_t)m * es);
  libxsmm_hip_memset(dgold, 0xef, total); libxsmm_hip_memset(dgot, 0xef, total);
  if (fused) {
    memset(&pe[0], 0, sizeof(pe[0]));
    pe[0].op.tertiary = &br; pe[0].a.primary = da0; pe[0].b.primary = db0; pe[0].c.primary = dgold; pe[0].d.primary = dd0;
    libxsmm_hip_gemm_ext_batch_strided(kernel_ext, &pe[0], batch, (long long)blk, (long long)blk, (long long)blk, 0, 0);
  }
  else {
    memset(&p[0], 0, sizeof(p[0]));
    p[0].a.primary = da0; p[0].b.primary = db0; p[0].c.primary = dgold;
    libxsmm_hip_gemm_batch_strided(kernel, &p[0], batch, (long long)blk, (long long)blk, (long long)blk);
  }
  libxsmm_hip_sync();
  /* the shards: every device holds ONLY its own block of A / B / C (and a replica of the shared bias) */
  for (s = 0; s < nshards; ++s) {
    size_t b, e, n;
    libxsmm_hip_shard_range(batch, 1, nshards, s, &b, &e);
    n = (e - b) * blk;
    devices[s] = s % ndev;
    libxsmm_hip_set_device(devices[s]);
    sa[s] = (char*)libxsmm_hip_malloc(n ? n : 1); sb[s] = (char*)libxsmm_hip_malloc(n ? n : 1); sc[s] = (char*)libxsmm_hip_malloc(n ? n : 1);
    sd[s] = (char*)libxsmm_hip_malloc((size_t)m * es);
    if (!sa[s] || !sb[s] || !sc[s] || !sd[s]) return 3;
    if (n) { libxsmm_hip_memcpy_h2d(sa[s], ha + b * blk, n); libxsmm_hip_memcpy_h2d(sb[s], hb + b * blk, n); libxsmm_hip_memset(sc[s], 0xef, n); }
    libxsmm_hip_memcpy_h2d(sd[s], hd, (size_t)m * es);
    memset(&p[s], 0, sizeof(p[s])); memset(&pe[s], 0, sizeof(pe[s]));
    p[s].a.primary = sa[s]; p[s].b.primary = sb[s]; p[s].c.primary = sc[s];
    pe[s].op.tertiary = &br; pe[s].a.primary = sa[s]; pe[s].b.primary = sb[s]; pe[s].c.primary = sc[s]; pe[s].d.primary = sd[s];
  }
  libxsmm_hip_set_device(0);
  (void)libxsmm_hip_launch_count(1);
  t0 = libxsmm_timer_tick();
  for (r = 0; r < reps + 1 && EXIT_SUCCESS == rc; ++r) {
    if (1 == r) t0 = libxsmm_timer_tick();                                /* rep 0 creates the shard streams: untimed */
    rc = fused ? libxsmm_hip_gemm_ext_batch_strided_sharded(kernel_ext, pe, batch, (long long)blk, (long long)blk, (long long)blk, 0, 0, nshards, devices, 0, dgot)
               : libxsmm_hip_gemm_batch_strided_sharded(kernel, p, batch, (long long)blk, (long long)blk, (long long)blk, nshards, devices, 0, dgot);
  }
  t1 = libxsmm_timer_tick();                                              /* blocking thread: every shard and every gather copy has finished */
  seconds = libxsmm_timer_duration(t0, t1);
  libxsmm_hip_memcpy_d2h(hgold, dgold, total); libxsmm_hip_memcpy_d2h(hgot, dgot, total);
  same = 0 == memcmp(hgold, hgot, total);
  printf("{\"kernel\": \"%s\", \"m\": %d, \"batch\": %lu, \"shards\": %d, \"devices\": %d, \"reps\": %d, \"ms_per_sharded_launch_with_gather\": %.4f, \"GFLOPs\": %.1f, "
         "\"launches_per_rep\": %.1f, \"bit_identical\": %s, \"rc\": %d, \"error\": %d, \"error_string\": \"%s\"}\n",
         fused ? "bf16fused" : "f32", m, (unsigned long)batch, nshards, ndev, reps, seconds * 1e3 / reps, 2.0 * m * m * m * (double)batch * reps / seconds * 1e-9,
         (double)libxsmm_hip_launch_count(0) / (reps + 1), same ? "true" : "false", rc, libxsmm_hip_get_last_error(), libxsmm_hip_get_last_error_string());
  for (s = 0; s < nshards; ++s) { libxsmm_hip_set_device(devices[s]); libxsmm_hip_free(sa[s]); libxsmm_hip_free(sb[s]); libxsmm_hip_free(sc[s]); libxsmm_hip_free(sd[s]); }
  libxsmm_hip_set_device(0);
  libxsmm_hip_free(da0); libxsmm_hip_free(db0); libxsmm_hip_free(dd0); libxsmm_hip_free(dgold); libxsmm_hip_free(dgot);
  free(ha); free(hb); free(hd); free(hgold); free(hgot);
  return (same && EXIT_SUCCESS == rc && 0 == libxsmm_hip_get_last_error()) ? 0 : 1;
}
