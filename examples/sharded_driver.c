/* sharded_driver.c -- the multi-GPU path of SURVEY 8(e) from a plain C host: ONE process, ONE thread, no launcher, no Python.
 * The reference scales out through the caller's loop over independent problems [ref: samples/xgemm/gemm_kernel.c:4063-4066]; here the
 * batch axis is cut into one contiguous block per shard (libxsmm_hip_shard_range), every shard's A / B / C live on the shard's device, one
 * call launches all shards (libxsmm_hip_gemm[_ext]_batch_strided_sharded) and gathers C onto device 0 -- each source over its own link.
 *
 *   sharded_driver M BATCH NSHARDS [f32|bf16fused] [REPS]
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
