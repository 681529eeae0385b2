/* loop_driver.c -- the reference's calling pattern, unmodified: ONE small GEMM per call, the batch loop in the caller
 * [ref: documentation/libxsmm_mm.md:95-107, samples/xgemm/gemm_kernel.c:3170-3262 (reps loop around a single call)], timed in the three
 * launch modes of libxsmm_amd (include/libxsmm_hip.h, libxsmm_hip_set_async):
 *
 *   loop_driver M BATCH MODE REPS [f32|f64]      MODE: sync | async | coalesce
 *
 * sync: every call blocks until C is valid (the reference's semantics, the default); async: stream-ordered, one launch per call;
 * coalesce: stream-ordered, consecutive calls through the handle are queued and leave as ONE batched launch at libxsmm_hip_sync().
 * Gold: the same problems through ONE explicit libxsmm_hip_gemm_batch_strided launch; the loop's C must equal it bit for bit.
 * Prints one JSON line: microseconds per call (wall clock around the loop + the final sync), GFLOP/s, launches, bit_identical.
 */
#include <libxsmm.h>
#include <libxsmm_hip.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char* argv[]) {
  const int m = argc > 1 ? atoi(argv[1]) : 32, batch = argc > 2 ? atoi(argv[2]) : 4096;
  const char* mode = argc > 3 ? argv[3] : "sync";
  const int reps = argc > 4 ? atoi(argv[4]) : 3;
  const int f64 = argc > 5 && 0 == strcmp(argv[5], "f64");
  const size_t es = f64 ? sizeof(double) : sizeof(float), blk = (size_t)m * m * es, total = blk * (size_t)batch;
  const libxsmm_datatype dt = f64 ? LIBXSMM_DATATYPE_F64 : LIBXSMM_DATATYPE_F32;
  const libxsmm_gemm_shape shape = libxsmm_create_gemm_shape(m, m, m, m, m, m, dt, dt, dt, dt);
  const libxsmm_gemmfunction kernel = libxsmm_dispatch_gemm(shape, LIBXSMM_GEMM_FLAG_BETA_0, LIBXSMM_GEMM_PREFETCH_NONE);
  char *ha, *hb, *hc, *hg, *da, *db, *dc, *dg;
  libxsmm_gemm_param p;
  libxsmm_timer_tickint t0, t1;
  unsigned long long launches;
  double seconds;
  size_t i;
  int r, same;
  if (NULL == kernel) { fprintf(stderr, "dispatch returned NULL\n"); return 2; }
  ha = (char*)malloc(total); hb = (char*)malloc(total); hc = (char*)malloc(total); hg = (char*)malloc(total);
  da = (char*)libxsmm_hip_malloc(total); db = (char*)libxsmm_hip_malloc(total); dc = (char*)libxsmm_hip_malloc(total); dg = (char*)libxsmm_hip_malloc(total);
  if (!ha || !hb || !hc || !hg || !da || !db || !dc || !dg) return 3;
  libxsmm_rng_set_seed(555);
  for (i = 0; i < total / es; ++i) {                                     /* multiples of 0.1 like the reference's drivers */
    const double va = (double)((int)(libxsmm_rng_f64() * 10.0) - 4) / 10.0, vb = (double)((int)(libxsmm_rng_f64() * 10.0) - 4) / 10.0;
    if (f64) { ((double*)ha)[i] = va; ((double*)hb)[i] = vb; } else { ((float*)ha)[i] = (float)va; ((float*)hb)[i] = (float)vb; }
  }
  libxsmm_hip_memcpy_h2d(da, ha, total); libxsmm_hip_memcpy_h2d(db, hb, total);
  libxsmm_hip_memset(dc, 0xef, total); libxsmm_hip_memset(dg, 0xef, total);
  memset(&p, 0, sizeof(p));
  /* gold: one explicit batched launch */
  p.a.primary = da; p.b.primary = db; p.c.primary = dg;
  libxsmm_hip_gemm_batch_strided(kernel, &p, (size_t)batch, (long long)blk, (long long)blk, (long long)blk);
  libxsmm_hip_sync();
  libxsmm_hip_set_async(0 == strcmp(mode, "coalesce") ? 2 : (0 == strcmp(mode, "async") ? 1 : 0));
  for (i = 0; i < (size_t)batch; ++i) { p.a.primary = da + i * blk; p.b.primary = db + i * blk; p.c.primary = dc + i * blk; kernel(&p); }   /* warm-up */
  libxsmm_hip_sync();
  (void)libxsmm_hip_launch_count(1);
  t0 = libxsmm_timer_tick();
  for (r = 0; r < reps; ++r) {
    for (i = 0; i < (size_t)batch; ++i) {                                  /* the caller's loop, as written against the reference */
      p.a.primary = da + i * blk; p.b.primary = db + i * blk; p.c.primary = dc + i * blk;
      kernel(&p);
    }
    libxsmm_hip_sync();
  }
  t1 = libxsmm_timer_tick();
  seconds = libxsmm_timer_duration(t0, t1);
  launches = libxsmm_hip_launch_count(0);
  libxsmm_hip_memcpy_d2h(hc, dc, total); libxsmm_hip_memcpy_d2h(hg, dg, total);
  same = 0 == memcmp(hc, hg, total);
  printf("{\"mode\": \"%s\", \"dtype\": \"%s\", \"m\": %d, \"batch\": %d, \"reps\": %d, \"us_per_call\": %.4f, \"GFLOPs\": %.1f, \"launches_per_rep\": %.1f, \"bit_identical\": %s, \"error\": %d}\n",
         mode, f64 ? "f64" : "f32", m, batch, reps, seconds * 1e6 / ((double)reps * batch), 2.0 * m * m * m * batch * reps / seconds * 1e-9,
         (double)launches / reps, same ? "true" : "false", libxsmm_hip_get_last_error());
  libxsmm_hip_free(da); libxsmm_hip_free(db); libxsmm_hip_free(dc); libxsmm_hip_free(dg);
  free(ha); free(hb); free(hc); free(hg);
  return same && 0 == libxsmm_hip_get_last_error() ? 0 : 1;
}
