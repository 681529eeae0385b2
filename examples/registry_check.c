/* registry_check.c -- exercises the dispatch registry the way a long-running application does:
 * many distinct descriptors, repeated hits, caller-owned handle recycling, init/finalize cycles.
 * Needs no GPU when run with LIBXSMM_HIP_DRYRUN=1 (dispatch works, calling a kernel is an error);
 * on a GPU box it runs as is.  Every sub-command prints "ok ..." and exits 0, or says what broke.
 *
 *   registry_check capacity <n_dispatch> <expected_ok>   distinct unary descriptors; the first expected_ok must give
 *                                                        distinct non-NULL handles, the rest NULL; a re-dispatch hits
 *   registry_check hit <reps>                            ns per libxsmm_dispatch_brgemm hit (thread-local cache)
 *   registry_check cycle                                 finalize invalidates the per-thread cache; re-dispatch is valid
 *   registry_check info                                  libxsmm_get_registry_info / kernel info / kernel names
 *   registry_check threads <threads> <n>                 every thread dispatches the same n descriptors, each in its own order, while the
 *                                                        others do: one handle per descriptor whoever got there first (the reference's
 *                                                        tests/threadsafety.c), registry size n, every handle describes its descriptor
 */
#include <libxsmm.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static libxsmm_meltwfunction_unary unary_of(int i) {
  /* i -> a distinct (m, n, ldi) triple; all of them legal RELU descriptors */
  const libxsmm_blasint m = 1 + (i % 512), n = 1 + (i / 512) % 512, ld = 512 + (i / (512 * 512));
  const libxsmm_meltw_unary_shape s = libxsmm_create_meltw_unary_shape(m, n, ld, ld, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
  return libxsmm_dispatch_meltw_unary(LIBXSMM_MELTW_TYPE_UNARY_RELU, s, LIBXSMM_MELTW_FLAG_UNARY_NONE);
}

static int cmp_ptr(const void* a, const void* b) {
  const size_t x = *(const size_t*)a, y = *(const size_t*)b; return x < y ? -1 : (x > y ? 1 : 0);
}

static int run_capacity(int n, int expected_ok) {
  size_t* h = (size_t*)malloc(sizeof(size_t) * (size_t)n);
  int i, ok = 0;
  double t0 = now_s(), t1;
  for (i = 0; i < n; ++i) { h[i] = (size_t)unary_of(i); if (h[i]) ++ok; }
  t1 = now_s();
  if (ok != expected_ok) { printf("FAIL: %d of %d dispatches succeeded, expected %d\n", ok, n, expected_ok); return 1; }
  for (i = 0; i < n; ++i) if ((h[i] != 0) != (i < expected_ok)) { printf("FAIL: dispatch %d is %s\n", i, h[i] ? "non-NULL" : "NULL"); return 1; }
  /* a hit on an early, a middle and the last registered descriptor returns the same handle */
  if ((size_t)unary_of(0) != h[0] || (size_t)unary_of(expected_ok / 2) != h[expected_ok / 2] || (size_t)unary_of(expected_ok - 1) != h[expected_ok - 1]) {
    printf("FAIL: re-dispatch does not return the registered handle\n"); return 1;
  }
  { libxsmm_registry_info info;
    if (libxsmm_get_registry_info(&info) != EXIT_SUCCESS || (int)info.size != expected_ok) { printf("FAIL: registry size %d\n", (int)info.size); return 1; }
    printf("registry capacity=%d size=%d\n", (int)info.capacity, (int)info.size);
  }
  qsort(h, (size_t)expected_ok, sizeof(size_t), cmp_ptr);
  for (i = 1; i < expected_ok; ++i) if (h[i] == h[i - 1]) { printf("FAIL: two descriptors share a handle\n"); return 1; }
  /* caller-owned handles never run out because of the registered ones being full is a separate budget; with the limit lowered
     through LIBXSMM_HIP_MAX_HANDLES both share it, which is what the n > expected_ok case of the test drives */
  printf("ok capacity: %d registered in %.3f s (%.0f ns per first dispatch)\n", expected_ok, t1 - t0, 1e9 * (t1 - t0) / n);
  free(h);
  return 0;
}

static int run_hit(long reps) {
  const libxsmm_gemm_shape s = libxsmm_create_gemm_shape(32, 32, 32, 32, 32, 32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
  const libxsmm_gemm_batch_reduce_config c = libxsmm_create_gemm_batch_reduce_config(LIBXSMM_GEMM_BATCH_REDUCE_STRIDE, 4096, 4096, 0);
  libxsmm_gemmfunction f0 = libxsmm_dispatch_brgemm(s, LIBXSMM_GEMM_FLAG_BETA_0, LIBXSMM_GEMM_PREFETCH_NONE, c), f = f0;
  long i; size_t acc = 0; double t0, t1;
  int j;
  if (!f0) { printf("FAIL: dispatch returned NULL\n"); return 1; }
  for (j = 0; j < 8; ++j) (void)unary_of(j);       /* other residents of the thread cache */
  t0 = now_s();
  for (i = 0; i < reps; ++i) { f = libxsmm_dispatch_brgemm(s, LIBXSMM_GEMM_FLAG_BETA_0, LIBXSMM_GEMM_PREFETCH_NONE, c); acc += (size_t)f; }
  t1 = now_s();
  if (f != f0 || acc != (size_t)f0 * (size_t)reps) { printf("FAIL: hits returned different handles\n"); return 1; }
  printf("ok hit: %.1f ns per dispatch hit\n", 1e9 * (t1 - t0) / (double)reps);
  return 0;
}

static int run_cycle(void) {
  libxsmm_meltwfunction_unary a = unary_of(7), b;
  libxsmm_kernel_info info;
  int i;
  if (!a) { printf("FAIL: dispatch returned NULL\n"); return 1; }
  for (i = 0; i < 3; ++i) {
    libxsmm_finalize();
    /* other kernels take the recycled slots first: a stale cache entry would now point at one of them */
    (void)unary_of(100 + i); (void)unary_of(200 + i);
    b = unary_of(7);
    if (!b) { printf("FAIL: dispatch after finalize returned NULL\n"); return 1; }
    if (libxsmm_get_kernel_info((const void*)b, &info) != EXIT_SUCCESS) { printf("FAIL: handle after finalize is not live\n"); return 1; }
    { libxsmm_xmeltwfunction x; libxsmm_meltwkernel_info mi; x.meltw_unary = b;
      if (libxsmm_get_meltwkernel_info(x, &mi) != EXIT_SUCCESS || mi.m != 8 || mi.n != 1) { printf("FAIL: handle after finalize describes another kernel (m=%u n=%u)\n", mi.m, mi.n); return 1; } }
  }
  printf("ok cycle\n");
  return 0;
}

static int run_info(void) {
  libxsmm_meltwfunction_unary u = unary_of(3);
  const libxsmm_meltw_binary_shape bs = libxsmm_create_meltw_binary_shape(8, 8, 8, 8, 8, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
  libxsmm_meltwfunction_binary b = libxsmm_dispatch_meltw_binary(LIBXSMM_MELTW_TYPE_BINARY_ADD, bs, LIBXSMM_MELTW_FLAG_BINARY_NONE);
  if (!u || !b) { printf("FAIL: dispatch returned NULL\n"); return 1; }
  printf("names: %s | %s\n", libxsmm_hip_kernel_name((const void*)u, 0), libxsmm_hip_kernel_name((const void*)b, 0));
  if (strcmp(libxsmm_hip_kernel_name((const void*)u, 0), libxsmm_hip_kernel_name((const void*)b, 0)) == 0) { printf("FAIL: TPP handles share one kernel name\n"); return 1; }
  libxsmm_release_kernel((const void*)u);      /* registered: a warned no-op */
  if ((const void*)unary_of(3) != (const void*)u) { printf("FAIL: release of a registered kernel removed it\n"); return 1; }
  printf("ok info\n");
  return 0;
}

typedef struct worker_t { int id, n, threads; size_t* handles; int failed; } worker_t;

static void* worker_main(void* arg) {
  worker_t* w = (worker_t*)arg;
  int r, i;
  for (r = 0; r < 3; ++r) {                  /* first pass registers (racing with the others), later passes hit */
    for (i = 0; i < w->n; ++i) {
      const int d = (int)(((long)i * (2 * w->id + 1) + 7919L * w->id + r) % w->n);      /* a thread-specific walk over all descriptors */
      const size_t h = (size_t)unary_of(d);
      if (!h) { w->failed = 1; return NULL; }
      if (w->handles[d] == 0) w->handles[d] = h; else if (w->handles[d] != h) { w->failed = 2; return NULL; }
    }
  }
  for (i = 0; i < w->n; ++i) if (w->handles[i] == 0) w->handles[i] = (size_t)unary_of(i);   /* walks with a common factor skip some: fill in */
  return NULL;
}

static int run_threads(int threads, int n) {
  pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  worker_t* w = (worker_t*)calloc((size_t)threads, sizeof(worker_t));
  int t, i;
  for (t = 0; t < threads; ++t) {
    w[t].id = t; w[t].n = n; w[t].threads = threads; w[t].handles = (size_t*)calloc((size_t)n, sizeof(size_t));
    if (pthread_create(&tid[t], NULL, worker_main, &w[t]) != 0) { printf("FAIL: pthread_create\n"); return 1; }
  }
  for (t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
  for (t = 0; t < threads; ++t) if (w[t].failed) { printf("FAIL: thread %d: %s\n", t, w[t].failed == 1 ? "NULL handle" : "two handles for one descriptor"); return 1; }
  for (i = 0; i < n; ++i) {
    libxsmm_xmeltwfunction x; libxsmm_meltwkernel_info mi;
    for (t = 1; t < threads; ++t) if (w[t].handles[i] != w[0].handles[i]) { printf("FAIL: descriptor %d has different handles in threads 0 and %d\n", i, t); return 1; }
    x.meltw_unary = (libxsmm_meltwfunction_unary)w[0].handles[i];
    if (libxsmm_get_meltwkernel_info(x, &mi) != EXIT_SUCCESS || (int)mi.m != 1 + (i % 512) || (int)mi.n != 1 + (i / 512) % 512) {
      printf("FAIL: handle of descriptor %d describes m=%u n=%u\n", i, mi.m, mi.n); return 1;
    }
  }
  { libxsmm_registry_info info;
    if (libxsmm_get_registry_info(&info) != EXIT_SUCCESS || (int)info.size != n) { printf("FAIL: registry size %d, expected %d\n", (int)info.size, n); return 1; } }
  printf("ok threads: %d threads x %d descriptors, one handle each\n", threads, n);
  return 0;
}

int main(int argc, char* argv[]) {
  libxsmm_init();
  if (argc >= 4 && strcmp(argv[1], "capacity") == 0) return run_capacity(atoi(argv[2]), atoi(argv[3]));
  if (argc >= 3 && strcmp(argv[1], "hit") == 0) return run_hit(atol(argv[2]));
  if (argc >= 2 && strcmp(argv[1], "cycle") == 0) return run_cycle();
  if (argc >= 2 && strcmp(argv[1], "info") == 0) return run_info();
  if (argc >= 4 && strcmp(argv[1], "threads") == 0) return run_threads(atoi(argv[2]), atoi(argv[3]));
  printf("usage: registry_check capacity <n> <expected_ok> | hit <reps> | cycle | info | threads <threads> <n>\n");
  return 2;
}
