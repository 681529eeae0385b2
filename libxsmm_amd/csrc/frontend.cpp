// frontend.cpp -- FsSpMDM (fixed-size sparse matrix x dense matrix) on top of the sparse panel
// kernel, plus the host-side helpers the reference's sample drivers link against (allocator,
// timer, RNG, bf16 conversions, matdiff).  The helpers run on the CPU by nature (they prepare and
// compare test data); no compute of the hot path lives here.
//
// FsSpMDM [ref: src/libxsmm_fsspmdm.c:24-560]: row-major C[MxN] = alpha*A*B + beta*C with a dense A
// that is sparsified at creation (alpha folded into the values), beta in {0,1}, N a multiple of the
// 64-byte "vector length" of the reference's AVX-512 host (kept so that the same inputs are accepted
// and rejected).  The reference builds up to three register-blocked sparse kernels plus a dense
// fallback and optionally auto-tunes between them; on the GPU there is one kernel -- lanes along N,
// the operator staged as scalars -- so the tuning knobs (timer_tick, LIBXSMM_FSSPMDM_HINT) are accepted
// and ignored.
#include <hip/hip_runtime_api.h>
#include "internal.hpp"
#include "lowp.hpp"

#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>

using namespace xamd;

struct libxsmm_fsspmdm {
  libxsmm_gemmfunction kernel;
  libxsmm_datatype datatype;
  int M, N, K, ldb, ldc;
};

template <typename T>
static void xgemm(libxsmm_datatype dt, const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const T* a, const libxsmm_blasint* lda, const T* b, const libxsmm_blasint* ldb, const T* beta, T* c, const libxsmm_blasint* ldc)
{
  if (!m) return;
  const bool ta = transa && (*transa == 'T' || *transa == 't' || *transa == 'C' || *transa == 'c');
  const bool tb = transb && (*transb == 'T' || *transb == 't' || *transb == 'C' || *transb == 'c');
  const T fbeta = beta ? *beta : (T)1;
  const libxsmm_blasint kk = k ? *k : *m, nn = n ? *n : kk;
  const libxsmm_blasint la = std::max<libxsmm_blasint>(lda ? *lda : (ta ? kk : *m), 1), lb = std::max<libxsmm_blasint>(ldb ? *ldb : (tb ? nn : kk), 1);
  const libxsmm_blasint lc = std::max<libxsmm_blasint>(ldc ? *ldc : *m, 1);
  const libxsmm_bitfield flags = (ta ? LIBXSMM_GEMM_FLAG_TRANS_A : 0) | (tb ? LIBXSMM_GEMM_FLAG_TRANS_B : 0) | (fbeta != (T)0 ? 0 : LIBXSMM_GEMM_FLAG_BETA_0);
  const libxsmm_gemmfunction f = libxsmm_dispatch_gemm(libxsmm_create_gemm_shape(*m, nn, kk, la, lb, lc, dt, dt, dt, dt), flags, LIBXSMM_GEMM_PREFETCH_NONE);
  if (!f) { std::printf("LIBXSMM_GEMM failed\n"); return; }
  libxsmm_gemm_param p; std::memset(&p, 0, sizeof(p));
  p.a.primary = const_cast<T*>(a); p.b.primary = const_cast<T*>(b); p.c.primary = c;
  f(&p);
}

extern "C" {

LIBXSMM_API libxsmm_fsspmdm* libxsmm_fsspmdm_create(libxsmm_datatype datatype, libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta, const void* a_dense, int c_is_nt,
  libxsmm_timer_tickint (*timer_tick)(void))
{
  (void)c_is_nt; (void)timer_tick;
  if (!a_dense || !runtime_ready()) return nullptr;                                          // [ref: fsspmdm.c:47-54]
  if (datatype != LIBXSMM_DATATYPE_F32 && datatype != LIBXSMM_DATATYPE_F64) return nullptr;
  const int typesz = (datatype == LIBXSMM_DATATYPE_F64) ? 8 : 4;
  const int vl = 64 / typesz;                                                                  // [ref: fsspmdm.c:58-62]
  const double fbeta = beta ? (datatype == LIBXSMM_DATATYPE_F64 ? *(const double*)beta : (double)*(const float*)beta) : 1.0;
  const double falpha = alpha ? (datatype == LIBXSMM_DATATYPE_F64 ? *(const double*)alpha : (double)*(const float*)alpha) : 1.0;
  if (M <= 0 || N <= 0 || K <= 0 || (N % vl) != 0 || !(fbeta == 1.0 || fbeta == 0.0) || lda < K || ldc < N || ldb < N) return nullptr;   // [ref: fsspmdm.c:83-85]
  // dense -> CSR with alpha folded in; entries that become exactly zero are dropped [ref: fsspmdm.c:196-236]
  std::vector<unsigned int> rowptr((size_t)M + 1, 0), colidx;
  std::vector<double> values;
  for (int i = 0; i < M; ++i) {
    rowptr[i] = (unsigned int)values.size();
    for (int j = 0; j < K; ++j) {
      double v;
      if (datatype == LIBXSMM_DATATYPE_F64) v = falpha * ((const double*)a_dense)[(size_t)i * lda + j];
      else v = (double)((float)falpha * ((const float*)a_dense)[(size_t)i * lda + j]);
      if (v != 0.0) { values.push_back(v); colidx.push_back((unsigned int)j); }
    }
  }
  rowptr[M] = (unsigned int)values.size();
  if (values.empty()) return nullptr;                                                          // empty matrix [ref: fsspmdm.c:133-141]
  const libxsmm_gemm_shape shape = libxsmm_create_gemm_shape(M, N, K, 0, ldb, ldc, datatype, datatype, datatype, datatype);
  const libxsmm_bitfield flags = (fbeta == 0.0) ? LIBXSMM_GEMM_FLAG_BETA_0 : 0;
  libxsmm_gemmfunction kernel = libxsmm_create_spgemm_csr_areg(shape, flags, LIBXSMM_GEMM_PREFETCH_NONE, N, rowptr.data(), colidx.data(), values.data());
  if (!kernel) return nullptr;
  libxsmm_fsspmdm* h = new libxsmm_fsspmdm();
  h->kernel = kernel; h->datatype = datatype; h->M = M; h->N = N; h->K = K; h->ldb = ldb; h->ldc = ldc;
  return h;
}
LIBXSMM_API libxsmm_dfsspmdm* libxsmm_dfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  double alpha, double beta, const double* a_dense, int c_is_nt, libxsmm_timer_tickint (*timer_tick)(void)) {
  return libxsmm_fsspmdm_create(LIBXSMM_DATATYPE_F64, M, N, K, lda, ldb, ldc, &alpha, &beta, a_dense, c_is_nt, timer_tick);
}
LIBXSMM_API libxsmm_sfsspmdm* libxsmm_sfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  float alpha, float beta, const float* a_dense, int c_is_nt, libxsmm_timer_tickint (*timer_tick)(void)) {
  return libxsmm_fsspmdm_create(LIBXSMM_DATATYPE_F32, M, N, K, lda, ldb, ldc, &alpha, &beta, a_dense, c_is_nt, timer_tick);
}
LIBXSMM_API void libxsmm_fsspmdm_execute(const libxsmm_fsspmdm* h, const void* B, void* C) {   // [ref: fsspmdm.c:491-514]
  if (!h) return;
  libxsmm_gemm_param p; std::memset(&p, 0, sizeof(p));
  p.b.primary = const_cast<void*>(B); p.c.primary = C;
  h->kernel(&p);
}
LIBXSMM_API void libxsmm_dfsspmdm_execute(const libxsmm_dfsspmdm* h, const double* B, double* C) { libxsmm_fsspmdm_execute(h, B, C); }
LIBXSMM_API void libxsmm_sfsspmdm_execute(const libxsmm_sfsspmdm* h, const float* B, float* C) { libxsmm_fsspmdm_execute(h, B, C); }
LIBXSMM_API void libxsmm_fsspmdm_destroy(libxsmm_fsspmdm* h) {
  if (!h) return;
  libxsmm_release_kernel((const void*)h->kernel);
  delete h;
}
LIBXSMM_API void libxsmm_dfsspmdm_destroy(libxsmm_dfsspmdm* h) { libxsmm_fsspmdm_destroy(h); }
LIBXSMM_API void libxsmm_sfsspmdm_destroy(libxsmm_sfsspmdm* h) { libxsmm_fsspmdm_destroy(h); }

// ---- created (sparse) kernels, sharded (round 6; include/libxsmm_hip.h): one handle per shard, created on the shard's device, kept together ---------------
}  // extern "C"
struct libxsmm_hip_sharded_kernel {
  enum Kind { CSR, CSC, BCSC, FSSPMDM } kind;
  struct Shard { int device; size_t begin, end; libxsmm_gemmfunction kernel; libxsmm_fsspmdm* fs; };
  std::vector<Shard> shards;
  size_t elem;                // bytes per element of C
  size_t rows;                // rows of the row-major result whose columns are split (CSR / CSC: M * N, FsSpMDM: M); BCSC: bytes of one M-block of C
  size_t axis;                // length of the split axis
};
namespace {
// lane tiles stay whole: sixteen elements cover the widest per-lane vector (four floats) of a quarter wave and every 16-byte alignment rule of the generated kernels
constexpr size_t kShardGranule = 16;
template <typename Create>
libxsmm_hip_sharded_kernel* build_sharded(libxsmm_hip_sharded_kernel::Kind kind, size_t axis, size_t granule, size_t elem, size_t rows, int nshards, const int* devices, Create create) {
  const int ndev = libxsmm_hip_device_count();
  if (ndev <= 0 || nshards <= 0 || nshards > 64 || axis == 0) return nullptr;
  libxsmm_hip_sharded_kernel* set = new libxsmm_hip_sharded_kernel();
  set->kind = kind; set->elem = elem; set->rows = rows; set->axis = axis;
  const int home = libxsmm_hip_get_device();
  bool ok = true;
  for (int s = 0; s < nshards && ok; ++s) {
    size_t b = 0, e = 0;
    libxsmm_hip_shard_range(axis, granule, nshards, s, &b, &e);
    if (e == b) continue;
    libxsmm_hip_sharded_kernel::Shard sh; sh.device = devices ? devices[s] : s % ndev; sh.begin = b; sh.end = e; sh.kernel = nullptr; sh.fs = nullptr;
    if (sh.device < 0 || sh.device >= ndev) { ok = false; break; }
    libxsmm_hip_set_device(sh.device);
    ok = create(sh, e - b);
    if (ok) set->shards.push_back(sh);
  }
  libxsmm_hip_set_device(home);
  if (!ok || set->shards.empty()) { libxsmm_hip_sharded_destroy(set); return nullptr; }
  return set;
}
}  // namespace
extern "C" {
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_create_packed_spgemm_csr_sharded(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_blasint packed_width, const unsigned int* row_ptr, const unsigned int* column_idx, const void* values, int nshards, const int* devices) {
  if (packed_width <= 0) return nullptr;
  return build_sharded(libxsmm_hip_sharded_kernel::CSR, (size_t)packed_width, kShardGranule, (size_t)LIBXSMM_TYPESIZE(shape.out_type), (size_t)shape.m * (size_t)shape.n, nshards, devices,
    [&](libxsmm_hip_sharded_kernel::Shard& sh, size_t width) { sh.kernel = libxsmm_create_packed_spgemm_csr(shape, flags, prefetch, (libxsmm_blasint)width, row_ptr, column_idx, values); return sh.kernel != nullptr; });
}
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_create_packed_spgemm_csc_sharded(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_blasint packed_width, const unsigned int* column_ptr, const unsigned int* row_idx, const void* values, int nshards, const int* devices) {
  if (packed_width <= 0 || shape.ldc == 0) return nullptr;       // (ldc == 0: C sparse -- the packed axis is a REDUCTION there, not a batch of independent columns: not shardable by this call)
  return build_sharded(libxsmm_hip_sharded_kernel::CSC, (size_t)packed_width, kShardGranule, (size_t)LIBXSMM_TYPESIZE(shape.out_type), (size_t)shape.m * (size_t)shape.n, nshards, devices,
    [&](libxsmm_hip_sharded_kernel::Shard& sh, size_t width) { sh.kernel = libxsmm_create_packed_spgemm_csc(shape, flags, prefetch, (libxsmm_blasint)width, column_ptr, row_idx, values); return sh.kernel != nullptr; });
}
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_create_packed_spgemm_bcsc_sharded(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_spgemm_config cfg, int nshards, const int* devices) {
  // the BCSC creator's shape [ref: samples/xgemm_sparse/spmm_kernel.c:548-556]: m = the number of M-blocks, packed_width = rows per block, ldc = N; C is [m_blocks][N][packed_width]
  if (cfg.packed_width <= 0 || shape.m <= 0 || shape.ldc <= 0) return nullptr;
  return build_sharded(libxsmm_hip_sharded_kernel::BCSC, (size_t)shape.m, 1, (size_t)LIBXSMM_TYPESIZE(shape.out_type), (size_t)shape.ldc * (size_t)cfg.packed_width * (size_t)LIBXSMM_TYPESIZE(shape.out_type), nshards, devices,
    [&](libxsmm_hip_sharded_kernel::Shard& sh, size_t blocks) {
      libxsmm_gemm_shape part = shape; part.m = (libxsmm_blasint)blocks;
      sh.kernel = libxsmm_create_packed_spgemm_bcsc(part, flags, prefetch, cfg); return sh.kernel != nullptr; });
}
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_fsspmdm_create_sharded(libxsmm_datatype datatype, libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K, libxsmm_blasint lda,
  const void* alpha, const void* beta, const void* a_dense, int nshards, const int* devices) {
  if (M <= 0 || N <= 0 || K <= 0) return nullptr;
  return build_sharded(libxsmm_hip_sharded_kernel::FSSPMDM, (size_t)N, kShardGranule, (size_t)LIBXSMM_TYPESIZE(datatype), (size_t)M, nshards, devices,
    [&](libxsmm_hip_sharded_kernel::Shard& sh, size_t width) {
      sh.fs = libxsmm_fsspmdm_create(datatype, M, (libxsmm_blasint)width, K, lda, (libxsmm_blasint)width, (libxsmm_blasint)width, alpha, beta, a_dense, 0, nullptr);
      if (sh.fs) sh.kernel = sh.fs->kernel;
      return sh.fs != nullptr; });
}
LIBXSMM_API int libxsmm_hip_sharded_count(const libxsmm_hip_sharded_kernel* set) { return set ? (int)set->shards.size() : 0; }
LIBXSMM_API int libxsmm_hip_sharded_range(const libxsmm_hip_sharded_kernel* set, int shard, int* device, size_t* begin, size_t* end) {
  if (!set || shard < 0 || (size_t)shard >= set->shards.size()) return EXIT_FAILURE;
  const libxsmm_hip_sharded_kernel::Shard& sh = set->shards[(size_t)shard];
  if (device) *device = sh.device;
  if (begin) *begin = sh.begin;
  if (end) *end = sh.end;
  return EXIT_SUCCESS;
}
LIBXSMM_API libxsmm_gemmfunction libxsmm_hip_sharded_handle(const libxsmm_hip_sharded_kernel* set, int shard) {
  return (set && shard >= 0 && (size_t)shard < set->shards.size()) ? set->shards[(size_t)shard].kernel : nullptr;
}
LIBXSMM_API int libxsmm_hip_sharded_launch(libxsmm_hip_sharded_kernel* set, const libxsmm_gemm_param* shard_params, int gather_device, void* gather_dst, size_t gather_dst_pitch) {
  if (!set || !shard_params || set->shards.empty()) return EXIT_FAILURE;
  libxsmm_hip_shard sh[64];
  const int n = (int)set->shards.size();
  for (int i = 0; i < n; ++i) {
    const libxsmm_hip_sharded_kernel::Shard& s = set->shards[(size_t)i];
    std::memset(&sh[i], 0, sizeof(sh[i]));
    sh[i].device = s.device; sh[i].kernel = (const void*)s.kernel; sh[i].param = &shard_params[i]; sh[i].count = 0;
    if (!gather_dst) continue;
    sh[i].gather_src = shard_params[i].c.primary;
    const size_t width = s.end - s.begin;
    if (set->kind == libxsmm_hip_sharded_kernel::BCSC) { sh[i].gather_bytes = width * set->rows; sh[i].gather_dst_offset = s.begin * set->rows; }       // rows = bytes of one M-block of C
    else if (gather_dst_pitch == 0) { sh[i].gather_bytes = set->rows * width * set->elem; sh[i].gather_dst_offset = set->rows * s.begin * set->elem; }   // slabs back to back
    else {
      sh[i].gather_bytes = width * set->elem; sh[i].gather_rows = set->rows; sh[i].gather_src_pitch = width * set->elem; sh[i].gather_dst_pitch = gather_dst_pitch;
      sh[i].gather_dst_offset = s.begin * set->elem;
      if (set->rows == 1) sh[i].gather_rows = 0;
    }
  }
  return libxsmm_hip_launch_shards(sh, n, gather_device, gather_dst);
}
LIBXSMM_API void libxsmm_hip_sharded_destroy(libxsmm_hip_sharded_kernel* set) {
  if (!set) return;
  const int home = libxsmm_hip_get_device();
  for (libxsmm_hip_sharded_kernel::Shard& sh : set->shards) {
    libxsmm_hip_set_device(sh.device);
    if (sh.fs) libxsmm_fsspmdm_destroy(sh.fs);
    else if (sh.kernel) libxsmm_release_kernel((const void*)sh.kernel);
  }
  libxsmm_hip_set_device(home);
  delete set;
}

// ---- BLAS-style entry points: alpha is taken as 1, beta as 0 or 1, exactly like LIBXSMM_XGEMM
// [ref: src/libxsmm_main.h:215-240, src/libxsmm_main.c:3933-3949] -------------------------------------------
LIBXSMM_API void libxsmm_dgemm(const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const double* alpha, const double* a, const libxsmm_blasint* lda, const double* b, const libxsmm_blasint* ldb, const double* beta, double* c, const libxsmm_blasint* ldc) {
  (void)alpha; xgemm<double>(LIBXSMM_DATATYPE_F64, transa, transb, m, n, k, a, lda, b, ldb, beta, c, ldc);
}
LIBXSMM_API void libxsmm_sgemm(const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const float* alpha, const float* a, const libxsmm_blasint* lda, const float* b, const libxsmm_blasint* ldb, const float* beta, float* c, const libxsmm_blasint* ldc) {
  (void)alpha; xgemm<float>(LIBXSMM_DATATYPE_F32, transa, transb, m, n, k, a, lda, b, ldb, beta, c, ldc);
}

// ---- allocator: pinned, device-visible host memory so unmodified drivers keep working ------------
LIBXSMM_API void* libxsmm_aligned_malloc(size_t size, size_t alignment) {
  (void)alignment;   // hipHostMalloc returns page-aligned memory
  void* p = nullptr;
  if (libxsmm_hip_device_count() > 0) {
    if (hipHostMalloc(&p, size ? size : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
  }
  if (posix_memalign(&p, 64, size ? size : 1) != 0) return nullptr;
  return p;
}
LIBXSMM_API void* libxsmm_malloc(size_t size) { return libxsmm_aligned_malloc(size, 64); }
LIBXSMM_API void libxsmm_free(const void* memory) {
  if (!memory) return;
  if (libxsmm_hip_device_count() > 0) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, memory) == hipSuccess && attr.type == hipMemoryTypeHost) { (void)hipHostFree(const_cast<void*>(memory)); return; }
    (void)hipGetLastError();
  }
  std::free(const_cast<void*>(memory));
}

// ---- input preparation: Matrix-Market readers and the BCSC builder -------------------------------------------------
// The reference keeps these in its samples [samples/xgemm_norm_packed/common_edge_proxy.h:29-320 (CSR / CSC readers),
// samples/xgemm_sparse/spmm_kernel.c:306-347 (dense -> BCSC)]; here they are library calls whose outputs live in device-visible memory
// (libxsmm_aligned_malloc: pinned host memory when a device is present), so that pattern arrays go straight to libxsmm_create_* / the BCSC
// call and value arrays straight into a.primary / b.primary without staging.  Entries may come in any order (the reference's reader
// needs them sorted by row); rows / columns without entries get empty ranges.
LIBXSMM_API int libxsmm_hip_mtx_read(const char* path, int by_column, libxsmm_datatype value_type, unsigned int** ptr, unsigned int** idx, void** values,
  unsigned int* rows, unsigned int* cols, unsigned int* nnz) {
  if (!path || !ptr || !idx || !values || !rows || !cols || !nnz || (value_type != LIBXSMM_DATATYPE_F32 && value_type != LIBXSMM_DATATYPE_F64)) return EXIT_FAILURE;
  *ptr = *idx = nullptr; *values = nullptr; *rows = *cols = *nnz = 0;
  FILE* f = std::fopen(path, "r");
  if (!f) return EXIT_FAILURE;
  // nothing may leave through the extern "C" boundary: a header that claims an absurd entry count ends in bad_alloc / length_error otherwise
  try {
  long fsize = 0;
  if (std::fseek(f, 0, SEEK_END) == 0) { fsize = std::ftell(f); std::rewind(f); }
  char line[512];
  unsigned int r = 0, c = 0, n = 0; bool header = false;
  std::vector<unsigned int> er, ec; std::vector<double> ev;
  while (std::fgets(line, sizeof(line), f)) {
    const size_t len = std::strlen(line);
    if (len + 1 == sizeof(line) && line[len - 1] != '\n') { std::fclose(f); return EXIT_FAILURE; }      // a longer line would be split and parsed as two entries
    if (line[0] == '%' || line[0] == '\n') continue;
    if (!header) {
      if (std::sscanf(line, "%u %u %u", &r, &c, &n) != 3 || r == 0 || c == 0) { std::fclose(f); return EXIT_FAILURE; }
      // an entry is at least "i j\n" = 4 bytes: a count the file cannot hold is a broken header, not a reason to reserve gigabytes
      if (fsize > 0 && (unsigned long long)n > (unsigned long long)fsize / 4ull) { std::fclose(f); return EXIT_FAILURE; }
      header = true; er.reserve(n); ec.reserve(n); ev.reserve(n);
      continue;
    }
    unsigned int i, j; double v = 1.0;
    const int got = std::sscanf(line, "%u %u %lf", &i, &j, &v);
    if (got < 2 || i == 0 || j == 0 || i > r || j > c) { std::fclose(f); return EXIT_FAILURE; }     // pattern files carry no value: 1.0
    er.push_back(i - 1); ec.push_back(j - 1); ev.push_back(got == 3 ? v : 1.0);
  }
  std::fclose(f); f = nullptr;
  if (!header || er.size() != n) return EXIT_FAILURE;
  const unsigned int outer = by_column ? c : r;
  const std::vector<unsigned int>& ko = by_column ? ec : er; const std::vector<unsigned int>& ki = by_column ? er : ec;
  const size_t es = value_type == LIBXSMM_DATATYPE_F32 ? 4 : 8;
  std::vector<unsigned int> pos((size_t)outer), order(n);       // everything that may throw comes before the C allocations
  unsigned int* p = (unsigned int*)libxsmm_aligned_malloc(sizeof(unsigned int) * ((size_t)outer + 1), 64);
  unsigned int* x = (unsigned int*)libxsmm_aligned_malloc(sizeof(unsigned int) * std::max<size_t>(n, 1), 64);
  void* v = libxsmm_aligned_malloc(es * std::max<size_t>(n, 1), 64);
  if (!p || !x || !v) { libxsmm_free(p); libxsmm_free(x); libxsmm_free(v); return EXIT_FAILURE; }
  std::fill(p, p + outer + 1, 0u);
  for (unsigned int z = 0; z < n; ++z) ++p[ko[z] + 1];
  for (unsigned int o = 0; o < outer; ++o) p[o + 1] += p[o];
  std::copy(p, p + outer, pos.begin());
  for (unsigned int z = 0; z < n; ++z) order[pos[ko[z]]++] = z;            // stable counting sort by the outer index ...
  for (unsigned int o = 0; o < outer; ++o)                                  // ... then by the inner index inside every row / column
    std::sort(order.begin() + p[o], order.begin() + p[o + 1], [&](unsigned int a, unsigned int b) { return ki[a] < ki[b]; });
  for (unsigned int z = 0; z < n; ++z) {
    x[z] = ki[order[z]];
    if (es == 4) ((float*)v)[z] = (float)ev[order[z]]; else ((double*)v)[z] = ev[order[z]];
  }
  *ptr = p; *idx = x; *values = v; *rows = r; *cols = c; *nnz = n;
  return EXIT_SUCCESS;
  } catch (...) {
    if (f) std::fclose(f);
    return EXIT_FAILURE;
  }
}

// Dense K x N operand in the reference driver's layout (column n = K contiguous values: B[n*K + k]) -> BCSC with bk x bn blocks: blocks that are
// entirely zero are dropped; values[blk][dn][dk] (k fastest), colptr over the N / bn block columns, rowidx = k-block.  `type` gives the element
// size only (F32, BF16, I8 / U8 ...).
LIBXSMM_API int libxsmm_hip_bcsc_from_dense(libxsmm_datatype type, const void* dense, int K, int N, int bk, int bn,
  unsigned int** colptr, unsigned int** rowidx, void** values, unsigned int* nnzb) {
  const size_t es = libxsmm_typesize(type);
  if (!dense || !colptr || !rowidx || !values || !nnzb || es == 0 || bk <= 0 || bn <= 0 || K <= 0 || N <= 0 || K % bk || N % bn) return EXIT_FAILURE;
  const int nkb = K / bk, nnb = N / bn;
  const unsigned char* d = (const unsigned char*)dense;
  std::vector<unsigned int> cp(1, 0u), ri;
  for (int nb = 0; nb < nnb; ++nb) {
    for (int kb = 0; kb < nkb; ++kb) {
      bool nz = false;
      for (int dn = 0; dn < bn && !nz; ++dn) {
        const unsigned char* row = d + ((size_t)(nb * bn + dn) * K + (size_t)kb * bk) * es;
        for (size_t q = 0; q < (size_t)bk * es; ++q) if (row[q] != 0) { nz = true; break; }      // any set bit keeps the block (so does -0.0)
      }
      if (nz) ri.push_back((unsigned int)kb);
    }
    cp.push_back((unsigned int)ri.size());
  }
  unsigned int* p = (unsigned int*)libxsmm_aligned_malloc(sizeof(unsigned int) * cp.size(), 64);
  unsigned int* x = (unsigned int*)libxsmm_aligned_malloc(sizeof(unsigned int) * std::max<size_t>(ri.size(), 1), 64);
  unsigned char* v = (unsigned char*)libxsmm_aligned_malloc(es * (size_t)bk * bn * std::max<size_t>(ri.size(), 1), 64);
  if (!p || !x || !v) { libxsmm_free(p); libxsmm_free(x); libxsmm_free(v); return EXIT_FAILURE; }
  std::copy(cp.begin(), cp.end(), p); std::copy(ri.begin(), ri.end(), x);
  size_t blk = 0;
  for (int nb = 0; nb < nnb; ++nb)
    for (unsigned int b = cp[nb]; b < cp[nb + 1]; ++b, ++blk)
      for (int dn = 0; dn < bn; ++dn)
        std::memcpy(v + ((blk * bn + dn) * bk) * es, d + ((size_t)(nb * bn + dn) * K + (size_t)ri[b] * bk) * es, (size_t)bk * es);
  *colptr = p; *rowidx = x; *values = v; *nnzb = (unsigned int)ri.size();
  return EXIT_SUCCESS;
}

// ---- timer / rng -----------------------------------------------------------------------------------------
LIBXSMM_API libxsmm_timer_tickint libxsmm_timer_tick(void) {
  return (libxsmm_timer_tickint)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
LIBXSMM_API double libxsmm_timer_duration(libxsmm_timer_tickint t0, libxsmm_timer_tickint t1) { return (t1 >= t0 ? (double)(t1 - t0) : 0.0) * 1e-9; }

// The scalar generators are the C library's 48-bit LCG (lrand48 / drand48), seeded by libxsmm_rng_set_seed -- a driver seeded with 555
// therefore builds the same matrices on both libraries [ref: src/libxsmm_utils.c:20-87].  libxsmm_rng_u32 draws uniformly in [0, n) by
// rejecting the incomplete tail of the 31-bit range.
LIBXSMM_API double libxsmm_rng_f64(void) { return drand48(); }
LIBXSMM_API unsigned int libxsmm_rng_u32(unsigned int n) {
  if (n < 2) return 0;
  const unsigned int range = 1u << 31, lim = n < range ? n : range, usable = (range / lim) * lim;
  unsigned int r;
  do r = (unsigned int)lrand48(); while (r >= usable);
  return n <= lim ? r % lim : (unsigned int)(((double)n / lim) * r + 0.5);     // n beyond the generator's 31 bits: stretched
}
LIBXSMM_API void libxsmm_rng_seq(void* data, size_t nbytes) {                   // consecutive lrand48 words, the tail from one more draw
  unsigned char* dst = (unsigned char*)data;
  for (size_t done = 0; done < nbytes; done += 4) {
    const unsigned int r = (unsigned int)lrand48();
    std::memcpy(dst + done, &r, nbytes - done < 4 ? nbytes - done : 4);
  }
}

// ---- bf16 conversions [ref: src/libxsmm_math.c:640-704] ------------------------------------------------
static unsigned int f2u(float f) { unsigned int u; std::memcpy(&u, &f, 4); return u; }
LIBXSMM_API float libxsmm_convert_bf16_to_f32(libxsmm_bfloat16 in) { const unsigned int u = (unsigned int)in << 16; float f; std::memcpy(&f, &u, 4); return f; }
static unsigned int bf16_front(unsigned int u, bool* special) {
  if ((u & 0x7f800000u) == 0) u &= 0x80000000u;
  *special = (u & 0x7f800000u) == 0x7f800000u;
  if (*special && (u & 0x007fffffu)) u |= 0x00400000u;
  return u;
}
LIBXSMM_API libxsmm_bfloat16 libxsmm_convert_f32_to_bf16_rne(float in) {
  bool sp; unsigned int u = bf16_front(f2u(in), &sp);
  if (!sp) u += 0x00007fffu + ((u >> 16) & 1u);
  return (libxsmm_bfloat16)(u >> 16);
}
LIBXSMM_API libxsmm_bfloat16 libxsmm_convert_f32_to_bf16_truncate(float in) { bool sp; return (libxsmm_bfloat16)(bf16_front(f2u(in), &sp) >> 16); }
LIBXSMM_API void libxsmm_rne_convert_fp32_bf16(const float* in, libxsmm_bfloat16* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = libxsmm_convert_f32_to_bf16_rne(in[i]); }
LIBXSMM_API void libxsmm_truncate_convert_f32_bf16(const float* in, libxsmm_bfloat16* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = libxsmm_convert_f32_to_bf16_truncate(in[i]); }
LIBXSMM_API void libxsmm_convert_bf16_f32(const libxsmm_bfloat16* in, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = libxsmm_convert_bf16_to_f32(in[i]); }

// ---- matdiff: the subset of statistics the drivers read [ref: src/libxsmm_matdiff.h; libxsmm_math.c:35-300] ----
LIBXSMM_API void libxsmm_matdiff_clear(libxsmm_matdiff_info* info) {
  if (!info) return;
  std::memset(info, 0, sizeof(*info));
  info->m = info->n = info->i = -1;                            // no location with a difference yet
  info->min_ref = info->min_tst = INFINITY; info->max_ref = info->max_tst = -INFINITY;
  info->rsq = INFINITY;                                        // invalid rather than 1.0 [ref: libxsmm_math.c:456-466]
}
static double md_load(libxsmm_datatype t, const void* p, size_t i) {
  switch (t) {
    case LIBXSMM_DATATYPE_F64: return ((const double*)p)[i];
    case LIBXSMM_DATATYPE_F32: return ((const float*)p)[i];
    case LIBXSMM_DATATYPE_BF16: return libxsmm_convert_bf16_to_f32(((const libxsmm_bfloat16*)p)[i]);
    case LIBXSMM_DATATYPE_I32: return ((const int*)p)[i];
    case LIBXSMM_DATATYPE_I16: return ((const short*)p)[i];
    case LIBXSMM_DATATYPE_I8: return ((const signed char*)p)[i];
    case LIBXSMM_DATATYPE_I64: return (double)((const long long*)p)[i];
    case LIBXSMM_DATATYPE_U32: return ((const unsigned int*)p)[i];
    case LIBXSMM_DATATYPE_U16: return ((const unsigned short*)p)[i];
    case LIBXSMM_DATATYPE_U8: return ((const unsigned char*)p)[i];
    case LIBXSMM_DATATYPE_F16: return lowp::f16_to_f32(((const unsigned short*)p)[i]);      // [ref: libxsmm_math.c matdiff type switch]
    case LIBXSMM_DATATYPE_BF8: return lowp::bf8_to_f32(((const unsigned char*)p)[i]);
    case LIBXSMM_DATATYPE_HF8: return lowp::hf8_to_f32(((const unsigned char*)p)[i]);
    default: return NAN;
  }
}
// Statistics as the reference defines them [behaviour: src/libxsmm_matdiff.h, src/libxsmm_math.c:35-300; pinned by the reference's own
// tests/matdiff.c, run by tests/test_reference_drivers_gpu.py]: element (i, j) = x[j * ld + i]; a single column is looked at as a row;
//   linf_abs / linf_rel  largest |r - t| (with its location and values) / largest |r - t| / |r|   (|t| when r = 0, else the difference itself)
//   normi_abs            max over j of sum_i |r - t|   (one contiguous line), normi_rel = that over the same norm of the reference
//   norm1_abs            max over i of sum_j |r - t|   (across the lines),    norm1_rel likewise
//   l2_abs, l2_rel       sqrt(sum d^2), sqrt(sum (d / |r|)^2);  normf_rel = sqrt(sum d^2 / sum r^2);  l1_ref / l1_tst = sum |r|, sum |t|
//   rsq                  max(0, 1 - sum d^2 / sum (r - avg_ref)^2), avg = l1 / count;  var = that sum / count
// A NaN or infinity in the test set (that the reference does not share) ends the scan: every error statistic becomes infinity.
LIBXSMM_API int libxsmm_matdiff(libxsmm_matdiff_info* info, libxsmm_datatype datatype, libxsmm_blasint m, libxsmm_blasint n,
  const void* ref, const void* tst, const libxsmm_blasint* ldref, const libxsmm_blasint* ldtst) {
  bool swapped = false;
  if (!ref && tst) { ref = tst; tst = nullptr; swapped = true; }
  size_t ldr = ldref ? (size_t)*ldref : (size_t)m, ldt = ldtst ? (size_t)*ldtst : (size_t)m;
  if (!info || !ref || m < 0 || n < 0 || (size_t)m > ldr || (size_t)m > ldt || typesize((int)datatype) == 0) return EXIT_FAILURE;
  long long rows = m, lines = n;
  if (n == 1) { rows = 1; lines = m; ldr = ldt = 1; }          // a column vector is treated as a row vector (same statistics for both)
  libxsmm_matdiff_clear(info);
  const double inf = info->min_ref;                            // clear() leaves +infinity here
  const size_t count = (size_t)m * (size_t)n;
  const auto div_or = [](double num, double den, double fallback) { return den > 0 ? num / den : fallback; };
  double sum_d2 = 0, sum_r2 = 0, sum_t2 = 0, sum_rel2 = 0, l1_ref = 0, l1_tst = 0, line_ref_max = 0, line_tst_max = 0;
  int bad = 0;                                                 // 1: test value not finite, 2: reference value not finite
  for (long long j = 0; j < lines && !bad; ++j) {
    double line_d = 0, line_r = 0, line_t = 0;
    for (long long i = 0; i < rows; ++i) {
      const double r = md_load(datatype, ref, (size_t)j * ldr + (size_t)i), t = tst ? md_load(datatype, tst, (size_t)j * ldt + (size_t)i) : 0.0;
      const double ra = std::fabs(r), ta = std::fabs(t);
      if (r < info->min_ref) info->min_ref = r;
      if (r > info->max_ref) info->max_ref = r;
      if (t == t && (ta < inf || t == r)) {
        const double d = tst ? std::fabs(r - t) : 0.0, rel = div_or(d, ra, ta);
        if (t < info->min_tst) info->min_tst = t;
        if (t > info->max_tst) info->max_tst = t;
        if (info->linf_abs < d) { info->linf_abs = d; info->v_ref = r; info->v_tst = t; info->m = (libxsmm_blasint)i; info->n = (libxsmm_blasint)j; }
        if (info->linf_rel < rel) info->linf_rel = rel;
        if (rel * rel < inf) sum_rel2 += rel * rel;
        line_r += ra; line_t += ta; line_d += d;
        sum_r2 += r * r; sum_t2 += t * t;
        if (d * d < inf) sum_d2 += d * d;
      } else {
        bad = (r == r && ra < inf) ? 1 : 2;
        info->m = (libxsmm_blasint)i; info->n = (libxsmm_blasint)j; info->v_ref = r; info->v_tst = t;
        break;
      }
    }
    if (bad) break;
    l1_ref += line_r; l1_tst += line_t;
    if (info->normi_abs < line_d) info->normi_abs = line_d;
    if (line_ref_max < line_r) line_ref_max = line_r;
    if (line_tst_max < line_t) line_tst_max = line_t;
  }
  info->l1_ref = l1_ref; info->l1_tst = l1_tst;
  if (!bad) {
    if (count) { info->avg_ref = l1_ref / (double)count; info->avg_tst = l1_tst / (double)count; }
    info->normi_rel = div_or(info->normi_abs, line_ref_max, line_tst_max);
    info->normf_rel = std::sqrt(div_or(sum_d2, sum_r2, std::min(sum_t2 * sum_t2, sum_d2)));
    double cross_ref_max = 0, var_r = 0, var_t = 0;
    for (long long i = 0; i < rows; ++i) {
      double cross_d = 0, cross_r = 0;
      for (long long j = 0; j < lines; ++j) {
        const double r = md_load(datatype, ref, (size_t)j * ldr + (size_t)i), t = tst ? md_load(datatype, tst, (size_t)j * ldt + (size_t)i) : 0.0;
        const double rd = r - info->avg_ref, td = t - info->avg_tst;
        var_r += rd * rd; var_t += td * td;
        cross_r += std::fabs(r); cross_d += tst ? std::fabs(r - t) : 0.0;
      }
      if (info->norm1_abs < cross_d) info->norm1_abs = cross_d;
      if (cross_ref_max < cross_r) cross_ref_max = cross_r;
    }
    info->norm1_rel = div_or(info->norm1_abs, cross_ref_max, info->norm1_abs);
    info->rsq = std::max(0.0, 1.0 - div_or(sum_d2, var_r, sum_d2));
    info->var_ref = count ? var_r / (double)count : var_r; info->var_tst = count ? var_t / (double)count : var_t;
    info->l2_abs = std::sqrt(sum_d2); info->l2_rel = std::sqrt(sum_rel2);
  } else {
    info->norm1_abs = info->norm1_rel = info->normi_abs = info->normi_rel = info->normf_rel = info->linf_abs = info->linf_rel = info->l2_abs = info->l2_rel = inf;
    if (bad == 1) { info->l1_tst = info->var_tst = inf; info->avg_tst = info->v_tst; info->min_tst = inf; info->max_tst = -inf; }
    else { info->l1_ref = info->var_ref = inf; info->avg_ref = info->v_ref; info->min_ref = inf; info->max_ref = -inf; }
  }
  if (n == 1) std::swap(info->m, info->n);
  if (swapped) {
    info->min_tst = info->min_ref; info->min_ref = 0; info->max_tst = info->max_ref; info->max_ref = 0; info->avg_tst = info->avg_ref; info->avg_ref = 0;
    info->var_tst = info->var_ref; info->var_ref = 0; info->l1_tst = info->l1_ref; info->l1_ref = 0; info->v_tst = info->v_ref; info->v_ref = 0;
  }
  return EXIT_SUCCESS;
}

LIBXSMM_API void libxsmm_matdiff_reduce(libxsmm_matdiff_info* out, const libxsmm_matdiff_info* in) {
  if (!out || !in) { libxsmm_matdiff_clear(out); return; }
  const double eps_in = libxsmm_matdiff_epsilon(in), eps_out = libxsmm_matdiff_epsilon(out);
  if (out->linf_abs <= in->linf_abs) { out->linf_abs = in->linf_abs; out->linf_rel = in->linf_rel; }
  if (out->norm1_abs <= in->norm1_abs) { out->norm1_abs = in->norm1_abs; out->norm1_rel = in->norm1_rel; }
  if (out->normi_abs <= in->normi_abs) { out->normi_abs = in->normi_abs; out->normi_rel = in->normi_rel; }
  if (out->l2_abs <= in->l2_abs) { out->l2_abs = in->l2_abs; out->l2_rel = in->l2_rel; }
  out->normf_rel = std::max(out->normf_rel, in->normf_rel);
  out->var_ref = std::max(out->var_ref, in->var_ref); out->var_tst = std::max(out->var_tst, in->var_tst);
  out->max_ref = std::max(out->max_ref, in->max_ref); out->max_tst = std::max(out->max_tst, in->max_tst);
  out->min_ref = std::min(out->min_ref, in->min_ref); out->min_tst = std::min(out->min_tst, in->min_tst);
  if (eps_out < eps_in) { out->rsq = in->rsq; out->v_ref = in->v_ref; out->v_tst = in->v_tst; out->m = in->m; out->n = in->n; out->i = in->r; }
  out->avg_ref = 0.5 * (out->avg_ref + in->avg_ref); out->avg_tst = 0.5 * (out->avg_tst + in->avg_tst);
  out->l1_ref += in->l1_ref; out->l1_tst += in->l1_tst;
  ++out->r;
}
LIBXSMM_API double libxsmm_matdiff_epsilon(const libxsmm_matdiff_info* in) {
  if (!in) return 0.0;
  if (in->rsq > 0) return std::min(in->normf_rel, in->linf_abs) / in->rsq;   // [ref: libxsmm_math.c:322-332]
  return std::max(std::min(in->norm1_abs, in->normi_abs), std::max(in->linf_abs, in->l2_abs));
}

}  // extern "C"
