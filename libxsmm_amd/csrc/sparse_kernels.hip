// sparse_kernels.hip -- packed / sparse kernels for gfx950.
//
// One device kernel serves the three "fixed sparse operator times a dense packed panel" entry
// points of the reference:
//   libxsmm_create_packed_spgemm_csr (A sparse)  C[m][n][p] (+)= sum_z a[z] * B[col[z]][n][p]
//       [ref: src/generator_packed_spgemm_csr_asparse_avx_avx2_avx512.c:336-470]
//   libxsmm_create_packed_spgemm_csc / _csr (B sparse)  C[m][n][p] (+)= sum_z A[m][row[z]][p] * b[z]
//       [ref: samples/xgemm_norm_packed/bsparse_packed_csc.c:133-150]
//   libxsmm_fsspmdm / libxsmm_create_spgemm_csr_areg  C[i][j] = sum_z a[z] * B[col[z]][j] (+ C)
//       [ref: src/libxsmm_fsspmdm.c:491-514; samples/xgemm_sparse_Ainregs/pyfr_driver_asp_reg.c:351-375]
// All of them are  Y[r][q] (+)= sum_{z in row r} val[z] * X[idx[z]][q]  with q running over a long
// contiguous axis (packed width P, or n*P, or the N of FsSpMDM): lanes run along q, so every
// global access is a coalesced row segment, the sparsity pattern and the values are wave-uniform
// (scalar loads), and each X element is read from HBM exactly once per slab: the block stages its
// column slice of X (inner x width) in LDS, then walks the pattern out of LDS.  These kernels are
// HBM-bound (flops/byte ~ nnz/(K+M)/4); the roofline is bytes = (K + M(1+[beta=1])) * ncols * size.
//
// The block-sparse BCSC kernel keeps its pattern at run time (colptr/rowidx arrive with every call)
// [ref: samples/xgemm_sparse/spmm_kernel.c:423-456].
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "internal.hpp"

namespace xamd {

// kernel-argument-block pointers are generic to the compiler; everything dereferenced here is global memory
#define GM __attribute__((address_space(1)))

template <typename T, int VEC> struct VecOf;
template <> struct VecOf<float, 1> { typedef float type; };
template <> struct VecOf<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<double, 1> { typedef double type; };
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };

template <typename T> __device__ __forceinline__ T load_val(const void* vals, unsigned int z, int vals_are_f64) {
  return vals_are_f64 ? (T)((GM const double*)vals)[z] : ((GM const T*)vals)[z];
}

// grid: x = column blocks, y = slabs.  Dynamic LDS: inner * blockDim.x * VEC elements when STAGE.
template <typename T, int VEC, bool STAGE>
__global__ void spmm_panel_kernel(SpmmArgs p) {
  typedef typename VecOf<T, VEC>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  vec_t* tile = (vec_t*)smem;                                  // [inner][blockDim.x]
  const int tid = threadIdx.x, nthr = blockDim.x;
  const long long q0 = ((long long)blockIdx.x * nthr + tid) * VEC;
  const bool active = q0 < p.ncols;                            // ncols % VEC == 0 by construction
  GM const T* x = (GM const T*)p.x + (long long)blockIdx.y * p.outer_x;
  GM T* y = (GM T*)p.y + (long long)blockIdx.y * p.outer_y;
  GM const unsigned int* ptr = (GM const unsigned int*)p.ptr;
  GM const unsigned int* idx = (GM const unsigned int*)p.idx;
  GM const unsigned int* vmap = (GM const unsigned int*)p.vmap;
  if (STAGE) {
    // each thread copies its own column(s): no other thread reads them -> no barrier needed
    if (active) for (int k = 0; k < p.inner; ++k) tile[(long long)k * nthr + tid] = *(GM const vec_t*)(x + (long long)k * p.ld_x + q0);
  }
  if (!active) return;
  for (int r = 0; r < p.rows; ++r) {
    const unsigned int z0 = ptr[r], z1 = ptr[r + 1];
    if (z0 == z1 && (p.skip_empty || !p.beta0)) continue;      // untouched row
    vec_t acc;
    GM vec_t* yp = (GM vec_t*)(y + (long long)r * p.ld_y + q0);
    if (p.beta0) { for (int v = 0; v < VEC; ++v) ((T*)&acc)[v] = (T)0; } else acc = *yp;
    for (unsigned int z = z0; z < z1; ++z) {
      const T a = load_val<T>(p.vals, vmap ? vmap[z] : z, p.vals_are_f64);
      const unsigned int k = idx[z];
      const vec_t xv = STAGE ? tile[(long long)k * nthr + tid] : *(GM const vec_t*)(x + (long long)k * p.ld_x + q0);
#pragma unroll
      for (int v = 0; v < VEC; ++v) ((T*)&acc)[v] = fma(a, ((const T*)&xv)[v], ((T*)&acc)[v]);
    }
    *yp = acc;
  }
}

template <typename T, int VEC>
static int launch_spmm_t(const SpmmArgs& a, hipStream_t st, const char** name) {
  // pick the block width so that the staged slice fits a 64 KiB LDS budget (>= 2 blocks per CU)
  int nthr = 256;
  const size_t per_thread = (size_t)a.inner * VEC * sizeof(T);
  while (nthr > 64 && per_thread * nthr > 65536) nthr >>= 1;
  const bool stage = per_thread * nthr <= 65536;
  const long long cols_per_block = (long long)nthr * VEC;
  dim3 grid((unsigned int)((a.ncols + cols_per_block - 1) / cols_per_block), (unsigned int)a.nouter);
  if (stage) {
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)spmm_panel_kernel<T, VEC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr_set = true; }
    hipLaunchKernelGGL((spmm_panel_kernel<T, VEC, true>), grid, dim3(nthr), per_thread * nthr, st, a);
    if (name) *name = "spmm_panel_kernel<lds>";
  } else {
    hipLaunchKernelGGL((spmm_panel_kernel<T, VEC, false>), grid, dim3(nthr), 0, st, a);
    if (name) *name = "spmm_panel_kernel<direct>";
  }
  return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Streaming form of the same operation (the hot path).  One wave per workgroup owns a slab of
// 64*E consecutive columns; it
//   1. fires `inner` LDS-DMA loads (global_load_lds: global -> LDS without touching VGPRs, GL bytes
//      per lane, NL loads per row) so the whole K x slab slice of X is in flight at once,
//   2. (first slab only, under the shadow of 1.) copies the pattern into LDS as (k*RB, value) pairs
//      so that the walk below never waits on a dependent global/scalar load,
//   3. walks the rows: per non-zero one broadcast ds_read of the pair, one ds_read of its own
//      column(s), E FMAs; per row one coalesced store.
// X is read from HBM exactly once and Y written once; nothing is shared between waves, so there is
// no barrier, and occupancy (160 KiB LDS / (inner*RB + pattern)) supplies the load/compute overlap.
// ------------------------------------------------------------------------------------------------
template <typename T> struct PatPair;
template <> struct PatPair<float> { unsigned int koff; float v; };
template <> struct __attribute__((aligned(16))) PatPair<double> { unsigned int koff; unsigned int pad; double v; };
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int GL> __device__ __forceinline__ void glds(GM const char* src, char* lds_dst);   // lane-linear LDS destination
template <> __device__ __forceinline__ void glds<4>(GM const char* src, char* lds_dst) { __builtin_amdgcn_global_load_lds((GM const void*)src, (lds_ptr_t)lds_dst, 4, 0, 0); }
template <> __device__ __forceinline__ void glds<16>(GM const char* src, char* lds_dst) { __builtin_amdgcn_global_load_lds((GM const void*)src, (lds_ptr_t)lds_dst, 16, 0, 0); }

template <typename T, int GL, int NL>
__global__ __launch_bounds__(64) void spmm_stream_kernel(SpmmArgs p, unsigned int slabs_per_outer, unsigned int total_slabs) {
  constexpr int RB = 64 * GL * NL;                         // staged bytes per X row
  constexpr int E = GL * NL / (int)sizeof(T);              // columns per lane
  typedef T vec_t __attribute__((ext_vector_type(E)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                                        // [inner][RB]
  PatPair<T>* pairs = (PatPair<T>*)(smem + (size_t)p.inner * RB);            // [nnz]
  unsigned int* rp = (unsigned int*)(pairs + p.nnz);                         // [rows + 1]
  const int lane = threadIdx.x;
  const long long ldx_b = p.ld_x * (long long)sizeof(T), ldy = p.ld_y;
  bool first = true;
  for (unsigned int s = blockIdx.x; s < total_slabs; s += gridDim.x) {
    const unsigned int outer = s / slabs_per_outer, cs = s - outer * slabs_per_outer;
    const long long q0 = (long long)cs * (64 * E);
    const long long left = p.ncols - q0;                                     // > 0
    const int valid_b = (int)(left * (long long)sizeof(T) < (long long)RB ? left * (long long)sizeof(T) : (long long)RB);
    GM const char* xb = (GM const char*)p.x + ((long long)outer * p.outer_x + q0) * (long long)sizeof(T) + lane * GL;
    for (int k = 0; k < p.inner; ++k) {
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        if (j * 64 * GL + lane * GL < valid_b)
          glds<GL>(xb + (long long)k * ldx_b + j * 64 * GL, tile + k * RB + j * 64 * GL);
      }
    }
    if (first) {
      first = false;
      GM const unsigned int* ptr = (GM const unsigned int*)p.ptr;
      GM const unsigned int* idx = (GM const unsigned int*)p.idx;
      GM const unsigned int* vmap = (GM const unsigned int*)p.vmap;
      for (unsigned int z = lane; z < p.nnz; z += 64) {
        PatPair<T> pr;
        pr.koff = idx[z] * (unsigned int)RB;
        pr.v = load_val<T>(p.vals, vmap ? vmap[z] : z, p.vals_are_f64);
        pairs[z] = pr;
      }
      for (int r = lane; r <= p.rows; r += 64) rp[r] = ptr[r];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const bool active = (long long)lane * E < left;                          // ncols % E == 0 by construction
    GM T* yb = (GM T*)p.y + (long long)outer * p.outer_y + q0 + lane * E;
    const char* mine = tile + lane * (E * (int)sizeof(T));
    for (int r = 0; r < p.rows; ++r) {
      const unsigned int z0 = rp[r], z1 = rp[r + 1];
      if (z0 == z1 && (p.skip_empty || !p.beta0)) continue;                  // untouched row
      GM vec_t* yp = (GM vec_t*)(yb + (long long)r * ldy);
      vec_t old;
      if (!p.beta0 && active) old = *yp;
      vec_t acc = (vec_t)(T)0;
#pragma unroll 4
      for (unsigned int z = z0; z < z1; ++z) {
        const PatPair<T> pr = pairs[z];
        const vec_t xv = *(const vec_t*)(mine + pr.koff);
        acc = __builtin_elementwise_fma((vec_t)pr.v, xv, acc);
      }
      if (active) { if (!p.beta0) acc += old; *yp = acc; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // every LDS read retired before the tile is refilled
  }
}

static int g_num_cus = 0;
template <typename T, int GL, int NL>
static int launch_spmm_stream(const SpmmArgs& a, hipStream_t st, const char** name, const char* label) {
  constexpr int RB = 64 * GL * NL;
  constexpr int E = GL * NL / (int)sizeof(T);
  const size_t lds = (size_t)a.inner * RB + (size_t)a.nnz * sizeof(PatPair<T>) + ((size_t)a.rows + 1) * sizeof(unsigned int);
  if (lds > 160 * 1024) return -1;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)spmm_stream_kernel<T, GL, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  if (g_num_cus == 0) {
    int dev = 0, cus = 0; (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_num_cus = cus;
  }
  const long long per_outer = (a.ncols + 64 * E - 1) / (64 * E);
  const long long total = per_outer * a.nouter;
  if (per_outer >= (1ll << 31) || total >= (1ll << 31)) return -1;
  long long resident = (long long)(160 * 1024 / lds); if (resident > 32) resident = 32;
  long long grid = (long long)g_num_cus * resident;
  if (grid > total) grid = total;
  hipLaunchKernelGGL((spmm_stream_kernel<T, GL, NL>), dim3((unsigned int)grid), dim3(64), lds, st, a, (unsigned int)per_outer, (unsigned int)total);
  if (name) *name = label;
  return (int)hipGetLastError();
}

int launch_spmm(const SpmmArgs& a, void* stream, const char** name) {
  hipStream_t st = (hipStream_t)stream;
  if (a.ncols <= 0 || a.rows <= 0 || a.nouter <= 0) { if (name) *name = "(empty)"; return 0; }
  const int sz = (a.dtype == LIBXSMM_DATATYPE_F64) ? 8 : 4;
  // widest vector such that every row segment start stays aligned
  auto aligned = [&](int vec) {
    const unsigned long long bytes = (unsigned long long)vec * sz;
    return (a.ncols % vec == 0) && (a.ld_x % vec == 0) && (a.ld_y % vec == 0) && (a.outer_x % vec == 0) && (a.outer_y % vec == 0) &&
           ((unsigned long long)(size_t)a.x % bytes == 0) && ((unsigned long long)(size_t)a.y % bytes == 0);
  };
  // streaming kernel first (pattern + K x slab slice in LDS); the panel kernel below takes what does not fit
  {
    static const int variant = []() { const char* e = getenv("LIBXSMM_HIP_SPMM_VARIANT"); return e ? atoi(e) : 0; }();
    int rc = -1;
    if (variant >= 0 && a.nnz > 0) {
      if (a.dtype == LIBXSMM_DATATYPE_F64) {
        if (variant == 1 && aligned(2)) rc = launch_spmm_stream<double, 16, 1>(a, st, name, "spmm_stream_kernel<f64,16x1>");
        if (rc < 0) rc = launch_spmm_stream<double, 4, 2>(a, st, name, "spmm_stream_kernel<f64,4x2>");
      } else {
        if (variant == 1 && aligned(4)) rc = launch_spmm_stream<float, 16, 1>(a, st, name, "spmm_stream_kernel<f32,16x1>");
        if (rc < 0) rc = launch_spmm_stream<float, 4, 1>(a, st, name, "spmm_stream_kernel<f32,4x1>");
      }
      if (rc >= 0) return rc;
    }
  }
  if (a.dtype == LIBXSMM_DATATYPE_F64) {
    if (aligned(2)) return launch_spmm_t<double, 2>(a, st, name);
    return launch_spmm_t<double, 1>(a, st, name);
  }
  if (aligned(4)) return launch_spmm_t<float, 4>(a, st, name);
  if (aligned(2)) return launch_spmm_t<float, 2>(a, st, name);
  return launch_spmm_t<float, 1>(a, st, name);
}

// ------------------------------------------------------------------------------------------------
// BCSC: C[mb][n][i] = beta*C + sum_{blk in block-column n/bn} sum_dk A[mb][k0+dk][i] * Bv[blk][n%bn][dk]
// generic form: one thread per (i, n) of one M-block; lanes along i (contiguous in A and C).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(unsigned short x) { return __uint_as_float((unsigned int)x << 16); }
__device__ __forceinline__ unsigned short f2bf_rne(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7f800000u) == 0u) u &= 0x80000000u;
  if ((u & 0x7f800000u) == 0x7f800000u) { if (u & 0x007fffffu) u |= 0x00400000u; }
  else u += 0x00007fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void bcsc_generic_kernel(BcscArgs p) {
  const int tiles_i = (p.M + 63) / 64;
  const long long per_block = (long long)tiles_i * p.N;
  const long long blk = blockIdx.x * 4LL + threadIdx.y;
  if (blk >= per_block * p.m_blocks) return;
  const int mb = (int)(blk / per_block);
  const int t = (int)(blk % per_block);
  const int i = (t % tiles_i) * 64 + threadIdx.x;
  const int n = t / tiles_i;
  if (i >= p.M) return;
  const int nb = n / p.bn, dn = n % p.bn;
  const long long cidx = (long long)mb * p.N * p.M + (long long)n * p.M + i;
  const bool f32 = (p.a_type == LIBXSMM_DATATYPE_F32);
  float acc = 0.0f;
  if (!p.beta0) acc = (p.c_type == LIBXSMM_DATATYPE_F32) ? ((GM const float*)p.c)[cidx] : bf2f(((GM const unsigned short*)p.c)[cidx]);
  const long long abase = (long long)mb * p.K * p.M;
  GM const unsigned int* colptr = (GM const unsigned int*)p.colptr;
  GM const unsigned int* rowidx = (GM const unsigned int*)p.rowidx;
  for (unsigned int b = colptr[nb]; b < colptr[nb + 1]; ++b) {
    const int k0 = (int)rowidx[b] * p.bk;
    const long long boff = ((long long)b * p.bn + dn) * p.bk;
    for (int dk = 0; dk < p.bk; ++dk) {
      const int k = k0 + dk;
      float av, bv;
      if (f32) {
        av = ((GM const float*)p.a)[abase + (long long)k * p.M + i];
        bv = ((GM const float*)p.bvals)[boff + dk];
      } else {
        const long long ai = p.vnni_a ? ((long long)(k / 2) * (p.M * 2) + (long long)i * 2 + (k % 2)) : ((long long)k * p.M + i);
        av = bf2f(((GM const unsigned short*)p.a)[abase + ai]);
        bv = bf2f(((GM const unsigned short*)p.bvals)[boff + dk]);
      }
      acc = fmaf(av, bv, acc);
    }
  }
  if (p.c_type == LIBXSMM_DATATYPE_F32) ((GM float*)p.c)[cidx] = acc; else ((GM unsigned short*)p.c)[cidx] = f2bf_rne(acc);
}

int launch_bcsc(const BcscArgs& a, void* stream, const char** name) {
  hipStream_t st = (hipStream_t)stream;
  if (a.m_blocks <= 0 || a.M <= 0 || a.N <= 0) { if (name) *name = "(empty)"; return 0; }
  const long long blocks = (long long)((a.M + 63) / 64) * a.N * a.m_blocks;
  hipLaunchKernelGGL(bcsc_generic_kernel, dim3((unsigned int)((blocks + 3) / 4)), dim3(64, 4), 0, st, a);
  if (name) *name = "bcsc_generic_kernel";
  return (int)hipGetLastError();
}

}  // namespace xamd
