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
#include <utility>
#include "internal.hpp"
#include "bf16_cvt.hpp"

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
    // (a 16-row x 1 form of the streaming kernel was measured in round 2 and lost; removed in round 6)
    if (a.nnz > 0) {
      const int rc = a.dtype == LIBXSMM_DATATYPE_F64 ? launch_spmm_stream<double, 4, 2>(a, st, name, "spmm_stream_kernel<f64,4x2>")
                                                     : launch_spmm_stream<float, 4, 1>(a, st, name, "spmm_stream_kernel<f32,4x1>");
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
  const long long abase = (long long)mb * p.K * p.M;
  GM const unsigned int* colptr = (GM const unsigned int*)p.colptr;
  GM const unsigned int* rowidx = (GM const unsigned int*)p.rowidx;
  if (p.a_type == LIBXSMM_DATATYPE_I8 || p.a_type == LIBXSMM_DATATYPE_U8) {     // 8-bit integers -> int32, exact [ref: spmm_kernel.c:153-217]
    const bool ua = p.a_type == LIBXSMM_DATATYPE_U8, ub = p.b_type == LIBXSMM_DATATYPE_U8;
    int iacc = p.beta0 ? 0 : ((GM const int*)p.c)[cidx];
    for (unsigned int b = colptr[nb]; b < colptr[nb + 1]; ++b) {
      const int k0 = (int)rowidx[b] * p.bk;
      const long long boff = ((long long)b * p.bn + dn) * p.bk;
      for (int dk = 0; dk < p.bk; ++dk) {
        const int k = k0 + dk;
        const long long ai = p.vnni_a ? ((long long)(k / 4) * (p.M * 4) + (long long)i * 4 + (k % 4)) : ((long long)k * p.M + i);
        const unsigned char ab = ((GM const unsigned char*)p.a)[abase + ai], bb = ((GM const unsigned char*)p.bvals)[boff + dk];
        iacc += (ua ? (int)ab : (int)(signed char)ab) * (ub ? (int)bb : (int)(signed char)bb);
      }
    }
    ((GM int*)p.c)[cidx] = iacc;
    return;
  }
  float acc = 0.0f;
  if (!p.beta0) acc = (p.c_type == LIBXSMM_DATATYPE_F32) ? ((GM const float*)p.c)[cidx] : bf2f(((GM const unsigned short*)p.c)[cidx]);
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

// ------------------------------------------------------------------------------------------------
// BCSC on the matrix cores (bf16, VNNI-2 A, bk % 32 == 0, bn in {16,32,64}, M % 16 == 0).
// One wave owns (M-block mb, 64 rows i, 64 columns n) of C and keeps it in 64 accumulator VGPRs
// (up to 4 x 4 tiles of v_mfma_f32_16x16x32_bf16, D[i][n]: lane = column n, registers = 4 consecutive i,
// so a lane stores 4 consecutive i of one C column as one 8/16-byte access).
// The pattern arrives with the call, so the wave first inverts it: a wave-private LDS table
// tbl[n-block][k-block] = block id (or none) plus a bitmask of the k-blocks that matter.  The main loop then
// runs k-block OUTER: the A operand of a k-block (the only big stream: m_blocks*M*K*2 bytes) is fetched once,
// straight into MFMA operand registers (VNNI dwords, lanes along i), while the next one is already in flight;
// the B blocks (k contiguous: a 16 x 32 block is one contiguous 1 KiB load) stay L2 resident.
// [ref semantics: samples/xgemm_sparse/spmm_kernel.c:74-217]
// ------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short bf16x8v __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt2(float lo, float hi) { return bf16_pk_exact(lo, hi); }      // the reference's conversion exactly (bf16_cvt.hpp)
template <typename F, int... Is> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int kBcscTbl = 1024;     // table entries per wave (n-blocks per wave x k-blocks)
constexpr int kBcscTblDma = 256;   // same for the LDS-DMA kernel, which spends its LDS on the A ring instead

template <int BN16>                // bn / 16
__global__ __launch_bounds__(256) void bcsc_mfma_bf16_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int total) {
  constexpr int NBL = 4 / BN16;    // n-blocks per wave (64 columns)
  __shared__ unsigned int tbl_all[4][kBcscTbl];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  if (wid >= total) return;
  unsigned int* tbl = tbl_all[wave];
  const unsigned int tn = wid % tiles_n, tmp = wid / tiles_n, ti = tmp % tiles_i, mb = tmp / tiles_i;
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const int i0 = (int)ti * 64, n0 = (int)tn * 64;
  const int mt = (p.M - i0 >= 64) ? 4 : (p.M - i0) / 16;                  // i-tiles of this wave
  const int nbl_cnt = ((p.N - n0 >= 64) ? 64 : (p.N - n0)) / (16 * BN16);  // n-blocks of this wave
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = p.bk / 32;
  GM const unsigned int* colptr = (GM const unsigned int*)p.colptr;
  GM const unsigned int* rowidx = (GM const unsigned int*)p.rowidx;
  // ---- invert the pattern for this wave's columns ----
  for (int e = lane; e < nbl_cnt * nkb; e += 64) tbl[e] = 0xffffffffu;
  for (int nbl = 0; nbl < nbl_cnt; ++nbl) {
    const unsigned int c0 = colptr[nb0 + nbl], c1 = colptr[nb0 + nbl + 1];
    for (unsigned int b = c0 + lane; b < c1; b += 64) tbl[nbl * nkb + rowidx[b]] = b;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- accumulators ----
  f32x4v acc[4][4];                                                        // [n-tile][i-tile]
  const bool c_f32 = (p.c_type == LIBXSMM_DATATYPE_F32);
  GM char* cbase = (GM char*)p.c + ((long long)mb * p.N * p.M) * (c_f32 ? 4 : 2);
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    acc[nt][it] = (f32x4v)0.0f;
    if (!p.beta0 && it < mt && nt < nbl_cnt * BN16) {
      const long long e = (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg;
      if (c_f32) acc[nt][it] = *(GM const f32x4v*)(cbase + e * 4);
      else {
        const u32x2v v = *(GM const u32x2v*)(cbase + e * 2);
        acc[nt][it][0] = __uint_as_float(v[0] << 16); acc[nt][it][1] = __uint_as_float(v[0] & 0xffff0000u);
        acc[nt][it][2] = __uint_as_float(v[1] << 16); acc[nt][it][3] = __uint_as_float(v[1] & 0xffff0000u);
      }
    }
  });
  // ---- A operand of chunk (k-block kb, step): lane (row i = lx, k group kg) holds 8 consecutive k = 4 VNNI dwords ----
  GM const unsigned int* A2 = (GM const unsigned int*)p.a + (long long)mb * (p.K / 2) * p.M + i0 + lx;
  auto load_a = [&](u32x4v (&dst)[4], int kb, int st) {
    const long long kp0 = (long long)kb * (p.bk / 2) + 16 * st + 4 * kg;
    sfor<4>([&](auto tc) {
      constexpr int t = tc.value;
      if (t < mt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[t][e] = A2[(kp0 + e) * p.M + 16 * t];
      }
    });
  };
  GM const char* bv = (GM const char*)p.bvals;
  u32x4v a_cur[4], a_nxt[4];
  for (int kgp = 0; kgp < nkb; kgp += 64) {
    // k-blocks of this group that any of the wave's n-blocks uses
    bool used = false;
    const int kb_l = kgp + lane;
    if (kb_l < nkb) for (int nbl = 0; nbl < nbl_cnt; ++nbl) used = used || (tbl[nbl * nkb + kb_l] != 0xffffffffu);
    unsigned long long mask = __ballot(used);
    if (mask == 0ull) continue;
    int kb = kgp + (int)__builtin_ctzll(mask); mask &= mask - 1ull;
    int st = 0;
    load_a(a_cur, kb, 0);
    for (;;) {
      // next chunk: next step of this k-block, else the next used k-block
      int kb_n = kb, st_n = st + 1; bool more = true;
      if (st_n == steps) { st_n = 0; if (mask != 0ull) { kb_n = kgp + (int)__builtin_ctzll(mask); mask &= mask - 1ull; } else more = false; }
      if (more) load_a(a_nxt, kb_n, st_n);
      sfor<NBL>([&](auto nc) {
        constexpr int nbl = nc.value;
        if (nbl < nbl_cnt) {
          const unsigned int blk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tbl[nbl * nkb + kb]);
          if (blk != 0xffffffffu) {
            sfor<BN16>([&](auto sc) {
              constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
              const u32x4v bfrag = *(GM const u32x4v*)(bv + (((long long)blk * (16 * BN16) + 16 * s2 + lx) * p.bk + 32 * st + 8 * kg) * 2);
              sfor<4>([&](auto tc) {
                constexpr int t = tc.value;
                if (t < mt) acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a_cur[t]), __builtin_bit_cast(bf16x8v, bfrag), acc[nt][t], 0, 0, 0);
              });
            });
          }
        }
      });
      if (!more) break;
      sfor<4>([&](auto tc) { a_cur[tc.value] = a_nxt[tc.value]; });
      kb = kb_n; st = st_n;
    }
  }
  // ---- store: lane = column n, 4 consecutive i per accumulator ----
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    if (it < mt && nt < nbl_cnt * BN16) {
      const long long e = (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg;
      if (c_f32) *(GM f32x4v*)(cbase + e * 4) = acc[nt][it];
      else { u32x2v v; v[0] = cvt2(acc[nt][it][0], acc[nt][it][1]); v[1] = cvt2(acc[nt][it][2], acc[nt][it][3]); *(GM u32x2v*)(cbase + e * 2) = v; }
    }
  });
}

// ------------------------------------------------------------------------------------------------
// BCSC with 8-bit integer operands on the matrix cores (v_mfma_i32_16x16x32_i8): unsigned A x signed B or signed A x unsigned B
// -> int32, exact.  Structure of bcsc_mfma_bf16_kernel: one wave owns (M-block, 64 rows, 64 columns) of C in 64 accumulator VGPRs,
// inverts the pattern for its columns, then runs k-block outer with the A operand double-buffered in registers.  A is VNNI-4
// ([K/4][M][4]: a dword = four k of one row): lane (row lx, k group kg) of a 32-deep step takes the two dwords k = 8 kg .. 8 kg + 7;
// a B block is [bn][bk] bytes, k contiguous: lane (column lx, kg) takes the matching 8 bytes.
// The matrix core multiplies SIGNED bytes.  The unsigned operand u is fed as u ^ 0x80 = u - 128 and the missing 128 * sum_k(other
// operand) comes from one more MFMA per fragment against an all-ones operand, accumulated apart and added once at the end:
//   unsigned A: + 128 * sum_k b[k][n] (per column tile),   unsigned B: + 128 * sum_k a[i][k] over the k-blocks stored for that n-block.
// [ref semantics: samples/xgemm_sparse/spmm_kernel.c:153-217]
// ------------------------------------------------------------------------------------------------
typedef int i32x4v __attribute__((ext_vector_type(4)));
template <int BN16, bool UA>       // UA: A unsigned (B signed); else A signed, B unsigned
__global__ __launch_bounds__(256) void bcsc_mfma_i8_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int total) {
  constexpr int NBL = 4 / BN16;
  __shared__ unsigned int tbl_all[4][kBcscTbl];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  if (wid >= total) return;
  unsigned int* tbl = tbl_all[wave];
  const unsigned int tn = wid % tiles_n, tmp = wid / tiles_n, ti = tmp % tiles_i, mb = tmp / tiles_i;
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const int i0 = (int)ti * 64, n0 = (int)tn * 64;
  const int mt = (p.M - i0 >= 64) ? 4 : (p.M - i0) / 16;
  const int nbl_cnt = ((p.N - n0 >= 64) ? 64 : (p.N - n0)) / (16 * BN16);
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = p.bk / 32;
  GM const unsigned int* colptr = (GM const unsigned int*)p.colptr;
  GM const unsigned int* rowidx = (GM const unsigned int*)p.rowidx;
  for (int e = lane; e < nbl_cnt * nkb; e += 64) tbl[e] = 0xffffffffu;
  for (int nbl = 0; nbl < nbl_cnt; ++nbl) {
    const unsigned int c0 = colptr[nb0 + nbl], c1 = colptr[nb0 + nbl + 1];
    for (unsigned int b = c0 + lane; b < c1; b += 64) tbl[nbl * nkb + rowidx[b]] = b;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  i32x4v acc[4][4], corr_b[4], corr_a[NBL][4];       // corr_b[n-tile]: sum_k b (unsigned A); corr_a[n-block][i-tile]: sum_k a (unsigned B)
  GM int* cbase = (GM int*)p.c + (long long)mb * p.N * p.M;
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    acc[nt][it] = (i32x4v)0;
    if (!p.beta0 && it < mt && nt < nbl_cnt * BN16) acc[nt][it] = *(GM const i32x4v*)(cbase + (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg);
  });
  sfor<4>([&](auto c) { corr_b[c.value] = (i32x4v)0; });
  sfor<NBL * 4>([&](auto c) { corr_a[c.value / 4][c.value % 4] = (i32x4v)0; });
  const long long ones = 0x0101010101010101ll;
  GM const unsigned int* A4 = (GM const unsigned int*)p.a + (long long)mb * (p.K / 4) * p.M + i0 + lx;
  auto load_a = [&](long long (&dst)[4], int kb, int st) {
    const long long kq0 = (long long)kb * (p.bk / 4) + 8 * st + 2 * kg;
    sfor<4>([&](auto tc) {
      constexpr int t = tc.value;
      if (t < mt) {
        unsigned int lo, hi;
        if (p.nt_a) { lo = __builtin_nontemporal_load(A4 + kq0 * p.M + 16 * t); hi = __builtin_nontemporal_load(A4 + (kq0 + 1) * p.M + 16 * t); }
        else { lo = A4[kq0 * p.M + 16 * t]; hi = A4[(kq0 + 1) * p.M + 16 * t]; }
        if (UA) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
        dst[t] = (long long)(((unsigned long long)hi << 32) | lo);
      }
    });
  };
  GM const char* bv = (GM const char*)p.bvals;
  long long a_cur[4], a_nxt[4];
  for (int kgp = 0; kgp < nkb; kgp += 64) {
    bool used = false;
    const int kb_l = kgp + lane;
    if (kb_l < nkb) for (int nbl = 0; nbl < nbl_cnt; ++nbl) used = used || (tbl[nbl * nkb + kb_l] != 0xffffffffu);
    unsigned long long mask = __ballot(used);
    if (mask == 0ull) continue;
    int kb = kgp + (int)__builtin_ctzll(mask); mask &= mask - 1ull;
    int st = 0;
    load_a(a_cur, kb, 0);
    for (;;) {
      int kb_n = kb, st_n = st + 1; bool more = true;
      if (st_n == steps) { st_n = 0; if (mask != 0ull) { kb_n = kgp + (int)__builtin_ctzll(mask); mask &= mask - 1ull; } else more = false; }
      if (more) load_a(a_nxt, kb_n, st_n);
      sfor<NBL>([&](auto nc) {
        constexpr int nbl = nc.value;
        if (nbl < nbl_cnt) {
          const unsigned int blk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tbl[nbl * nkb + kb]);
          if (blk != 0xffffffffu) {
            sfor<BN16>([&](auto sc) {
              constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
              long long bfrag = *(GM const long long*)(bv + ((long long)blk * (16 * BN16) + 16 * s2 + lx) * p.bk + 32 * st + 8 * kg);
              if (!UA) bfrag ^= (long long)0x8080808080808080ull;
              sfor<4>([&](auto tc) {
                constexpr int t = tc.value;
                if (t < mt) acc[nt][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_cur[t], bfrag, acc[nt][t], 0, 0, 0);
              });
              if (UA) corr_b[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ones, bfrag, corr_b[nt], 0, 0, 0);
            });
            if (!UA) sfor<4>([&](auto tc) { constexpr int t = tc.value; if (t < mt) corr_a[nbl][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_cur[t], ones, corr_a[nbl][t], 0, 0, 0); });
          }
        }
      });
      if (!more) break;
      sfor<4>([&](auto tc) { a_cur[tc.value] = a_nxt[tc.value]; });
      kb = kb_n; st = st_n;
    }
  }
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    if (it < mt && nt < nbl_cnt * BN16) {
      i32x4v v = acc[nt][it];
      if (UA) v += corr_b[nt] * 128; else v += corr_a[nt / BN16][it] * 128;
      *(GM i32x4v*)(cbase + (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg) = v;
    }
  });
}

// ------------------------------------------------------------------------------------------------
// BCSC in f32 on the matrix cores (v_mfma_f32_16x16x4_f32), bk % 16 == 0, bn in {16, 32, 64}, M % 16 == 0.  Same tile ownership and
// pattern handling as the bf16 kernel.  A is [K][M] column-major (rows of M contiguous floats): lane (row lx, group kg) reads the four
// rows k = 4 kg + s of a 16-deep chunk (64-byte segments per lane group); a B block is [bn][bk] with k contiguous: lane (column lx,
// group kg) reads its four k = 4 kg .. 4 kg + 3 as ONE 16-byte load.  MFMA step s then multiplies k = 4 kg + s of every group: the
// k order inside a chunk is group-interleaved, which is a valid summation order (tolerance of the reference's f32 check), not bitwise
// the oracle's.  f32 matrix rate makes this HBM-bound: A is streamed once, B blocks stay in L2.
// ------------------------------------------------------------------------------------------------
template <int BN16>
__global__ __launch_bounds__(256) void bcsc_mfma_f32_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int total) {
  constexpr int NBL = 4 / BN16;
  __shared__ unsigned int tbl_all[4][kBcscTbl];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  if (wid >= total) return;
  unsigned int* tbl = tbl_all[wave];
  const unsigned int tn = wid % tiles_n, tmp = wid / tiles_n, ti = tmp % tiles_i, mb = tmp / tiles_i;
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const int i0 = (int)ti * 64, n0 = (int)tn * 64;
  const int mt = (p.M - i0 >= 64) ? 4 : (p.M - i0) / 16;
  const int nbl_cnt = ((p.N - n0 >= 64) ? 64 : (p.N - n0)) / (16 * BN16);
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = p.bk / 16;
  GM const unsigned int* colptr = (GM const unsigned int*)p.colptr;
  GM const unsigned int* rowidx = (GM const unsigned int*)p.rowidx;
  for (int e = lane; e < nbl_cnt * nkb; e += 64) tbl[e] = 0xffffffffu;
  for (int nbl = 0; nbl < nbl_cnt; ++nbl) {
    const unsigned int c0 = colptr[nb0 + nbl], c1 = colptr[nb0 + nbl + 1];
    for (unsigned int b = c0 + lane; b < c1; b += 64) tbl[nbl * nkb + rowidx[b]] = b;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f32x4v acc[4][4];
  GM float* cbase = (GM float*)p.c + (long long)mb * p.N * p.M;
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    acc[nt][it] = (f32x4v)0.0f;
    if (!p.beta0 && it < mt && nt < nbl_cnt * BN16) acc[nt][it] = *(GM const f32x4v*)(cbase + (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg);
  });
  GM const float* Ab = (GM const float*)p.a + (long long)mb * p.K * p.M + i0 + lx;
  const bool nta = p.nt_a != 0;                  // wave-uniform
  auto load_a = [&](f32x4v (&dst)[4], int kb, int st) {
    const long long k0 = (long long)kb * p.bk + 16 * st + 4 * kg;
    sfor<4>([&](auto tc) {
      constexpr int t = tc.value;
      if (t < mt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[t][e] = nta ? __builtin_nontemporal_load(Ab + (k0 + e) * p.M + 16 * t) : Ab[(k0 + e) * p.M + 16 * t];
      }
    });
  };
  GM const float* bv = (GM const float*)p.bvals;
  f32x4v a_cur[4], a_nxt[4];
  for (int kgp = 0; kgp < nkb; kgp += 64) {
    bool used = false;
    const int kb_l = kgp + lane;
    if (kb_l < nkb) for (int nbl = 0; nbl < nbl_cnt; ++nbl) used = used || (tbl[nbl * nkb + kb_l] != 0xffffffffu);
    unsigned long long mask = __ballot(used);
    if (mask == 0ull) continue;
    int kb = kgp + (int)__builtin_ctzll(mask); mask &= mask - 1ull;
    int st = 0;
    load_a(a_cur, kb, 0);
    for (;;) {
      int kb_n = kb, st_n = st + 1; bool more = true;
      if (st_n == steps) { st_n = 0; if (mask != 0ull) { kb_n = kgp + (int)__builtin_ctzll(mask); mask &= mask - 1ull; } else more = false; }
      if (more) load_a(a_nxt, kb_n, st_n);
      sfor<NBL>([&](auto nc) {
        constexpr int nbl = nc.value;
        if (nbl < nbl_cnt) {
          const unsigned int blk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tbl[nbl * nkb + kb]);
          if (blk != 0xffffffffu) {
            sfor<BN16>([&](auto sc) {
              constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
              const f32x4v bfrag = *(GM const f32x4v*)(bv + ((long long)blk * (16 * BN16) + 16 * s2 + lx) * p.bk + 16 * st + 4 * kg);
              sfor<4>([&](auto tc) {
                constexpr int t = tc.value;
                if (t < mt) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t][e], bfrag[e], acc[nt][t], 0, 0, 0);
                }
              });
            });
          }
        }
      });
      if (!more) break;
      sfor<4>([&](auto tc) { a_cur[tc.value] = a_nxt[tc.value]; });
      kb = kb_n; st = st_n;
    }
  }
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    if (it < mt && nt < nbl_cnt * BN16) *(GM f32x4v*)(cbase + (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg) = acc[nt][it];
  });
}

// Inverts the BCSC pattern once per call: table[n-block][k-block] = block id (0xffffffff: none).  Every wave of the main
// kernel needs the same rows of it; building it there costs each wave three dependent global round trips up front.
__global__ void bcsc_invert_kernel(const unsigned int* colptr_, const unsigned int* rowidx_, unsigned int* table, int nblk_n, int nkb) {
  GM const unsigned int* colptr = (GM const unsigned int*)colptr_; GM const unsigned int* rowidx = (GM const unsigned int*)rowidx_;
  GM unsigned int* t = (GM unsigned int*)table;
  const int nb = blockIdx.x;
  if (nb >= nblk_n) return;
  for (int e = threadIdx.x; e < nkb; e += blockDim.x) t[(long long)nb * nkb + e] = 0xffffffffu;
  __syncthreads();
  // (a k-block id beyond K / bk in a device-resident pattern is dropped instead of written past the table: the host-pattern path refuses such patterns up front)
  for (unsigned int b = colptr[nb] + threadIdx.x; b < colptr[nb + 1]; b += blockDim.x) { const unsigned int kb = rowidx[b]; if (kb < (unsigned int)nkb) t[(long long)nb * nkb + kb] = b; }
}

int launch_bcsc_invert(const unsigned int* colptr, const unsigned int* rowidx, unsigned int* table, int nblk_n, int nkb, void* stream) {
  hipLaunchKernelGGL(bcsc_invert_kernel, dim3((unsigned int)nblk_n), dim3(64), 0, (hipStream_t)stream, colptr, rowidx, table, nblk_n, nkb);
  return (int)hipGetLastError();
}

// Same algorithm with the A operand staged by LDS-DMA: a k-chunk of A for the wave's 64 rows ([16 k-pairs][64 i] dwords,
// 4 KiB) arrives with four fully coalesced global_load_lds_dwordx4 (whole 256-byte rows instead of 64-byte dword
// segments), the MFMA operand is then read with conflict-free ds_read_b32 (rows of odd k-groups are rotated by 16 words
// on the SOURCE side so that the two k-groups of a half-wave hit different banks).  Per chunk: wait, read the operand
// registers, fetch the B fragments of this chunk, start the DMA of the NEXT chunk into the same image, run the MFMAs.
template <int BN16, int AUX_A = 0>      // AUX_A: cache-policy bits of the A stream (2 = nt when the launch is larger than the Infinity Cache)
__global__ __launch_bounds__(256) void bcsc_mfma_bf16_dma_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int total, const unsigned int* gtable) {
  constexpr int NBL = 4 / BN16;
  __shared__ unsigned int tbl_all[4][kBcscTblDma];
  __shared__ __attribute__((aligned(16))) unsigned int abuf_all[4][3][1024];     // ring of three 4 KiB chunk images per wave
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  if (wid >= total) return;
  unsigned int* tbl = tbl_all[wave];
  unsigned int (*abuf)[1024] = abuf_all[wave];
  const unsigned int tn = wid % tiles_n, tmp = wid / tiles_n, ti = tmp % tiles_i, mb = tmp / tiles_i;
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const int i0 = (int)ti * 64, n0 = (int)tn * 64;
  const int mt = (p.M - i0 >= 64) ? 4 : (p.M - i0) / 16;
  const int nbl_cnt = ((p.N - n0 >= 64) ? 64 : (p.N - n0)) / (16 * BN16);
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = p.bk / 32;
  {   // this wave's rows of the inverted pattern: one coalesced read
    GM const unsigned int* gt = (GM const unsigned int*)gtable + (long long)nb0 * nkb;
    for (int e = lane; e < nbl_cnt * nkb; e += 64) tbl[e] = gt[e];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  f32x4v acc[4][4];
  const bool c_f32 = (p.c_type == LIBXSMM_DATATYPE_F32);
  GM char* cbase = (GM char*)p.c + ((long long)mb * p.N * p.M) * (c_f32 ? 4 : 2);
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    acc[nt][it] = (f32x4v)0.0f;
    if (!p.beta0 && it < mt && nt < nbl_cnt * BN16) {
      const long long e = (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg;
      if (c_f32) acc[nt][it] = *(GM const f32x4v*)(cbase + e * 4);
      else {
        const u32x2v v = *(GM const u32x2v*)(cbase + e * 2);
        acc[nt][it][0] = __uint_as_float(v[0] << 16); acc[nt][it][1] = __uint_as_float(v[0] & 0xffff0000u);
        acc[nt][it][2] = __uint_as_float(v[1] << 16); acc[nt][it][3] = __uint_as_float(v[1] & 0xffff0000u);
      }
    }
  });
  // DMA source of LDS slot (lane + 64x): row kp_l = slot >> 4, the 16-byte group that lands there = (slot & 15) rotated back
  GM const unsigned int* A2 = (GM const unsigned int*)p.a + (long long)mb * (p.K / 2) * p.M + i0;
  unsigned int src_off[4]; bool src_ok[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const unsigned int S = (unsigned int)lane + 64u * x, kp_l = S >> 4, g = ((S & 15u) - 4u * ((kp_l >> 2) & 1u)) & 15u;
    src_ok[x] = (int)(4u * g) < 16 * mt;                                  // groups beyond the wave's rows re-read group 0 (never consumed):
    src_off[x] = kp_l * (unsigned int)p.M + (src_ok[x] ? 4u * g : 0u);    // every DMA instruction always issues -> the wait below can count
  }
  auto issue_a = [&](int kb, int st, int slot) {
    GM const unsigned int* rowbase = A2 + ((long long)kb * (p.bk / 2) + 16 * st) * p.M;
#pragma unroll
    for (int x = 0; x < 4; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(rowbase + src_off[x]), (lds_ptr_t)((char*)abuf[slot] + 1024 * x), 16, 0, AUX_A);
  };
  // chunk sequence of this k-group: (k-block, step) pairs in order; `next_chunk` advances a cursor
  auto next_chunk = [&](int& kb, int& st, unsigned long long& m, int kgp) -> bool {
    if (st + 1 < steps) { ++st; return true; }
    if (m == 0ull) return false;
    st = 0; kb = kgp + (int)__builtin_ctzll(m); m &= m - 1ull;
    return true;
  };
  // operand read: lane (row i = 16t + lx, k group kg) takes dwords of rows 4kg + e at word (i + 16*(kg & 1)) % 64
  const int rot = 16 * (kg & 1);
  GM const char* bv = (GM const char*)p.bvals;
  for (int kgp = 0; kgp < nkb; kgp += 64) {
    bool used = false;
    const int kb_l = kgp + lane;
    if (kb_l < nkb) for (int nbl = 0; nbl < nbl_cnt; ++nbl) used = used || (tbl[nbl * nkb + kb_l] != 0xffffffffu);
    unsigned long long mask = __ballot(used);
    if (mask == 0ull) continue;
    // cursors: c0 = chunk being consumed, c1 = c0 + 1, c2 = c0 + 2 (the DMA runs two chunks ahead, the B fragments one)
    int kb0 = kgp + (int)__builtin_ctzll(mask), st0 = 0; mask &= mask - 1ull;
    int kb1 = kb0, st1 = st0; unsigned long long m1 = mask; const bool has1 = next_chunk(kb1, st1, m1, kgp);
    int kb2 = kb1, st2 = st1; unsigned long long m2 = m1; bool has2 = has1 && next_chunk(kb2, st2, m2, kgp);
    bool more1 = has1;
    unsigned int blk_cur[NBL], blk_nxt[NBL]; u32x4v bf_cur[NBL][BN16], bf_nxt[NBL][BN16];
    auto fetch_b = [&](unsigned int (&blk)[NBL], u32x4v (&bf)[NBL][BN16], int kb_, int st_) {
      sfor<NBL>([&](auto nc) {
        constexpr int nbl = nc.value;
        blk[nbl] = 0xffffffffu;
        if (nbl < nbl_cnt) {
          blk[nbl] = (unsigned int)__builtin_amdgcn_readfirstlane((int)tbl[nbl * nkb + kb_]);
          if (blk[nbl] != 0xffffffffu)
            sfor<BN16>([&](auto sc) { constexpr int s2 = sc.value;
              bf[nbl][s2] = *(GM const u32x4v*)(bv + (((long long)blk[nbl] * (16 * BN16) + 16 * s2 + lx) * p.bk + 32 * st_ + 8 * kg) * 2); });
        }
      });
    };
    int slot = 0;
    issue_a(kb0, st0, 0);
    fetch_b(blk_cur, bf_cur, kb0, st0);
    if (more1) issue_a(kb1, st1, 1);
    for (;;) {
      // chunk c0 (image `slot`) and its B fragments must have landed; the 4 DMA instructions of chunk c1 may stay in flight
      if (more1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      u32x4v a_cur[4];
      sfor<4>([&](auto tc) {
        constexpr int t = tc.value;
        if (t < mt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a_cur[t][e] = abuf[slot][(4 * kg + e) * 64 + ((16 * t + lx + rot) & 63)];
        }
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (more1) fetch_b(blk_nxt, bf_nxt, kb1, st1);                              // B of c1: older than the DMA of c2 queued next
      if (has2) issue_a(kb2, st2, (slot + 2) % 3);                                // image of c0 - 1, whose reads retired last iteration
      sfor<NBL>([&](auto nc) {
        constexpr int nbl = nc.value;
        if (nbl < nbl_cnt && blk_cur[nbl] != 0xffffffffu) {
          sfor<BN16>([&](auto sc) {
            constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
            sfor<4>([&](auto tc) {
              constexpr int t = tc.value;
              if (t < mt) acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a_cur[t]), __builtin_bit_cast(bf16x8v, bf_cur[nbl][s2]), acc[nt][t], 0, 0, 0);
            });
          });
        }
      });
      if (!more1) break;
      sfor<NBL>([&](auto nc) { constexpr int nbl = nc.value; blk_cur[nbl] = blk_nxt[nbl]; sfor<BN16>([&](auto sc) { bf_cur[nbl][sc.value] = bf_nxt[nbl][sc.value]; }); });
      slot = (slot + 1) % 3;
      kb1 = kb2; st1 = st2; more1 = has2;
      if (has2) has2 = next_chunk(kb2, st2, m2, kgp);
    }
  }
  if (!c_f32 && mt == 4 && nbl_cnt * BN16 == 4 && (p.M % 8) == 0 && ((((size_t)cbase) & 15) == 0)) {
    // Full 64 x 64 bf16 tile: a lane holds 4 consecutive rows of one column, so a direct store writes 32-byte pieces of sixteen 128-byte
    // lines.  Pass the tile through the (now idle) A ring instead -- 8-byte slots XOR-swizzled by the column so that neither side has
    // bank conflicts -- and write every column's 64 rows as one full line: 8 lanes x 16 bytes.
    unsigned int* tile = &abuf[0][0];                                               // 64 columns x 128 bytes
    sfor<16>([&](auto ic) {
      constexpr int nt = ic.value / 4, it = ic.value % 4;
      const int n = 16 * nt + lx;
      u32x2v v; v[0] = cvt2(acc[nt][it][0], acc[nt][it][1]); v[1] = cvt2(acc[nt][it][2], acc[nt][it][3]);
      *(u32x2v*)(tile + n * 32 + 2 * ((4 * it + kg) ^ (n & 15))) = v;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int n = 8 * r + (lane >> 3), j = lane & 7, x = n & 15;
      u32x4v w = *(const u32x4v*)(tile + n * 32 + 4 * (j ^ (x >> 1)));
      if (x & 1) { const unsigned int t0 = w[0], t1 = w[1]; w[0] = w[2]; w[1] = w[3]; w[2] = t0; w[3] = t1; }
      *(GM u32x4v*)(cbase + ((long long)(n0 + n) * p.M + i0) * 2 + 16 * j) = w;
    }
    return;
  }
  sfor<16>([&](auto ic) {
    constexpr int nt = ic.value / 4, it = ic.value % 4;
    if (it < mt && nt < nbl_cnt * BN16) {
      const long long e = (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg;
      if (c_f32) *(GM f32x4v*)(cbase + e * 4) = acc[nt][it];
      else { u32x2v v; v[0] = cvt2(acc[nt][it][0], acc[nt][it][1]); v[1] = cvt2(acc[nt][it][2], acc[nt][it][3]); *(GM u32x2v*)(cbase + e * 2) = v; }
    }
  });
}

// bf16 BCSC with waves that STREAM over M-blocks.  The one-tile-per-wave kernels above spend most of a wave's life in its set-up chain (pattern
// rows -> first DMA -> first operands; profiles/r02_bcsc_counters.txt: 51 % of the wave cycles wait on a memory counter) because a tile is only
// ~5 chunks of work.  Here a wave keeps its (i-tile, n-tile), reads the pattern rows and compacts the used k-blocks ONCE, and then walks the
// M-blocks g, g + MBG, g + 2 MBG, ... as one flat sequence of chunks: the LDS-DMA ring and the B-fragment register ring run ahead across tile
// boundaries, so the next tile's first operands are in flight while this tile's C leaves (through an LDS image of its own, not the ring).
// s_waitcnt counts loads and stores in issue order: the wait before a chunk allows for the younger chunk and for the stores of a tile that
// ended since the chunk was issued (only the 8-store LDS epilogue is counted; the direct epilogue just makes the next waits stricter).
// beta = 0 only (a C read would sit in the middle of the counted sequence); one k-group (K / bk <= 64).
// F32 (round 3): the same kernel on f32 operands.  A chunk is then 16 k (not 16 k-PAIRS) of the wave's rows -- the LDS image, its rotation and the
// operand reads are word for word the bf16 ones, a row of the image is one k instead of a VNNI pair --, a B fragment is four consecutive k of a column
// (again one 16-byte load) and a chunk is four v_mfma_f32_16x16x4_f32 per (n-tile, i-tile), k = 4 kg + e in step e as in bcsc_mfma_f32_kernel.
constexpr int kBcscBLds = 10240;     // bytes of LDS for a copy of the whole B value array (BL): 2 : 8 of 256 x 64 in bf16 is 8 KiB
template <int BN16, int AUX_A = 0, int RT = 4, int WPS = 2, bool F32 = false, bool BL = false>     // RT: 16-row tiles per wave (4: 64 rows, 2: 32 rows -> half the accumulators, more waves per SIMD)
__global__ __launch_bounds__(256, WPS) void bcsc_mfma_bf16_stream_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int mbg, unsigned int total_waves, const unsigned int* gtable) {
  // D: depth of the B-fragment REGISTER ring (the chunk being consumed + one behind it); DA: depth of the A ring in LDS.  Round 6: DA = 3 -- the A chunks (HBM) run one
  // chunk further ahead than the B fragments (L2), 8 KiB instead of 4 KiB of A in flight per wave behind the chunk being consumed, with the register budget unchanged
  // (a third B slot would be 16 registers the two-waves-per-SIMD budget does not have)
  constexpr int NBL = 4 / BN16, D = 2, DA = 3, W = 16 * RT, SPR = 4 * RT, NI = RT;     // W words per image row, SPR 16-byte slots per row, NI DMA instructions per chunk
  __shared__ unsigned int tbl_all[4][kBcscTblDma];
  __shared__ unsigned int klist_all[4][64];
  __shared__ __attribute__((aligned(16))) unsigned int abuf_all[4][DA][16 * W];
  __shared__ __attribute__((aligned(16))) unsigned int ctile_all[4][F32 ? 4 : 1024];       // 32 columns x 128 bytes: bf16 C leaves in two halves
  __shared__ __attribute__((aligned(16))) unsigned int bimg[BL ? kBcscBLds / 4 : 4];       // BL: the whole bf16 value array of B, read by every wave of the workgroup
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  if constexpr (BL) {          // every wave of the workgroup helps to bring B in and meets the others before any of them leaves
    const unsigned int pieces = (unsigned int)p.nnzb * (unsigned int)(p.bn * p.bk) / 8u;        // 16-byte pieces of the bf16 value array
    for (unsigned int e = threadIdx.x; e < pieces; e += 256u) ((u32x4v*)bimg)[e] = ((GM const u32x4v*)p.bvals)[e];
    __syncthreads();
  }
  if (wid >= total_waves) return;
  unsigned int* tbl = tbl_all[wave];
  unsigned int* klist = klist_all[wave];
  unsigned int (*abuf)[16 * W] = abuf_all[wave];
  unsigned int* tile = ctile_all[wave];
  const unsigned int tt_count = tiles_i * tiles_n, tt = wid % tt_count, g0 = wid / tt_count;
  const unsigned int tn = tt % tiles_n, ti = tt / tiles_n;
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const int i0 = (int)ti * (16 * RT), n0 = (int)tn * 64;
  const int mt = (p.M - i0 >= 16 * RT) ? RT : (p.M - i0) / 16;
  const int nbl_cnt = ((p.N - n0 >= 64) ? 64 : (p.N - n0)) / (16 * BN16);
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = F32 ? p.bk / 16 : p.bk / 32;
  {
    GM const unsigned int* gt = (GM const unsigned int*)gtable + (long long)nb0 * nkb;
    for (int e = lane; e < nbl_cnt * nkb; e += 64) tbl[e] = gt[e];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  bool used = false;
  if (lane < nkb) for (int nbl = 0; nbl < nbl_cnt; ++nbl) used = used || (tbl[nbl * nkb + lane] != 0xffffffffu);
  const unsigned long long mask = __ballot(used);
  if (used) klist[__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (unsigned int)lane;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int nch = __builtin_popcountll(mask) * steps;
  const int nmb = ((unsigned int)p.m_blocks > g0) ? (int)(((unsigned int)p.m_blocks - g0 + mbg - 1u) / mbg) : 0;
  const bool c_f32 = F32 || (p.c_type == LIBXSMM_DATATYPE_F32);
  const long long c_mb_bytes = (long long)p.N * p.M * (c_f32 ? 4 : 2);
  const bool lds_store = !F32 && !c_f32 && mt == RT && nbl_cnt * BN16 == 4 && (p.M % 8) == 0 && ((((size_t)p.c) & 15) == 0);
  f32x4v acc[4][RT];
  sfor<4 * RT>([&](auto ic) { acc[ic.value / RT][ic.value % RT] = (f32x4v)0.0f; });
  auto store_tile = [&](unsigned int mb) __attribute__((always_inline)) {        // C of M-block mb leaves; the accumulators restart at zero
    GM char* cbase = (GM char*)p.c + (long long)mb * c_mb_bytes;
    if (lds_store) {
      if constexpr (RT == 4) {
      sfor<2>([&](auto hc) {
        constexpr int h = hc.value;
        sfor<8>([&](auto ic) {
          constexpr int nt = 2 * h + ic.value / 4, it = ic.value % 4;
          const int n = 16 * (nt - 2 * h) + lx;
          u32x2v v; v[0] = cvt2(acc[nt][it][0], acc[nt][it][1]); v[1] = cvt2(acc[nt][it][2], acc[nt][it][3]);
          *(u32x2v*)(tile + n * 32 + 2 * ((4 * it + kg) ^ (n & 15))) = v;
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = 8 * r + (lane >> 3), j = lane & 7, x = n & 15;
          u32x4v w = *(const u32x4v*)(tile + n * 32 + 4 * (j ^ (x >> 1)));
          if (x & 1) { const unsigned int t0 = w[0], t1 = w[1]; w[0] = w[2]; w[1] = w[3]; w[2] = t0; w[3] = t1; }
          *(GM u32x4v*)(cbase + ((long long)(n0 + 32 * h + n) * p.M + i0) * 2 + 16 * j) = w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      });
      } else {           // 32 rows: a column is 64 bytes, four lanes write it; one pass over all 64 columns
        sfor<8>([&](auto ic) {
          constexpr int nt = ic.value / 2, it = ic.value % 2;
          const int n = 16 * nt + lx;
          u32x2v v; v[0] = cvt2(acc[nt][it][0], acc[nt][it][1]); v[1] = cvt2(acc[nt][it][2], acc[nt][it][3]);
          *(u32x2v*)(tile + n * 16 + 2 * ((4 * it + kg) ^ (n & 7))) = v;
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = 16 * r + (lane >> 2), j = lane & 3, x = n & 7;
          u32x4v w = *(const u32x4v*)(tile + n * 16 + 4 * (j ^ (x >> 1)));
          if (x & 1) { const unsigned int t0 = w[0], t1 = w[1]; w[0] = w[2]; w[1] = w[3]; w[2] = t0; w[3] = t1; }
          *(GM u32x4v*)(cbase + ((long long)(n0 + n) * p.M + i0) * 2 + 16 * j) = w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
      sfor<4 * RT>([&](auto ic) {
        constexpr int nt = ic.value / RT, it = ic.value % RT;
        if (it < mt && nt < nbl_cnt * BN16) {
          const long long e = (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg;
          if (c_f32) *(GM f32x4v*)(cbase + e * 4) = acc[nt][it];
          else { u32x2v v; v[0] = cvt2(acc[nt][it][0], acc[nt][it][1]); v[1] = cvt2(acc[nt][it][2], acc[nt][it][3]); *(GM u32x2v*)(cbase + e * 2) = v; }
        }
      });
    }
    sfor<4 * RT>([&](auto ic) { acc[ic.value / RT][ic.value % RT] = (f32x4v)0.0f; });
  };
  if (nch == 0) { for (int j = 0; j < nmb; ++j) store_tile(g0 + (unsigned int)j * mbg); return; }     // no block in these columns: C = 0
  // DMA source of LDS slot (lane + 64x): row kp_l = slot >> 4, the 16-byte group that lands there = (slot & 15) rotated back
  GM const unsigned int* A2 = (GM const unsigned int*)p.a + i0;
  const long long a_mb_words = (long long)(F32 ? p.K : p.K / 2) * p.M;
  unsigned int src_off[NI];
#pragma unroll
  for (int x = 0; x < NI; ++x) {
    const unsigned int S = (unsigned int)lane + 64u * x, kp_l = S / SPR, g = ((S % SPR) - 4u * ((kp_l >> 2) & 1u)) & (unsigned int)(SPR - 1);
    src_off[x] = kp_l * (unsigned int)p.M + (((int)(4u * g) < 16 * mt) ? 4u * g : 0u);
  }
  const int rot = 16 * (kg & 1);
  GM const char* bv = (GM const char*)p.bvals;
  const int total_f = nmb * nch;
  unsigned int blk_r[D][NBL]; u32x4v bf_r[D][NBL][BN16];
  // flat chunk f = (tile j, chunk c).  Two cursors: the B fragments of chunk f + 2 are asked for when chunk f is consumed (register slot f % 2), the A chunk f + 3 at the
  // same point (LDS slot f % 3): in issue order  ... B(f) A(f+1) | B(f+1) A(f+2) | B(f+2) A(f+3) ...
  int bj = 0, bc_ = 0, aj = 0, ac_ = 0;             // the NEXT chunk whose B fragments / whose A chunk is to be issued
  auto issue_b = [&](auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    const int q = bc_ / steps, st_ = bc_ - q * steps;
    const int kb_ = __builtin_amdgcn_readfirstlane((int)klist[q]);
    sfor<NBL>([&](auto nc) {
      constexpr int nbl = nc.value;
      blk_r[u][nbl] = (nbl < nbl_cnt) ? (unsigned int)__builtin_amdgcn_readfirstlane((int)tbl[nbl * nkb + kb_]) : 0xffffffffu;
      sfor<BN16>([&](auto sc) { constexpr int s2 = sc.value;
        GM const char* src = (blk_r[u][nbl] == 0xffffffffu) ? bv :
          F32 ? bv + (((long long)blk_r[u][nbl] * (16 * BN16) + 16 * s2 + lx) * p.bk + 16 * st_ + 4 * kg) * 4
              : bv + (((long long)blk_r[u][nbl] * (16 * BN16) + 16 * s2 + lx) * p.bk + 32 * st_ + 8 * kg) * 2;
        bf_r[u][nbl][s2] = *(GM const u32x4v*)src; });
    });
    (void)bj;
    if (++bc_ == nch) { bc_ = 0; ++bj; }
  };
  auto issue_a = [&](auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    const int q = ac_ / steps, st_ = ac_ - q * steps;
    const int kb_ = __builtin_amdgcn_readfirstlane((int)klist[q]);
    const unsigned int mb_a = g0 + (unsigned int)aj * mbg;
    GM const unsigned int* rowbase = A2 + (long long)mb_a * a_mb_words + ((long long)kb_ * (F32 ? p.bk : p.bk / 2) + 16 * st_) * p.M;
#pragma unroll
    for (int x = 0; x < NI; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(rowbase + src_off[x]), (lds_ptr_t)((char*)abuf[u] + 1024 * x), 16, 0, AUX_A);
    if (++ac_ == nch) { ac_ = 0; ++aj; }
  };
  // prologue, in the steady state's order: A(0) | B(0) A(1) | B(1) A(2)
  if (total_f > 0) issue_a(std::integral_constant<int, 0>{});
  if (!BL && total_f > 0) issue_b(std::integral_constant<int, 0>{});
  if (total_f > 1) issue_a(std::integral_constant<int, 1>{});
  if (!BL && total_f > 1) issue_b(std::integral_constant<int, 1>{});
  if (total_f > 2) issue_a(std::integral_constant<int, 2>{});
  int cj = 0, cc = 0;                               // the chunk being consumed
  for (int f0 = 0; f0 < total_f; f0 += 6) {
    sfor<6>([&](auto uc) {
      constexpr int u = uc.value % D, ua = uc.value % DA;           // register slot of the chunk's B fragments, LDS slot of its A chunk
      const int f = f0 + uc.value;
      if (f < total_f) {
        // chunk f must have landed: its B fragments are the younger of its two requests, so everything issued BEHIND B(f) may still be in flight -- A(f+1), B(f+1), A(f+2)
        // (as far as those chunks exist) and, when a tile ended after B(f) was issued (at the end of chunk f - 1 or f - 2: this chunk is the first or second of its tile),
        // the 8 stores of that tile's C (loads and stores retire this counter in issue order on gfx9).  Fewer stores counted than outstanding (several one-chunk
        // tiles) only makes the wait stricter.
        static_assert(D == 2 && DA == 3, "the wait table below is written for B one chunk and A two chunks in flight behind the consumed one");
        constexpr int NS = (RT == 4) ? 8 : 4;           // stores of one tile's LDS epilogue
        // (the 16 direct stores of an f32 tile are NOT counted: the wait is then stricter than needed -- measured equal on config #4's shape in f32 -- and does not
        // depend on how stores retire relative to the loads around them)
        // BL: no B requests -- the chunk's own A request is what is waited for (A(f+1), A(f+2) behind it), and a tile's stores are younger than it for the first THREE chunks
        // of the next tile (A(f) was issued three chunks earlier, in front of the stores at the end of chunk f - 3)
        constexpr int PB = BL ? 0 : 4;
        const int left = total_f - 1 - f;
        const bool stored = lds_store && cj > 0 && cc < (BL ? 3 : 2);
        if (left >= 2) { if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PB + 2 * NI + NS) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PB + 2 * NI) : "memory"); }
        else if (left == 1) { if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PB + NI + NS) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PB + NI) : "memory"); }
        else if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4v a_cur[RT];
        sfor<RT>([&](auto tc) {
          constexpr int t = tc.value;
          if (t < mt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a_cur[t][e] = abuf[ua][(4 * kg + e) * W + ((16 * t + lx + rot) & (W - 1))];
          }
        });
        unsigned int blk_c[NBL]; u32x4v bf_c[NBL][BN16];
        if constexpr (BL) {          // the chunk's B fragments out of the LDS copy of the value array: chunk cc of a tile = 32-deep step cc % steps of the used k-block number cc / steps
          const int q = cc / steps, st_ = cc - q * steps;
          const int kb_ = __builtin_amdgcn_readfirstlane((int)klist[q]);
          sfor<NBL>([&](auto nc) {
            constexpr int nbl = nc.value;
            blk_c[nbl] = (nbl < nbl_cnt) ? (unsigned int)__builtin_amdgcn_readfirstlane((int)tbl[nbl * nkb + kb_]) : 0xffffffffu;
            sfor<BN16>([&](auto sc) { constexpr int s2 = sc.value;
              const unsigned int b = blk_c[nbl] == 0xffffffffu ? 0u : blk_c[nbl];
              bf_c[nbl][s2] = *(const u32x4v*)((const char*)bimg + ((b * (16u * BN16) + 16u * s2 + (unsigned int)lx) * (unsigned int)p.bk + 32u * (unsigned int)st_ + 8u * (unsigned int)kg) * 2u); });
          });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!BL) {
          sfor<NBL>([&](auto nc) { blk_c[nc.value] = blk_r[u][nc.value]; sfor<BN16>([&](auto sc) { bf_c[nc.value][sc.value] = bf_r[u][nc.value][sc.value]; }); });
          if (f + 2 < total_f) issue_b(std::integral_constant<int, u>{});
        }
        if (f + 3 < total_f) issue_a(std::integral_constant<int, ua>{});
        sfor<NBL>([&](auto nc) {
          constexpr int nbl = nc.value;
          if (blk_c[nbl] != 0xffffffffu) {
            sfor<BN16>([&](auto sc) {
              constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
              sfor<RT>([&](auto tc) {
                constexpr int t = tc.value;
                if (t < mt) {
                  if constexpr (F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {     // (__builtin_bit_cast on a vector ELEMENT reads element 0 whatever e is -- hipcc 7.2: go through scalars)
                      const unsigned int av = a_cur[t][e], bw = bf_c[nbl][s2][e];
                      acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av), __uint_as_float(bw), acc[nt][t], 0, 0, 0);
                    }
                  } else acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a_cur[t]), __builtin_bit_cast(bf16x8v, bf_c[nbl][s2]), acc[nt][t], 0, 0, 0);
                }
              });
            });
          }
        });
        if (++cc == nch) { store_tile(g0 + (unsigned int)cj * mbg); cc = 0; ++cj; }
      }
    });
  }
}

// The streaming kernel re-cut for the shape config #4 has: whole 64 x 64 tiles (M and N multiples of 64), bf16 C, the value array of B in LDS.  What bounded the
// general kernel above was not its operand stream but the chain of dependent steps in front of every chunk -- two integer divisions, five table look-ups in LDS each
// waited for, a branch around every MFMA for tiles that are not whole (with A served from cache -- every M-block reading the first eight's -- it still took 38 of its 61 us: profiles/r06_bcsc_full.jsonl, tag a_alias).  Here
// a wave writes ONE 16-byte record per chunk of its pattern before the loop -- the chunk's offset inside an M-block of A and the LDS offsets of its (up to four) blocks
// of B, 0xffff for an absent one -- and a chunk is: wait for its A; one batch of LDS reads (A fragments, B fragments, the next chunk's record, the offset of the chunk
// to request) and one wait; four LDS-DMA requests into the slot just read; the MFMAs.  The ring slot is a run-time index, so the loop is not unrolled over slots.
// The epilogue's LDS traffic is written as instructions: a ds_write / ds_read the compiler can see gets its s_waitcnt vmcnt(0) -- it cannot tell the C tile from
// the ring the requests in flight write to -- which would wait for the next tile's first three chunks at every tile end.
constexpr int kBcscRecs = 64;        // chunk records per wave (ring depth 3; depth 2: half)
// DA: depth of the A ring.  3: two workgroups per CU (78 KiB each).  2: THREE workgroups per CU -- ring 32 KiB, C leaving in quarters through 2 KiB per wave, B up to 8 KiB,
// 32 records: 50 KiB -- the same 96 KiB of A in flight per CU spread over twelve waves instead of eight (what a wave does between its waits hides behind two others).
// (EARLY stays a template parameter: as a run-time branch the two wait statements -- each naming the loaded registers -- made the compiler copy those registers at the
//  branch, BEFORE the wait, i.e. before the loads had landed: wrong results, caught by the parity tests; profiles/r06_bcsc_full.jsonl has no entry for it.)
// F32: f32 operands and f32 C on v_mfma_f32_16x16x4_f32 -- a chunk is 16 k of 64 rows (the same 4 KiB image, a row is one k instead of a k pair), B up to 16 KiB in LDS,
// C leaves through LDS eight columns at a time (2 KiB per wave: what is left next to two workgroups' rings)
template <int BN16, int AUX_A, bool EARLY, int DA, bool F32 = false>     // EARLY: one n-tile and the host's mask of its used k-blocks (BcscArgs::kmask0) -- the first three chunks are requested before anything is loaded
__global__ __launch_bounds__(256, DA == 2 ? 3 : 2) void bcsc_mfma_bf16_stream_full_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int mbg, unsigned int total_waves, const unsigned int* gtable) {
  static_assert(DA >= 2 && DA <= 4, "ring depth 2, 3 or 4");
  static_assert(!(F32 && DA != 3), "the f32 form has one LDS plan");
  constexpr int NBL = 4 / BN16, NI = 4, NS = F32 ? 16 : 8;          // NS: stores of one tile
  // DA = 4 (bf16): the ring takes 64 KiB of the workgroup's 78, so C leaves in eight passes of eight columns (1 KiB per wave), B up to 8 KiB, 32 records
  constexpr int RECS = DA == 3 ? kBcscRecs : kBcscRecs / 2, BLDS = F32 ? 16384 : (DA == 3 ? kBcscBLds : 8192), CPASS = DA == 2 ? 4 : (DA == 4 ? 8 : 2), TILEW = F32 ? 512 : 2048 / CPASS;     // C leaves in CPASS passes of 64 / CPASS columns
  __shared__ __attribute__((aligned(16))) unsigned int recs_all[4][RECS][4];
  __shared__ __attribute__((aligned(16))) unsigned int abuf_all[4][DA][1024];
  __shared__ __attribute__((aligned(16))) unsigned int ctile_all[4][TILEW];             // 64 / CPASS columns x 128 bytes
  __shared__ __attribute__((aligned(16))) unsigned int bimg[BLDS / 4];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  // Start-up in ONE round trip to the L2 before the first A request leaves: the table rows of this wave's columns and the workgroup's share of B are asked for
  // together; the records are built from the table (exchanged through the idle C tile, not loaded a second time) and the first three chunks requested; only then is
  // B written to LDS and the workgroup's barrier passed -- waves beyond the last one take part in the copy and leave after the barrier.
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const bool live = wid < total_waves;
  unsigned int (*recs)[4] = recs_all[wave];
  unsigned int* abuf = abuf_all[wave][0];
  unsigned int* tile = ctile_all[wave];
  unsigned int tile_lds = (unsigned int)(unsigned long long)(lds_ptr_t)tile;
  const unsigned int tt_count = tiles_i * tiles_n, tt = live ? wid % tt_count : 0u, g0 = live ? wid / tt_count : 0u;
  const unsigned int tn = tt % tiles_n, ti = tt / tiles_n;
  const int i0 = (int)ti * 64, n0 = (int)tn * 64;
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = F32 ? p.bk / 16 : p.bk / 32, kw = F32 ? p.bk : p.bk / 2;          // kw: image rows (words per matrix row) of a k-block
  const int nmb = (live && (unsigned int)p.m_blocks > g0) ? (int)(((unsigned int)p.m_blocks - g0 + mbg - 1u) / mbg) : 0;
  // DMA source of LDS slot (lane + 64 x): row kp_l = slot >> 4, the 16-byte group that lands there = (slot & 15) rotated back
  GM const unsigned int* A2 = (GM const unsigned int*)p.a + i0;
  const long long a_mb_words = (long long)(F32 ? p.K : p.K / 2) * p.M;
  unsigned int src_off[NI];
#pragma unroll
  for (int x = 0; x < NI; ++x) {
    const unsigned int S = (unsigned int)lane + 64u * x, kp_l = S >> 4, g = ((S & 15u) - 4u * ((kp_l >> 2) & 1u)) & 15u;
    src_off[x] = kp_l * (unsigned int)p.M + 4u * g;
  }
  // EARLY: the chunk list follows from the host's mask alone -- the first three chunks are requested without a look at the table
  long long first[DA] = {};
  if constexpr (EARLY) {
    const int nch_e = __builtin_popcountll(p.kmask0) * steps, total_e = nmb * nch_e;
#pragma unroll
    for (int f = 0; f < DA; ++f) {
      if (f < total_e) {
        const int j = f / nch_e, c = f - j * nch_e, q = c / steps, st_ = c - q * steps;
        unsigned long long m = p.kmask0;
        for (int z = 0; z < q; ++z) m &= m - 1ull;
        const unsigned int kb = (unsigned int)__builtin_ctzll(m);
        first[f] = (long long)(g0 + (unsigned int)j * mbg) * a_mb_words + (long long)((kb * (unsigned int)kw + 16u * (unsigned int)st_) * (unsigned int)p.M);
      }
    }
  }
  // The table rows of this wave's columns and this thread's pieces of B, then (EARLY) the twelve requests, then ONE wait that leaves the requests in flight.  The loads
  // are written as instructions: a register the compiler knows to be loaded is waited for with s_waitcnt vmcnt(0) while LDS-DMA requests are pending, whatever their
  // place in the queue -- here that would be the first three chunks of A (the wait statement names the registers, so nothing reads them before it).
  GM const unsigned int* gt = (GM const unsigned int*)gtable + (long long)nb0 * nkb + (lane < nkb ? lane : 0);
  constexpr int BP = F32 ? 4 : 3;          // 16-byte pieces of B per thread
  static_assert(BP * 256 * 16 >= BLDS, "the load statements below ask for three (f32: four) pieces per thread");
  const unsigned int pieces = (unsigned int)p.nnzb * (unsigned int)(p.bn * p.bk) / (F32 ? 4u : 8u);
  unsigned int trow[4]; u32x4v bpiece[BP];
  {
    GM const unsigned int* tp[4]; GM const u32x4v* bp_[BP];
#pragma unroll
    for (int nbl = 0; nbl < 4; ++nbl) tp[nbl] = gt + (nbl < NBL ? nbl : NBL - 1) * nkb;
#pragma unroll
    for (int e = 0; e < BP; ++e) { const unsigned int x = threadIdx.x + 256u * e; bp_[e] = (GM const u32x4v*)p.bvals + (x < pieces ? x : 0u); }
    if constexpr (BP == 3)
      asm volatile("global_load_dword %0, %7, off\n\tglobal_load_dword %1, %8, off\n\tglobal_load_dword %2, %9, off\n\tglobal_load_dword %3, %10, off\n\t"
                   "global_load_dwordx4 %4, %11, off\n\tglobal_load_dwordx4 %5, %12, off\n\tglobal_load_dwordx4 %6, %13, off"
                   : "=&v"(trow[0]), "=&v"(trow[1]), "=&v"(trow[2]), "=&v"(trow[3]), "=&v"(bpiece[0]), "=&v"(bpiece[1]), "=&v"(bpiece[2])
                   : "v"(tp[0]), "v"(tp[1]), "v"(tp[2]), "v"(tp[3]), "v"(bp_[0]), "v"(bp_[1]), "v"(bp_[2]) : "memory");
    else
      asm volatile("global_load_dword %0, %8, off\n\tglobal_load_dword %1, %9, off\n\tglobal_load_dword %2, %10, off\n\tglobal_load_dword %3, %11, off\n\t"
                   "global_load_dwordx4 %4, %12, off\n\tglobal_load_dwordx4 %5, %13, off\n\tglobal_load_dwordx4 %6, %14, off\n\tglobal_load_dwordx4 %7, %15, off"
                   : "=&v"(trow[0]), "=&v"(trow[1]), "=&v"(trow[2]), "=&v"(trow[3]), "=&v"(bpiece[0]), "=&v"(bpiece[1]), "=&v"(bpiece[2]), "=&v"(bpiece[BP - 1])
                   : "v"(tp[0]), "v"(tp[1]), "v"(tp[2]), "v"(tp[3]), "v"(bp_[0]), "v"(bp_[1]), "v"(bp_[2]), "v"(bp_[BP - 1]) : "memory");
  }
  if constexpr (EARLY) {
#pragma unroll
    for (int f = 0; f < DA; ++f) {
#pragma unroll
      for (int x = 0; x < NI; ++x)
        __builtin_amdgcn_global_load_lds((GM const void*)(A2 + first[f] + src_off[x]), (lds_ptr_t)((char*)abuf + 4096 * f + 1024 * x), 16, 0, AUX_A);
    }
    if constexpr (BP == 3) asm volatile("s_waitcnt vmcnt(%7)" : "+v"(trow[0]), "+v"(trow[1]), "+v"(trow[2]), "+v"(trow[3]), "+v"(bpiece[0]), "+v"(bpiece[1]), "+v"(bpiece[2]) : "n"(DA * NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(trow[0]), "+v"(trow[1]), "+v"(trow[2]), "+v"(trow[3]), "+v"(bpiece[0]), "+v"(bpiece[1]), "+v"(bpiece[2]), "+v"(bpiece[BP - 1]) : "n"(DA * NI) : "memory");
  } else {
    if constexpr (BP == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(trow[0]), "+v"(trow[1]), "+v"(trow[2]), "+v"(trow[3]), "+v"(bpiece[0]), "+v"(bpiece[1]), "+v"(bpiece[2]) :: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(trow[0]), "+v"(trow[1]), "+v"(trow[2]), "+v"(trow[3]), "+v"(bpiece[0]), "+v"(bpiece[1]), "+v"(bpiece[2]), "+v"(bpiece[BP - 1]) :: "memory");
  }
  if (lane >= nkb) { trow[0] = 0xffffffffu; trow[1] = 0xffffffffu; trow[2] = 0xffffffffu; trow[3] = 0xffffffffu; }
  // one record per chunk, lane c building chunk c's -- without a look at LDS the compiler can see (an LDS access it sees while requests are in flight gets its
  // s_waitcnt vmcnt(0)): the used k-blocks are bits of a mask, the table row of another lane comes through a cross-lane read, the record leaves as an instruction
  bool used = false;
#pragma unroll
  for (int nbl = 0; nbl < NBL; ++nbl) used = used || (trow[nbl] != 0xffffffffu);
  const unsigned long long mask = EARLY ? p.kmask0 : __ballot(used);
  const int nch = __builtin_popcountll(mask) * steps;
  const unsigned int recs_lds = (unsigned int)(unsigned long long)(lds_ptr_t)&recs[0][0], abuf_lds = (unsigned int)(unsigned long long)(lds_ptr_t)abuf;
  u32x4v rec_mine;
  {
    const int q = lane / steps, st_ = lane - q * steps;
    unsigned long long m = mask;
    for (int z = 0; z < q; ++z) m &= m - 1ull;                          // (the q-th used k-block: the q-th set bit)
    const unsigned int kb = m ? (unsigned int)__builtin_ctzll(m) : 0u;
    unsigned int bo[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
#pragma unroll
    for (int nbl = 0; nbl < NBL; ++nbl) {
      const unsigned int blk = (unsigned int)__shfl((int)trow[nbl], (int)kb);
      if (blk != 0xffffffffu) bo[nbl] = (blk * (unsigned int)(16 * BN16) * (unsigned int)p.bk + (F32 ? 16u : 32u) * (unsigned int)st_) * (F32 ? 4u : 2u);
    }
    rec_mine[0] = (kb * (unsigned int)kw + 16u * (unsigned int)st_) * (unsigned int)p.M; rec_mine[1] = bo[0] | (bo[1] << 16); rec_mine[2] = bo[2] | (bo[3] << 16); rec_mine[3] = 0u;
    const unsigned int rad = recs_lds + 16u * (unsigned int)lane;
    if (lane < nch) asm volatile("ds_write_b128 %0, %1" :: "v"(rad), "v"(rec_mine) : "memory");
  }
  const long long c_mb_bytes = (long long)p.N * p.M * (F32 ? 4 : 2);
  f32x4v acc[4][4];
  sfor<16>([&](auto ic) { acc[ic.value / 4][ic.value % 4] = (f32x4v)0.0f; });
  auto store_tile = [&](unsigned int mb) __attribute__((always_inline)) {        // C of M-block mb leaves; the accumulators restart at zero
    GM char* cbase = (GM char*)p.c + (long long)mb * c_mb_bytes;
    if constexpr (F32) {
      // A lane holds 16 bytes of a column in each of its four row tiles: 64-byte pieces of 16 different lines per store.  Through LDS instead, eight columns (2 KiB) at a
      // time -- the lanes of one half of the sub-tile write their four pieces, every lane reads 16 bytes of the linear image back: two 1 KiB stores per pass, sixteen
      // per tile (118 -> 110 us with the stores alone made contiguous, profiles/r06_bcsc_f32_full.jsonl).  The 16-byte slot of a column is XORed with the column number.
      sfor<8>([&](auto pc) {
        constexpr int nt = pc.value / 2, half = pc.value % 2;
        if ((lx >> 3) == half) {
          const unsigned int c = (unsigned int)lx & 7u;
          sfor<4>([&](auto tc) {
            constexpr int it = tc.value;
            const unsigned int wad = tile_lds + c * 256u + 16u * ((unsigned int)(4 * it + kg) ^ c);
            const f32x4v v = acc[nt][it];
            asm volatile("ds_write_b128 %0, %1" :: "v"(wad), "v"(v) : "memory");
          });
        }
        f32x4v w2[2]; unsigned int ad[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { const unsigned int c = 4u * j + ((unsigned int)lane >> 4), sl = (unsigned int)lane & 15u; ad[j] = tile_lds + c * 256u + 16u * (sl ^ c); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w2[0]), "=&v"(w2[1]) : "v"(ad[0]), "v"(ad[1]) : "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = 4 * j + (lane >> 4), sl = lane & 15;
          GM f32x4v* dst = (GM f32x4v*)(cbase + ((long long)(n0 + 16 * nt + 8 * half + c) * p.M + i0) * 4 + 16 * sl);
          if (AUX_A != 0) __builtin_nontemporal_store(w2[j], dst); else *dst = w2[j];
        }
      });
    } else if constexpr (CPASS == 8) {
      // eight columns (1 KiB) per pass: the lanes of one half of a sub-tile write their four 8-byte pieces, every lane reads 16 bytes back, one 1 KiB store
      sfor<8>([&](auto pc) {
        constexpr int nt = pc.value / 2, half = pc.value % 2;
        if ((lx >> 3) == half) {
          const int n = lx & 7;
          sfor<4>([&](auto tc) {
            constexpr int it = tc.value;
            const float x4[4] = {acc[nt][it][0], acc[nt][it][1], acc[nt][it][2], acc[nt][it][3]};
            unsigned int o2[2];
            bf16_pk_exact_n<2>(x4, o2);
            u32x2v v; v[0] = o2[0]; v[1] = o2[1];
            const unsigned int wad = tile_lds + 4u * (unsigned int)(n * 32 + 2 * ((4 * it + kg) ^ n));
            asm volatile("ds_write_b64 %0, %1" :: "v"(wad), "v"(v) : "memory");
          });
        }
        const int n = lane >> 3, j = lane & 7;
        const unsigned int ad = tile_lds + 4u * (unsigned int)(n * 32 + 4 * (j ^ (n >> 1)));
        u32x4v w;
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"(ad) : "memory");
        if (n & 1) { const unsigned int t0 = w[0], t1 = w[1]; w[0] = w[2]; w[1] = w[3]; w[2] = t0; w[3] = t1; }
        GM u32x4v* dst = (GM u32x4v*)(cbase + ((long long)(n0 + 16 * nt + 8 * half + n) * p.M + i0) * 2 + 16 * j);
        if (AUX_A != 0) __builtin_nontemporal_store(w, dst); else *dst = w;
      });
    } else {
    constexpr int NTP = 4 / CPASS, RD = 8 / CPASS;      // 16-column sub-tiles and 1 KiB stores per pass
    sfor<CPASS>([&](auto hc) {
      constexpr int h = hc.value;
      sfor<4 * NTP>([&](auto ic) {
        constexpr int nt = NTP * h + ic.value / 4, it = ic.value % 4;
        const int n = 16 * (nt - NTP * h) + lx;
        const float x4[4] = {acc[nt][it][0], acc[nt][it][1], acc[nt][it][2], acc[nt][it][3]};
        unsigned int o2[2];
        bf16_pk_exact_n<2>(x4, o2);
        u32x2v v; v[0] = o2[0]; v[1] = o2[1];
        const unsigned int wad = tile_lds + 4u * (unsigned int)(n * 32 + 2 * ((4 * it + kg) ^ (n & 15)));
        asm volatile("ds_write_b64 %0, %1" :: "v"(wad), "v"(v) : "memory");
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      u32x4v w4[RD]; unsigned int ad[RD];
#pragma unroll
      for (int r = 0; r < RD; ++r) { const int n = 8 * r + (lane >> 3), j = lane & 7, x = n & 15; ad[r] = tile_lds + 4u * (unsigned int)(n * 32 + 4 * (j ^ (x >> 1))); }
      if constexpr (RD == 4)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w4[0]), "=&v"(w4[1]), "=&v"(w4[2]), "=&v"(w4[3]) : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]) : "memory");
      else
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w4[0]), "=&v"(w4[1]) : "v"(ad[0]), "v"(ad[1]) : "memory");
#pragma unroll
      for (int r = 0; r < RD; ++r) {
        const int n = 8 * r + (lane >> 3), j = lane & 7, x = n & 15;
        u32x4v w = w4[r];
        if (x & 1) { const unsigned int t0 = w[0], t1 = w[1]; w[0] = w[2]; w[1] = w[3]; w[2] = t0; w[3] = t1; }
        GM u32x4v* dst = (GM u32x4v*)(cbase + ((long long)(n0 + 16 * NTP * h + n) * p.M + i0) * 2 + 16 * j);
        if (AUX_A != 0) __builtin_nontemporal_store(w, dst); else *dst = w;
      }
    });
    }
    sfor<16>([&](auto ic) { acc[ic.value / 4][ic.value % 4] = (f32x4v)0.0f; });
  };
  const int rot = 16 * (kg & 1);
  unsigned int a_rd[4];                       // word index of this lane's four A fragments' first element inside a chunk image (row 4 kg, its 16 words of tile t)
#pragma unroll
  for (int t = 0; t < 4; ++t) a_rd[t] = (unsigned int)((4 * kg) * 64 + ((16 * t + lx + rot) & 63));
  unsigned int b_rd[BN16];                    // byte offset of this lane's piece inside a block's fragment (sub-tile s2)
#pragma unroll
  for (int s2 = 0; s2 < BN16; ++s2) b_rd[s2] = (unsigned int)(((16 * s2 + lx) * p.bk + (F32 ? 4 : 8) * kg) * (F32 ? 4 : 2));
  const int total_f = nmb * nch;             // (0 for a wave beyond the last one)
  int aj = 0, ac_ = 0;                        // the NEXT chunk whose A is to be requested
  unsigned int a_slot = 0;                    // ... and the ring slot it goes to
  auto issue_a = [&](unsigned int a_off) __attribute__((always_inline)) {
    GM const unsigned int* rowbase = A2 + (long long)(g0 + (unsigned int)aj * mbg) * a_mb_words + a_off;
    char* dst = (char*)abuf + 4096u * a_slot;
#pragma unroll
    for (int x = 0; x < NI; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(rowbase + src_off[x]), (lds_ptr_t)(dst + 1024 * x), 16, 0, AUX_A);
    a_slot = (a_slot == (unsigned int)(DA - 1)) ? 0u : a_slot + 1u;
    if (++ac_ == nch) { ac_ = 0; ++aj; }
  };
  // B arrived with the table (one round trip); it is in LDS before the first A request leaves, and the barrier below is a bare s_barrier: nothing waits for those requests
  const unsigned int bimg_lds = (unsigned int)(unsigned long long)(lds_ptr_t)bimg;
#pragma unroll
  for (int e = 0; e < BP; ++e) {
    const unsigned int x = threadIdx.x + 256u * e, bad = bimg_lds + 16u * x;
    if (x < pieces) asm volatile("ds_write_b128 %0, %1" :: "v"(bad), "v"(bpiece[e]) : "memory");
  }
  // the first three chunks (a wave with fewer chunks asks for the first bytes of A again, into slots it does not read: one straight line of twelve requests)
  u32x4v rec_c;                                           // chunk 0's record: lane 0's
#pragma unroll
  for (int e = 0; e < 4; ++e) rec_c[e] = (unsigned int)__builtin_amdgcn_readlane((int)rec_mine[e], 0);
  // (all three offsets are read before the first request leaves: a read of LDS the compiler can see waits for every request in flight)
#pragma unroll
  for (int f = 0; f < DA; ++f) {
    const bool real = f < total_f;
    const unsigned int a_off = real ? (unsigned int)__builtin_amdgcn_readlane((int)rec_mine[0], ac_) : 0u;
    if constexpr (!EARLY) first[f] = real ? (long long)(g0 + (unsigned int)aj * mbg) * a_mb_words + a_off : 0ll;
    if (real && ++ac_ == nch) { ac_ = 0; ++aj; }
  }
  if constexpr (!EARLY) {
#pragma unroll
    for (int f = 0; f < DA; ++f) {
#pragma unroll
      for (int x = 0; x < NI; ++x)
        __builtin_amdgcn_global_load_lds((GM const void*)(A2 + first[f] + src_off[x]), (lds_ptr_t)((char*)abuf + 4096 * f + 1024 * x), 16, 0, AUX_A);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (!live) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }        // (no request may land in LDS the workgroup has given back)
  if (nch == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); for (int j = 0; j < nmb; ++j) store_tile(g0 + (unsigned int)j * mbg); return; }     // no block in these columns: C = 0
  int cj = 0, cc = 0;                         // the chunk being consumed
  unsigned int c_slot = 0;
  for (int f = 0; f < total_f; ++f) {
    // chunk f must have landed; behind it in issue order: A(f+1) .. A(f+DA-1) as far as they exist, and the 8 stores of the previous tile while this chunk is one of the
    // first DA of its tile (A(f) was requested DA chunks earlier, in front of them); loads and stores retire this counter in issue order on gfx9
    const int left = total_f - 1 - f;
    const bool stored = cj > 0 && cc < DA;
    const int behind = left < DA - 1 ? left : DA - 1;          // chunks requested behind this one
    sfor<DA>([&](auto bc) {
      if (behind == bc.value) { if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(bc.value * NI + NS) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(bc.value * NI) : "memory"); }
    });
    const unsigned int r1 = (unsigned int)__builtin_amdgcn_readfirstlane((int)rec_c[1]), r2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)rec_c[2]);
    const unsigned int bo[4] = {r1 & 0xffffu, r1 >> 16, r2 & 0xffffu, r2 >> 16};
    // the chunk's LDS reads in one statement, one wait behind them: the A fragments (lane (row, kg): k pairs 4 kg .. 4 kg + 3 of its row, four image rows), the B
    // fragments (from offset 0 for an absent block: read, not used), the next chunk's record and the A offset of the chunk to request.  Written as instructions because
    // the compiler puts s_waitcnt vmcnt(0) in front of a read of LDS it can see while LDS-DMA requests are in flight -- the whole ring would land before every chunk.
    unsigned int a_ad[4], b_ad[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_ad[t] = abuf_lds + 4096u * c_slot + 4u * a_rd[t];
    sfor<NBL>([&](auto nc) {
      constexpr int nbl = nc.value;
      const unsigned int b = bo[nbl] == 0xffffu ? 0u : bo[nbl];
      sfor<BN16>([&](auto sc) { b_ad[nbl * BN16 + sc.value] = bimg_lds + b + b_rd[sc.value]; });
    });
    const int cn = (cc + 1 == nch) ? 0 : cc + 1;
    const unsigned int rec_ad = recs_lds + 16u * (unsigned int)cn, aoff_ad = recs_lds + 16u * (unsigned int)ac_;
    u32x2v ap[8]; u32x4v bq[4]; u32x4v rec_n; unsigned int a_off_v;
    asm volatile("ds_read2_b32 %0, %14 offset1:64\n\tds_read2_b32 %1, %14 offset0:128 offset1:192\n\t"
                 "ds_read2_b32 %2, %15 offset1:64\n\tds_read2_b32 %3, %15 offset0:128 offset1:192\n\t"
                 "ds_read2_b32 %4, %16 offset1:64\n\tds_read2_b32 %5, %16 offset0:128 offset1:192\n\t"
                 "ds_read2_b32 %6, %17 offset1:64\n\tds_read2_b32 %7, %17 offset0:128 offset1:192\n\t"
                 "ds_read_b128 %8, %18\n\tds_read_b128 %9, %19\n\tds_read_b128 %10, %20\n\tds_read_b128 %11, %21\n\t"
                 "ds_read_b128 %12, %22\n\tds_read_b32 %13, %23\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ap[0]), "=&v"(ap[1]), "=&v"(ap[2]), "=&v"(ap[3]), "=&v"(ap[4]), "=&v"(ap[5]), "=&v"(ap[6]), "=&v"(ap[7]),
                   "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(rec_n), "=&v"(a_off_v)
                 : "v"(a_ad[0]), "v"(a_ad[1]), "v"(a_ad[2]), "v"(a_ad[3]), "v"(b_ad[0]), "v"(b_ad[1]), "v"(b_ad[2]), "v"(b_ad[3]), "v"(rec_ad), "v"(aoff_ad)
                 : "memory");
    u32x4v a_cur[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { a_cur[t][0] = ap[2 * t][0]; a_cur[t][1] = ap[2 * t][1]; a_cur[t][2] = ap[2 * t + 1][0]; a_cur[t][3] = ap[2 * t + 1][1]; }
    u32x4v bf_c[NBL][BN16];
    sfor<4>([&](auto ic) { bf_c[ic.value / BN16][ic.value % BN16] = bq[ic.value]; });
    if (left >= DA) issue_a((unsigned int)__builtin_amdgcn_readfirstlane((int)a_off_v));        // into the slot just read
    sfor<NBL>([&](auto nc) {
      constexpr int nbl = nc.value;
      if (bo[nbl] != 0xffffu) {
        sfor<BN16>([&](auto sc) {
          constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
          sfor<4>([&](auto tc) {
            constexpr int t = tc.value;
            if constexpr (F32) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {     // (through scalars: see the general kernel)
                const unsigned int av = a_cur[t][e], bw = bf_c[nbl][s2][e];
                acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av), __uint_as_float(bw), acc[nt][t], 0, 0, 0);
              }
            } else acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a_cur[t]), __builtin_bit_cast(bf16x8v, bf_c[nbl][s2]), acc[nt][t], 0, 0, 0);
          });
        });
      }
    });
    rec_c = rec_n;
    c_slot = (c_slot == (unsigned int)(DA - 1)) ? 0u : c_slot + 1u;
    if (++cc == nch) { store_tile(g0 + (unsigned int)cj * mbg); cc = 0; ++cj; }
  }
}

// 8-bit integers with the A operand on an LDS-DMA ring (the structure of bcsc_mfma_bf16_dma_kernel, re-cut for bytes).  A chunk is 32 k of the
// wave's 64 rows: [8 k-quads][64 i] dwords = 2 KiB = two global_load_lds_dwordx4; rows of odd k-groups are rotated by 16 words on the source
// side, so the operand reads (lane (row, kg): rows 2 kg and 2 kg + 1) are conflict-free ds_read_b32.  The used k-blocks of a k-group are
// compacted into a list first, which makes "chunk c" addressable: the ring runs D chunks ahead and the B
// fragments travel with it in a register ring.  s_waitcnt counts in order, so every chunk issues the SAME number of memory instructions --
// absent blocks load one (wave-uniform) dummy address -- and the wait before chunk c is the static (4 + NI) x (chunks still in flight behind it).
// Full 64 x 64 tiles leave through the idle ring: a lane holds 4 rows of one column (16-byte pieces of 64 different lines); transposed in
// LDS, 16 lanes write one column's 256 bytes.
template <int BN16, bool UA, int AUX_A, int D, int WPS, int RT>     // RT: 16-row tiles per wave (4: 64 x 64 of C per wave, 2: 32 x 64)
__global__ __launch_bounds__(256, WPS) void bcsc_mfma_i8_dma_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int total, const unsigned int* gtable) {
  constexpr int NBL = 4 / BN16, W = 16 * RT, SPR = 4 * RT, NI = RT / 2;      // W: words per image row, SPR: 16-byte slots per row, NI: DMA instructions per chunk
  __shared__ unsigned int tbl_all[4][kBcscTblDma];
  __shared__ unsigned int klist_all[4][64];
  __shared__ __attribute__((aligned(16))) unsigned int abuf_all[4][D][8 * W];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  if (wid >= total) return;
  unsigned int* tbl = tbl_all[wave];
  unsigned int* klist = klist_all[wave];
  unsigned int (*abuf)[8 * W] = abuf_all[wave];
  const unsigned int tn = wid % tiles_n, tmp = wid / tiles_n, ti = tmp % tiles_i, mb = tmp / tiles_i;
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const int i0 = (int)ti * (16 * RT), n0 = (int)tn * 64;
  const int mt = (p.M - i0 >= 16 * RT) ? RT : (p.M - i0) / 16;
  const int nbl_cnt = ((p.N - n0 >= 64) ? 64 : (p.N - n0)) / (16 * BN16);
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = p.bk / 32;
  {
    GM const unsigned int* gt = (GM const unsigned int*)gtable + (long long)nb0 * nkb;
    for (int e = lane; e < nbl_cnt * nkb; e += 64) tbl[e] = gt[e];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  i32x4v acc[4][RT], corr_b[4], corr_a[NBL][RT];
  GM int* cbase = (GM int*)p.c + (long long)mb * p.N * p.M;
  sfor<4 * RT>([&](auto ic) {
    constexpr int nt = ic.value / RT, it = ic.value % RT;
    acc[nt][it] = (i32x4v)0;
    if (!p.beta0 && it < mt && nt < nbl_cnt * BN16) acc[nt][it] = *(GM const i32x4v*)(cbase + (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg);
  });
  sfor<4>([&](auto c) { corr_b[c.value] = (i32x4v)0; });
  sfor<NBL * RT>([&](auto c) { corr_a[c.value / RT][c.value % RT] = (i32x4v)0; });
  const long long ones = 0x0101010101010101ll;
  // DMA source of LDS slot (lane + 64x): k-quad row = slot >> 4, the 16-byte group that lands there = (slot & 15) rotated back
  GM const unsigned int* A4 = (GM const unsigned int*)p.a + (long long)mb * (p.K / 4) * p.M + i0;
  unsigned int src_off[NI];
#pragma unroll
  for (int x = 0; x < NI; ++x) {
    const unsigned int S = (unsigned int)lane + 64u * x, row = S / SPR, g = ((S % SPR) - 4u * ((row >> 1) & 1u)) & (unsigned int)(SPR - 1);
    src_off[x] = row * (unsigned int)p.M + (((int)(4u * g) < 16 * mt) ? 4u * g : 0u);       // groups beyond the wave's rows re-read group 0 (never consumed)
  }
  const int rot = 16 * (kg & 1);
  GM const char* bv = (GM const char*)p.bvals;
  for (int kgp = 0; kgp < nkb; kgp += 64) {
    bool used = false;
    const int kb_l = kgp + lane;
    if (kb_l < nkb) for (int nbl = 0; nbl < nbl_cnt; ++nbl) used = used || (tbl[nbl * nkb + kb_l] != 0xffffffffu);
    const unsigned long long mask = __ballot(used);
    if (mask == 0ull) continue;
    if (used) klist[__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (unsigned int)kb_l;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nch = __builtin_popcountll(mask) * steps;
    unsigned int blk_r[D][NBL]; long long bf_r[D][NBL][BN16];
    auto issue = [&](auto uc, int c) {        // chunk c -> ring position u: its B fragments first (older), then the two DMA instructions
      constexpr int u = decltype(uc)::value;
      const int q = c / steps, st_ = c - q * steps;
      const int kb_ = __builtin_amdgcn_readfirstlane((int)klist[q]);
      sfor<NBL>([&](auto nc) {
        constexpr int nbl = nc.value;
        blk_r[u][nbl] = (nbl < nbl_cnt) ? (unsigned int)__builtin_amdgcn_readfirstlane((int)tbl[nbl * nkb + kb_]) : 0xffffffffu;
        sfor<BN16>([&](auto sc) { constexpr int s2 = sc.value;
          GM const char* src = (blk_r[u][nbl] != 0xffffffffu) ? bv + ((long long)blk_r[u][nbl] * (16 * BN16) + 16 * s2 + lx) * p.bk + 32 * st_ + 8 * kg : bv;
          bf_r[u][nbl][s2] = *(GM const long long*)src; });
      });
      GM const unsigned int* rowbase = A4 + ((long long)kb_ * (p.bk / 4) + 8 * st_) * p.M;
#pragma unroll
      for (int x = 0; x < NI; ++x)
        __builtin_amdgcn_global_load_lds((GM const void*)(rowbase + src_off[x]), (lds_ptr_t)((char*)abuf[u] + 1024 * x), 16, 0, AUX_A);
    };
    sfor<D>([&](auto uc) { if (uc.value < nch) issue(uc, uc.value); });
    for (int c0 = 0; c0 < nch; c0 += D) {
      sfor<D>([&](auto uc) {
        constexpr int u = uc.value;
        const int c = c0 + u;
        if (c < nch) {
          const int behind = (nch - 1 - c < D - 1) ? nch - 1 - c : D - 1;          // chunks issued after c: each is 4 / BN16 * BN16 = 4 loads + NI DMA
          if (D > 3 && behind == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (4 + NI)) : "memory");
          else if (D > 2 && behind == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (4 + NI)) : "memory");
          else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + NI) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          long long a_cur[RT];
          sfor<RT>([&](auto tc) {
            constexpr int t = tc.value;
            if (t < mt) {
              unsigned int lo = abuf[u][(2 * kg) * W + ((16 * t + lx + rot) & (W - 1))], hi = abuf[u][(2 * kg + 1) * W + ((16 * t + lx + rot) & (W - 1))];
              if (UA) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
              a_cur[t] = (long long)(((unsigned long long)hi << 32) | lo);
            }
          });
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          unsigned int blk_c[NBL]; long long bf_c[NBL][BN16];
          sfor<NBL>([&](auto nc) { blk_c[nc.value] = blk_r[u][nc.value]; sfor<BN16>([&](auto sc) { bf_c[nc.value][sc.value] = bf_r[u][nc.value][sc.value]; }); });
          if (c + D < nch) issue(uc, c + D);                                          // the image just read is free again
          sfor<NBL>([&](auto nc) {
            constexpr int nbl = nc.value;
            if (blk_c[nbl] != 0xffffffffu) {
              sfor<BN16>([&](auto sc) {
                constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
                long long bfrag = bf_c[nbl][s2];
                if (!UA) bfrag ^= (long long)0x8080808080808080ull;
                sfor<RT>([&](auto tc) { constexpr int t = tc.value; if (t < mt) acc[nt][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_cur[t], bfrag, acc[nt][t], 0, 0, 0); });
                if (UA) corr_b[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ones, bfrag, corr_b[nt], 0, 0, 0);
              });
              if (!UA) sfor<RT>([&](auto tc) { constexpr int t = tc.value; if (t < mt) corr_a[nbl][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_cur[t], ones, corr_a[nbl][t], 0, 0, 0); });
            }
          });
        }
      });
    }
  }
  if (mt == RT && nbl_cnt * BN16 == 4 && (p.M % 4) == 0) {
    static_assert(D >= 2, "the C tile passes through two chunk images");
    unsigned int* tile = &abuf[0][0];                       // 16 columns x 16 RT rows of int32 per pass (two chunk images)
    sfor<4>([&](auto nc) {
      constexpr int nt = nc.value;
      sfor<RT>([&](auto tc) {
        constexpr int it = tc.value;
        i32x4v v = acc[nt][it];
        if (UA) v += corr_b[nt] * 128; else v += corr_a[nt / BN16][it] * 128;
        *(i32x4v*)(tile + lx * W + 4 * ((4 * it + kg) ^ (lx & (SPR - 1)))) = v;
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < RT; ++r) {                         // SPR lanes write one column's 16 RT rows as whole lines
        const int n = (64 / SPR) * r + lane / SPR, j = lane % SPR;
        const i32x4v w = *(const i32x4v*)(tile + n * W + 4 * (j ^ (n & (SPR - 1))));
        *(GM i32x4v*)(cbase + (long long)(n0 + 16 * nt + n) * p.M + i0 + 4 * j) = w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
    return;
  }
  sfor<4 * RT>([&](auto ic) {
    constexpr int nt = ic.value / RT, it = ic.value % RT;
    if (it < mt && nt < nbl_cnt * BN16) {
      i32x4v v = acc[nt][it];
      if (UA) v += corr_b[nt] * 128; else v += corr_a[nt / BN16][it] * 128;
      *(GM i32x4v*)(cbase + (long long)(n0 + 16 * nt + lx) * p.M + i0 + 16 * it + 4 * kg) = v;
    }
  });
}

// The full-tile streaming kernel (see bcsc_mfma_bf16_stream_full_kernel: records, B in LDS, every in-loop LDS access an instruction) for 8-bit integers:
// u8 x i8 (UA) and i8 x u8 -> i32 on v_mfma_i32_16x16x32_i8 with the unsigned operand fed as u - 128 and 128 * sum(other) added at the end, as in
// bcsc_mfma_i8_dma_kernel.  A chunk is 32 k of the wave's 64 rows in VNNI-4: 8 rows of 256 bytes = 2 KiB = two requests; the ring is six chunks deep; B up to
// 8 KiB; the int32 tile leaves through LDS eight columns (2 KiB per wave) at a time like the f32 form.  68 KiB of LDS, two waves per SIMD (at three, 168 registers, every
// variant spills: the 64 + 16 .. 64 accumulators).
template <int BN16, bool UA, int AUX_A, bool EARLY>
__global__ __launch_bounds__(256, 2) void bcsc_mfma_i8_stream_full_kernel(BcscArgs p, unsigned int tiles_i, unsigned int tiles_n, unsigned int mbg, unsigned int total_waves, const unsigned int* gtable) {
  constexpr int NBL = 4 / BN16, DA = 6, NI = 2, NS = 16, BLDS = 8192, BP = 2;
  __shared__ __attribute__((aligned(16))) unsigned int recs_all[4][kBcscRecs][4];
  __shared__ __attribute__((aligned(16))) unsigned int abuf_all[4][DA][512];
  __shared__ __attribute__((aligned(16))) unsigned int ctile_all[4][512];
  __shared__ __attribute__((aligned(16))) unsigned int bimg[BLDS / 4];
  const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int wid = blockIdx.x * 4u + wave;
  const int lane = threadIdx.x & 63, lx = lane & 15, kg = lane >> 4;
  const bool live = wid < total_waves;
  unsigned int* abuf = abuf_all[wave][0];
  const unsigned int tile_lds = (unsigned int)(unsigned long long)(lds_ptr_t)ctile_all[wave];
  const unsigned int recs_lds = (unsigned int)(unsigned long long)(lds_ptr_t)&recs_all[wave][0][0], abuf_lds = (unsigned int)(unsigned long long)(lds_ptr_t)abuf;
  const unsigned int bimg_lds = (unsigned int)(unsigned long long)(lds_ptr_t)bimg;
  const unsigned int tt_count = tiles_i * tiles_n, tt = live ? wid % tt_count : 0u, g0 = live ? wid / tt_count : 0u;
  const unsigned int tn = tt % tiles_n, ti = tt / tiles_n;
  const int i0 = (int)ti * 64, n0 = (int)tn * 64;
  const int nb0 = n0 / (16 * BN16);
  const int nkb = p.K / p.bk, steps = p.bk / 32, kw = p.bk / 4;          // kw: image rows (k quads) of a k-block
  const int nmb = (live && (unsigned int)p.m_blocks > g0) ? (int)(((unsigned int)p.m_blocks - g0 + mbg - 1u) / mbg) : 0;
  GM const unsigned int* A4 = (GM const unsigned int*)p.a + i0;
  const long long a_mb_words = (long long)(p.K / 4) * p.M;
  unsigned int src_off[NI];
#pragma unroll
  for (int x = 0; x < NI; ++x) {
    const unsigned int S = (unsigned int)lane + 64u * x, row = S >> 4, g = ((S & 15u) - 4u * ((row >> 1) & 1u)) & 15u;
    src_off[x] = row * (unsigned int)p.M + 4u * g;
  }
  long long first[DA] = {};
  if constexpr (EARLY) {
    const int nch_e = __builtin_popcountll(p.kmask0) * steps, total_e = nmb * nch_e;
#pragma unroll
    for (int f = 0; f < DA; ++f) {
      if (f < total_e) {
        const int j = f / nch_e, c = f - j * nch_e, q = c / steps, st_ = c - q * steps;
        unsigned long long m = p.kmask0;
        for (int z = 0; z < q; ++z) m &= m - 1ull;
        const unsigned int kb = (unsigned int)__builtin_ctzll(m);
        first[f] = (long long)(g0 + (unsigned int)j * mbg) * a_mb_words + (long long)((kb * (unsigned int)kw + 8u * (unsigned int)st_) * (unsigned int)p.M);
      }
    }
  }
  GM const unsigned int* gt = (GM const unsigned int*)gtable + (long long)nb0 * nkb + (lane < nkb ? lane : 0);
  const unsigned int pieces = (unsigned int)p.nnzb * (unsigned int)(p.bn * p.bk) / 16u;
  unsigned int trow[4]; u32x4v bpiece[BP];
  {
    GM const unsigned int* tp[4]; GM const u32x4v* bp_[BP];
#pragma unroll
    for (int nbl = 0; nbl < 4; ++nbl) tp[nbl] = gt + (nbl < NBL ? nbl : NBL - 1) * nkb;
#pragma unroll
    for (int e = 0; e < BP; ++e) { const unsigned int x = threadIdx.x + 256u * e; bp_[e] = (GM const u32x4v*)p.bvals + (x < pieces ? x : 0u); }
    asm volatile("global_load_dword %0, %6, off\n\tglobal_load_dword %1, %7, off\n\tglobal_load_dword %2, %8, off\n\tglobal_load_dword %3, %9, off\n\t"
                 "global_load_dwordx4 %4, %10, off\n\tglobal_load_dwordx4 %5, %11, off"
                 : "=&v"(trow[0]), "=&v"(trow[1]), "=&v"(trow[2]), "=&v"(trow[3]), "=&v"(bpiece[0]), "=&v"(bpiece[1])
                 : "v"(tp[0]), "v"(tp[1]), "v"(tp[2]), "v"(tp[3]), "v"(bp_[0]), "v"(bp_[1]) : "memory");
  }
  if constexpr (EARLY) {
#pragma unroll
    for (int f = 0; f < DA; ++f) {
#pragma unroll
      for (int x = 0; x < NI; ++x)
        __builtin_amdgcn_global_load_lds((GM const void*)(A4 + first[f] + src_off[x]), (lds_ptr_t)((char*)abuf + 2048 * f + 1024 * x), 16, 0, AUX_A);
    }
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(trow[0]), "+v"(trow[1]), "+v"(trow[2]), "+v"(trow[3]), "+v"(bpiece[0]), "+v"(bpiece[1]) : "n"(DA * NI) : "memory");
  } else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(trow[0]), "+v"(trow[1]), "+v"(trow[2]), "+v"(trow[3]), "+v"(bpiece[0]), "+v"(bpiece[1]) :: "memory");
  if (lane >= nkb) { trow[0] = 0xffffffffu; trow[1] = 0xffffffffu; trow[2] = 0xffffffffu; trow[3] = 0xffffffffu; }
  bool used = false;
#pragma unroll
  for (int nbl = 0; nbl < NBL; ++nbl) used = used || (trow[nbl] != 0xffffffffu);
  const unsigned long long mask = EARLY ? p.kmask0 : __ballot(used);
  const int nch = __builtin_popcountll(mask) * steps;
  u32x4v rec_mine;
  {
    const int q = lane / steps, st_ = lane - q * steps;
    unsigned long long m = mask;
    for (int z = 0; z < q; ++z) m &= m - 1ull;
    const unsigned int kb = m ? (unsigned int)__builtin_ctzll(m) : 0u;
    unsigned int bo[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
#pragma unroll
    for (int nbl = 0; nbl < NBL; ++nbl) {
      const unsigned int blk = (unsigned int)__shfl((int)trow[nbl], (int)kb);
      if (blk != 0xffffffffu) bo[nbl] = blk * (unsigned int)(16 * BN16) * (unsigned int)p.bk + 32u * (unsigned int)st_;
    }
    rec_mine[0] = (kb * (unsigned int)kw + 8u * (unsigned int)st_) * (unsigned int)p.M; rec_mine[1] = bo[0] | (bo[1] << 16); rec_mine[2] = bo[2] | (bo[3] << 16); rec_mine[3] = 0u;
    const unsigned int rad = recs_lds + 16u * (unsigned int)lane;
    if (lane < nch) asm volatile("ds_write_b128 %0, %1" :: "v"(rad), "v"(rec_mine) : "memory");
  }
  const long long c_mb_bytes = (long long)p.N * p.M * 4;
  i32x4v acc[4][4], corr_b[4], corr_a[NBL][4];
  sfor<16>([&](auto ic) { acc[ic.value / 4][ic.value % 4] = (i32x4v)0; });
  sfor<4>([&](auto c) { corr_b[c.value] = (i32x4v)0; });
  sfor<NBL * 4>([&](auto c) { corr_a[c.value / 4][c.value % 4] = (i32x4v)0; });
  const long long ones = 0x0101010101010101ll;
  auto store_tile = [&](unsigned int mb) __attribute__((always_inline)) {
    GM char* cbase = (GM char*)p.c + (long long)mb * c_mb_bytes;
    sfor<8>([&](auto pc) {
      constexpr int nt = pc.value / 2, half = pc.value % 2;
      if ((lx >> 3) == half) {
        const unsigned int c = (unsigned int)lx & 7u;
        sfor<4>([&](auto tc) {
          constexpr int it = tc.value;
          i32x4v v = acc[nt][it];
          if (UA) v += corr_b[nt] * 128; else v += corr_a[nt / BN16][it] * 128;
          const unsigned int wad = tile_lds + c * 256u + 16u * ((unsigned int)(4 * it + kg) ^ c);
          asm volatile("ds_write_b128 %0, %1" :: "v"(wad), "v"(v) : "memory");
        });
      }
      u32x4v w2[2]; unsigned int ad[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) { const unsigned int c = 4u * j + ((unsigned int)lane >> 4), sl = (unsigned int)lane & 15u; ad[j] = tile_lds + c * 256u + 16u * (sl ^ c); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w2[0]), "=&v"(w2[1]) : "v"(ad[0]), "v"(ad[1]) : "memory");
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 4 * j + (lane >> 4), sl = lane & 15;
        GM u32x4v* dst = (GM u32x4v*)(cbase + ((long long)(n0 + 16 * nt + 8 * half + c) * p.M + i0) * 4 + 16 * sl);
        if (AUX_A != 0) __builtin_nontemporal_store(w2[j], dst); else *dst = w2[j];
      }
    });
    sfor<16>([&](auto ic) { acc[ic.value / 4][ic.value % 4] = (i32x4v)0; });
    sfor<4>([&](auto c) { corr_b[c.value] = (i32x4v)0; });
    sfor<NBL * 4>([&](auto c) { corr_a[c.value / 4][c.value % 4] = (i32x4v)0; });
  };
  const int rot = 16 * (kg & 1);
  unsigned int a_rd[4];                       // word index of the low k quad (image row 2 kg) of this lane's row in tile t; the high one is a row (64 words) further
#pragma unroll
  for (int t = 0; t < 4; ++t) a_rd[t] = (unsigned int)((2 * kg) * 64 + ((16 * t + lx + rot) & 63));
  unsigned int b_rd[BN16];
#pragma unroll
  for (int s2 = 0; s2 < BN16; ++s2) b_rd[s2] = (unsigned int)((16 * s2 + lx) * p.bk + 8 * kg);
  const int total_f = nmb * nch;
  int aj = 0, ac_ = 0;
  unsigned int a_slot = 0;
  auto issue_a = [&](unsigned int a_off) __attribute__((always_inline)) {
    GM const unsigned int* rowbase = A4 + (long long)(g0 + (unsigned int)aj * mbg) * a_mb_words + a_off;
    char* dst = (char*)abuf + 2048u * a_slot;
#pragma unroll
    for (int x = 0; x < NI; ++x)
      __builtin_amdgcn_global_load_lds((GM const void*)(rowbase + src_off[x]), (lds_ptr_t)(dst + 1024 * x), 16, 0, AUX_A);
    a_slot = (a_slot == (unsigned int)(DA - 1)) ? 0u : a_slot + 1u;
    if (++ac_ == nch) { ac_ = 0; ++aj; }
  };
#pragma unroll
  for (int e = 0; e < BP; ++e) {
    const unsigned int x = threadIdx.x + 256u * e, bad = bimg_lds + 16u * x;
    if (x < pieces) asm volatile("ds_write_b128 %0, %1" :: "v"(bad), "v"(bpiece[e]) : "memory");
  }
  u32x4v rec_c;
#pragma unroll
  for (int e = 0; e < 4; ++e) rec_c[e] = (unsigned int)__builtin_amdgcn_readlane((int)rec_mine[e], 0);
#pragma unroll
  for (int f = 0; f < DA; ++f) {
    const bool real = f < total_f;
    const unsigned int a_off = real ? (unsigned int)__builtin_amdgcn_readlane((int)rec_mine[0], ac_) : 0u;
    if constexpr (!EARLY) first[f] = real ? (long long)(g0 + (unsigned int)aj * mbg) * a_mb_words + a_off : 0ll;
    if (real && ++ac_ == nch) { ac_ = 0; ++aj; }
  }
  if constexpr (!EARLY) {
#pragma unroll
    for (int f = 0; f < DA; ++f) {
#pragma unroll
      for (int x = 0; x < NI; ++x)
        __builtin_amdgcn_global_load_lds((GM const void*)(A4 + first[f] + src_off[x]), (lds_ptr_t)((char*)abuf + 2048 * f + 1024 * x), 16, 0, AUX_A);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (!live) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
  if (nch == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); for (int j = 0; j < nmb; ++j) store_tile(g0 + (unsigned int)j * mbg); return; }
  int cj = 0, cc = 0;
  unsigned int c_slot = 0;
  for (int f = 0; f < total_f; ++f) {
    // chunk f must have landed; behind it: A(f+1) .. A(f+DA-1) as far as they exist and the 16 stores of the previous tile while this chunk is one of the first DA of its tile
    const int left = total_f - 1 - f;
    const bool stored = cj > 0 && cc < DA;
    const int behind = left < DA - 1 ? left : DA - 1;          // chunks requested behind this one
    sfor<DA>([&](auto bc) {
      if (behind == bc.value) { if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(bc.value * NI + NS) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(bc.value * NI) : "memory"); }
    });
    const unsigned int r1 = (unsigned int)__builtin_amdgcn_readfirstlane((int)rec_c[1]), r2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)rec_c[2]);
    const unsigned int bo[4] = {r1 & 0xffffu, r1 >> 16, r2 & 0xffffu, r2 >> 16};
    unsigned int a_ad[4], b_ad[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_ad[t] = abuf_lds + 2048u * c_slot + 4u * a_rd[t];
    sfor<NBL>([&](auto nc) {
      constexpr int nbl = nc.value;
      const unsigned int b = bo[nbl] == 0xffffu ? 0u : bo[nbl];
      sfor<BN16>([&](auto sc) { b_ad[nbl * BN16 + sc.value] = bimg_lds + b + b_rd[sc.value]; });
    });
    const int cn = (cc + 1 == nch) ? 0 : cc + 1;
    const unsigned int rec_ad = recs_lds + 16u * (unsigned int)cn, aoff_ad = recs_lds + 16u * (unsigned int)ac_;
    u32x2v ap[4], bq[4]; u32x4v rec_n; unsigned int a_off_v;
    asm volatile("ds_read2_b32 %0, %10 offset1:64\n\tds_read2_b32 %1, %11 offset1:64\n\tds_read2_b32 %2, %12 offset1:64\n\tds_read2_b32 %3, %13 offset1:64\n\t"
                 "ds_read_b64 %4, %14\n\tds_read_b64 %5, %15\n\tds_read_b64 %6, %16\n\tds_read_b64 %7, %17\n\t"
                 "ds_read_b128 %8, %18\n\tds_read_b32 %9, %19\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ap[0]), "=&v"(ap[1]), "=&v"(ap[2]), "=&v"(ap[3]), "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(rec_n), "=&v"(a_off_v)
                 : "v"(a_ad[0]), "v"(a_ad[1]), "v"(a_ad[2]), "v"(a_ad[3]), "v"(b_ad[0]), "v"(b_ad[1]), "v"(b_ad[2]), "v"(b_ad[3]), "v"(rec_ad), "v"(aoff_ad)
                 : "memory");
    if (left >= DA) issue_a((unsigned int)__builtin_amdgcn_readfirstlane((int)a_off_v));        // into the slot just read
    long long a_cur[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned int lo = ap[t][0], hi = ap[t][1];
      if (UA) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
      a_cur[t] = (long long)(((unsigned long long)hi << 32) | lo);
    }
    sfor<NBL>([&](auto nc) {
      constexpr int nbl = nc.value;
      if (bo[nbl] != 0xffffu) {
        sfor<BN16>([&](auto sc) {
          constexpr int s2 = sc.value, nt = nbl * BN16 + s2;
          long long bfrag = (long long)(((unsigned long long)bq[nt][1] << 32) | bq[nt][0]);
          if (!UA) bfrag ^= (long long)0x8080808080808080ull;
          sfor<4>([&](auto tc) { constexpr int t = tc.value; acc[nt][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_cur[t], bfrag, acc[nt][t], 0, 0, 0); });
          if (UA) corr_b[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ones, bfrag, corr_b[nt], 0, 0, 0);
        });
        if (!UA) sfor<4>([&](auto tc) { constexpr int t = tc.value; corr_a[nbl][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_cur[t], ones, corr_a[nbl][t], 0, 0, 0); });
      }
    });
    rec_c = rec_n;
    c_slot = (c_slot == (unsigned int)(DA - 1)) ? 0u : c_slot + 1u;
    if (++cc == nch) { store_tile(g0 + (unsigned int)cj * mbg); cc = 0; ++cj; }
  }
}

// What one launch moves: the k-blocks of A some block of B refers to (known from a host-resident pattern when one n-tile covers all columns: BcscArgs::kmask0;
// all of A otherwise) and C.  A launch that moves more than the Infinity Cache holds cannot leave anything there for the next one: its A is requested non-temporal
// (config #4: 7 of 8 k-blocks of 256 MiB + 64 MiB of C -- 52.6 against 56.7 us; the same shape with bn = 32 touches half of A: 192 MiB, cacheable, 34.6 against 36.6 us;
// profiles/r06_bcsc_full.jsonl).
static unsigned long long bcsc_launch_bytes(const BcscArgs& a, unsigned long long a_elem, unsigned long long c_elem) {
  const unsigned long long mb = (unsigned long long)std::max(a.m_blocks, 0), M = (unsigned long long)std::max(a.M, 0);
  unsigned long long k_used = (unsigned long long)std::max(a.K, 0);
  if (a.nnzb > 0 && a.N <= 64 && a.bk > 0 && a.K / a.bk <= 64) k_used = (unsigned long long)__builtin_popcountll(a.kmask0) * (unsigned long long)a.bk;
  return mb * M * (k_used * a_elem + (unsigned long long)std::max(a.N, 0) * c_elem);
}

int launch_bcsc(const BcscArgs& a_in, void* stream, const char** name) {
  hipStream_t st = (hipStream_t)stream;
  BcscArgs a = a_in;
  {
    const unsigned long long es = a.a_type == LIBXSMM_DATATYPE_F32 ? 4ull : (a.a_type == LIBXSMM_DATATYPE_BF16 ? 2ull : 1ull);
    const unsigned long long moved = bcsc_launch_bytes(a, es, a.c_type == LIBXSMM_DATATYPE_BF16 ? 2ull : 4ull);
    a.nt_a = (a.stream_hint == 2 || (a.stream_hint == 0 && (moved > (256ull << 20) || rt_recent_operands_exceed_cache(a.a, moved, a.c)))) ? 1 : 0;
  }
  if (a.m_blocks <= 0 || a.M <= 0 || a.N <= 0) { if (name) *name = "(empty)"; return 0; }
  // matrix-core path: bf16 with VNNI-2 A, 32-deep k steps, 16-wide n sub-tiles, 16-row i tiles, 8-byte aligned C columns
  {
    constexpr bool off = false;
    const int nbl_per_wave = (a.bn > 0 && 64 % a.bn == 0) ? 64 / a.bn : 0;
    const bool shape_ok = a.a_type == LIBXSMM_DATATYPE_BF16 && a.vnni_a && a.bk % 32 == 0 && (a.bn == 16 || a.bn == 32 || a.bn == 64) && a.M % 16 == 0 &&
      (long long)nbl_per_wave * (a.K / a.bk) <= kBcscTbl && ((size_t)a.a % 4 == 0) && ((size_t)a.bvals % 16 == 0) && ((size_t)a.c % 16 == 0);
    if (!off && shape_ok) {
      const unsigned int tiles_i = (unsigned int)((a.M + 63) / 64), tiles_n = (unsigned int)((a.N + 63) / 64);
      const long long total = (long long)tiles_i * tiles_n * a.m_blocks;
      if (total < (1ll << 31)) {
        const dim3 grid((unsigned int)((total + 3) / 4));
        constexpr bool dma = true;
        const bool dma_ok = dma && a.table != nullptr && (long long)nbl_per_wave * (a.K / a.bk) <= kBcscTblDma && ((size_t)a.a % 16 == 0) && (a.M % 4 == 0) && ((long long)(a.K / 2) * a.M < (1ll << 30));
        if (dma_ok) {
          const int nkb = a.K / a.bk;
          const unsigned int* table = (const unsigned int*)a.table;
          // the inverted pattern: already in place when the pattern came from host memory (built there, cached per kernel: run_bcsc)
          if (!a.table_ready) hipLaunchKernelGGL(bcsc_invert_kernel, dim3((unsigned int)a.nblk_n), dim3(64), 0, st, a.colptr, a.rowidx, (unsigned int*)a.table, a.nblk_n, nkb);
          // A is read exactly once: stream it non-temporally when it cannot be cache resident anyway (or the caller says so)
          const bool nta = a.nt_a != 0;
          // many M-blocks per (i-tile, n-tile): waves that stream over M-blocks (two per SIMD: 2048 on the chip), each taking every mbg-th block
          constexpr int stream_mode = 1;
          const long long tt_count = (long long)tiles_i * tiles_n;
          if (stream_mode != 0 && a.beta0 && nkb <= 64 && tt_count <= 2048 && ((long long)a.m_blocks * tt_count >= 4096 || stream_mode == 2) && ((long long)(a.K / 2) * a.M) * (long long)a.m_blocks < (1ll << 40)) {
            // a value array that fits beside the rings: every workgroup keeps its own LDS copy of B and no wave asks the L2 for a
            // fragment again (bn = 32: 39.5 -> 35.7 us on 8192 M-blocks of 64 x 256, bn = 16 unchanged; profiles/r06_bcsc_b_in_lds.jsonl)
            const bool b_lds = a.nnzb > 0 && (long long)a.nnzb * a.bn * a.bk * 2 <= kBcscBLds && ((size_t)a.bvals % 16 == 0);
            // ... and whole 64 x 64 tiles with bf16 C: the kernel with one record per chunk
            const bool full = b_lds && a.c_type == LIBXSMM_DATATYPE_BF16 && a.M % 64 == 0 && a.N % 64 == 0 && ((size_t)a.c % 16 == 0) && nkb * (a.bk / 32) <= kBcscRecs;
            // ... on three workgroups per CU (ring depth 2) when B and the record list fit the smaller LDS plan
#if !defined(XAMD_BCSC_THREE)
#define XAMD_BCSC_THREE 0           // measured equal to ring depth 3 on two workgroups per CU (config #4 52.1-52.3 against 52.3-52.4 us, 32 768 M-blocks 213.8 against 209.4: the kernel is
#endif                              // bound by the memory system, not by what a wave does between its waits; profiles/r06_bcsc_three.jsonl) -- not instantiated unless built with -DXAMD_BCSC_THREE=1
            const bool three = XAMD_BCSC_THREE != 0 && full && (long long)a.nnzb * a.bn * a.bk * 2 <= 8192 && nkb * (a.bk / 32) <= kBcscRecs / 2;
            // 32 rows per wave (RT = 2: 161 VGPRs, three waves per SIMD) measured 72 us against 61 us: every B fragment then feeds two MFMAs instead of four
            const long long slots = three ? 3072 : 2048;      // waves per round: two per SIMD (the general kernel: 245 VGPRs; three spill inside the chunk loop: 105 instead of 61 us), three for `three`
            long long mbg = std::min<long long>(a.m_blocks, std::max<long long>(1, slots / tt_count));
            const long long per = (a.m_blocks + mbg - 1) / mbg;
            mbg = (a.m_blocks + per - 1) / per;
            const long long waves = mbg * tt_count;
            const dim3 sgrid((unsigned int)((waves + 3) / 4));
            if (full) {
              const bool early = tiles_n == 1;          // (kmask0 describes the first n-tile)
#if !defined(XAMD_BCSC_AUX_NT)
#define XAMD_BCSC_AUX_NT 2          // cache-policy bits of the A requests of a launch that streams (A/B builds: 3, 16, 18: profiles/r06_bcsc_full.jsonl)
#endif
#define LAUNCH_FULL4_(B_, X_, E_, D_) hipLaunchKernelGGL((bcsc_mfma_bf16_stream_full_kernel<B_, X_, E_, D_>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table)
#if !defined(XAMD_BCSC_DEEP)
#define XAMD_BCSC_DEEP 0            // ring depth 4 where B (<= 8 KiB) and the record list (<= 32) leave room for it: built, verified (124 parity tests) and measured equal on
                                    // config #4 (51.8-52.8 against 51.5-52.4 us), worse on short tiles (bn = 64: 30.5 against 26.7 us, the eight-pass epilogue): not instantiated
                                    // unless built with -DXAMD_BCSC_DEEP=1 (profiles/r06_bcsc_deep.jsonl).  The copy floor with LDS-DMA reads equals the one with register loads.
#endif
              const bool deep = XAMD_BCSC_DEEP != 0 && (long long)a.nnzb * a.bn * a.bk * 2 <= 8192 && nkb * (a.bk / 32) <= kBcscRecs / 2;
#define LAUNCH_FULL3_(B_, X_, E_) do { if constexpr (XAMD_BCSC_THREE != 0) { if (three) LAUNCH_FULL4_(B_, X_, E_, 2); else LAUNCH_FULL4_(B_, X_, E_, 3); } \
                                       else if constexpr (XAMD_BCSC_DEEP != 0) { if (deep) LAUNCH_FULL4_(B_, X_, E_, 4); else LAUNCH_FULL4_(B_, X_, E_, 3); } else LAUNCH_FULL4_(B_, X_, E_, 3); } while (0)
#define LAUNCH_FULL_(B_) do { if (early) { if (nta) LAUNCH_FULL3_(B_, XAMD_BCSC_AUX_NT, true); else LAUNCH_FULL3_(B_, 0, true); } \
                              else if (nta) LAUNCH_FULL3_(B_, XAMD_BCSC_AUX_NT, false); else LAUNCH_FULL3_(B_, 0, false); } while (0)
              if (a.bn == 16) LAUNCH_FULL_(1); else if (a.bn == 32) LAUNCH_FULL_(2); else LAUNCH_FULL_(4);
#undef LAUNCH_FULL_
#undef LAUNCH_FULL3_
#undef LAUNCH_FULL4_
              if (name) *name = "bcsc_mfma_bf16_stream_full_kernel";
              return (int)hipGetLastError();
            }
#define LAUNCH_STREAM_(B_) do { if (b_lds) { if (nta) hipLaunchKernelGGL((bcsc_mfma_bf16_stream_kernel<B_, 2, 4, 2, false, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); \
                                             else hipLaunchKernelGGL((bcsc_mfma_bf16_stream_kernel<B_, 0, 4, 2, false, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); } \
                                else if (nta) hipLaunchKernelGGL((bcsc_mfma_bf16_stream_kernel<B_, 2>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); \
                                else hipLaunchKernelGGL((bcsc_mfma_bf16_stream_kernel<B_, 0>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); } while (0)
            if (a.bn == 16) LAUNCH_STREAM_(1); else if (a.bn == 32) LAUNCH_STREAM_(2); else LAUNCH_STREAM_(4);
#undef LAUNCH_STREAM_
            if (name) *name = "bcsc_mfma_bf16_stream_kernel";
            return (int)hipGetLastError();
          }
#define LAUNCH_DMA_(B_) do { if (nta) hipLaunchKernelGGL((bcsc_mfma_bf16_dma_kernel<B_, 2>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total, table); \
                             else hipLaunchKernelGGL((bcsc_mfma_bf16_dma_kernel<B_, 0>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total, table); } while (0)
          if (a.bn == 16) LAUNCH_DMA_(1); else if (a.bn == 32) LAUNCH_DMA_(2); else LAUNCH_DMA_(4);
#undef LAUNCH_DMA_
          if (name) *name = "bcsc_mfma_bf16_dma_kernel";
          return (int)hipGetLastError();
        }
        if (a.bn == 16) hipLaunchKernelGGL((bcsc_mfma_bf16_kernel<1>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total);
        else if (a.bn == 32) hipLaunchKernelGGL((bcsc_mfma_bf16_kernel<2>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total);
        else hipLaunchKernelGGL((bcsc_mfma_bf16_kernel<4>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total);
        if (name) *name = "bcsc_mfma_bf16_kernel";
        return (int)hipGetLastError();
      }
    }
  }
  {   // f32 on the matrix cores: 16-deep k steps, 16-wide n sub-tiles, 16-row i tiles, 16-byte aligned B blocks and C columns
    constexpr bool off = false;
    const int nbl_per_wave = (a.bn > 0 && 64 % a.bn == 0) ? 64 / a.bn : 0;
    if (!off && a.a_type == LIBXSMM_DATATYPE_F32 && a.c_type == LIBXSMM_DATATYPE_F32 && a.bk % 16 == 0 && (a.bn == 16 || a.bn == 32 || a.bn == 64) && a.M % 16 == 0 && a.N % a.bn == 0 &&
        (long long)nbl_per_wave * (a.K / a.bk) <= kBcscTbl && ((size_t)a.a % 4 == 0) && ((size_t)a.bvals % 16 == 0) && ((size_t)a.c % 16 == 0) && (a.M % 4 == 0)) {
      const unsigned int tiles_i = (unsigned int)((a.M + 63) / 64), tiles_n = (unsigned int)((a.N + 63) / 64);
      const long long total = (long long)tiles_i * tiles_n * a.m_blocks;
      if (total < (1ll << 31)) {
        const dim3 grid((unsigned int)((total + 3) / 4));
        // many M-blocks per (i-tile, n-tile): the bf16 kernel's waves streaming over M-blocks, on f32 operands (round 3; the one-tile-per-wave kernel below
        // builds its pattern rows in every wave and fetches A one 16-deep step ahead: 0.65 of the HBM roofline on config #4's shape)
        constexpr int stream_mode = 1;
        const int nkb = a.K / a.bk;
        const long long tt_count = (long long)tiles_i * tiles_n;
        if (stream_mode != 0 && a.table != nullptr && a.beta0 && nkb <= 64 && (long long)nbl_per_wave * nkb <= kBcscTblDma && ((size_t)a.a % 16 == 0) && (long long)a.K * a.M < (1ll << 30) &&
            tt_count <= 2048 && ((long long)a.m_blocks * tt_count >= 4096 || stream_mode == 2) && ((long long)a.K * a.M) * (long long)a.m_blocks < (1ll << 40)) {
          const unsigned int* table = (const unsigned int*)a.table;
          if (!a.table_ready) hipLaunchKernelGGL(bcsc_invert_kernel, dim3((unsigned int)a.nblk_n), dim3(64), 0, st, a.colptr, a.rowidx, (unsigned int*)a.table, a.nblk_n, nkb);
          // two waves per SIMD (182 registers); three (168: ten spills) measured 139.7 us against 119.1 us
          long long mbg = std::min<long long>(a.m_blocks, std::max<long long>(1, 2048 / tt_count));
          const long long per = (a.m_blocks + mbg - 1) / mbg;
          mbg = (a.m_blocks + per - 1) / per;
          const long long waves = mbg * tt_count;
          const dim3 sgrid((unsigned int)((waves + 3) / 4));
          // host-resident / bound pattern, whole 64 x 64 tiles, B up to 16 KiB: the kernel with one record per chunk (see bcsc_mfma_bf16_stream_full_kernel)
          if (a.nnzb > 0 && (long long)a.nnzb * a.bn * a.bk * 4 <= 16384 && ((size_t)a.bvals % 16 == 0) && a.M % 64 == 0 && a.N % 64 == 0 && nkb * (a.bk / 16) <= kBcscRecs) {
            const bool early = tiles_n == 1;
#define LAUNCH_FULL_F32_(B_) do { if (early) { if (a.nt_a) hipLaunchKernelGGL((bcsc_mfma_bf16_stream_full_kernel<B_, 2, true, 3, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); \
                                               else hipLaunchKernelGGL((bcsc_mfma_bf16_stream_full_kernel<B_, 0, true, 3, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); } \
                                  else if (a.nt_a) hipLaunchKernelGGL((bcsc_mfma_bf16_stream_full_kernel<B_, 2, false, 3, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); \
                                  else hipLaunchKernelGGL((bcsc_mfma_bf16_stream_full_kernel<B_, 0, false, 3, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); } while (0)
            if (a.bn == 16) LAUNCH_FULL_F32_(1); else if (a.bn == 32) LAUNCH_FULL_F32_(2); else LAUNCH_FULL_F32_(4);
#undef LAUNCH_FULL_F32_
            if (name) *name = "bcsc_mfma_f32_stream_full_kernel";
            return (int)hipGetLastError();
          }
#define LAUNCH_STREAM_F32_(B_) do { if (a.nt_a) hipLaunchKernelGGL((bcsc_mfma_bf16_stream_kernel<B_, 2, 4, 2, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); \
                                    else hipLaunchKernelGGL((bcsc_mfma_bf16_stream_kernel<B_, 0, 4, 2, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); } while (0)
          if (a.bn == 16) LAUNCH_STREAM_F32_(1); else if (a.bn == 32) LAUNCH_STREAM_F32_(2); else LAUNCH_STREAM_F32_(4);
#undef LAUNCH_STREAM_F32_
          if (name) *name = "bcsc_mfma_f32_stream_kernel";
          return (int)hipGetLastError();
        }
        if (a.bn == 16) hipLaunchKernelGGL((bcsc_mfma_f32_kernel<1>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total);
        else if (a.bn == 32) hipLaunchKernelGGL((bcsc_mfma_f32_kernel<2>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total);
        else hipLaunchKernelGGL((bcsc_mfma_f32_kernel<4>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total);
        if (name) *name = "bcsc_mfma_f32_kernel";
        return (int)hipGetLastError();
      }
    }
  }
  {   // 8-bit integers on the matrix cores: VNNI-4 A, 32-deep k steps, 16-wide n sub-tiles, 16-row i tiles, dword / 8-byte / 16-byte aligned A / B / C
    const bool i8 = (a.a_type == LIBXSMM_DATATYPE_U8 && a.b_type == LIBXSMM_DATATYPE_I8) || (a.a_type == LIBXSMM_DATATYPE_I8 && a.b_type == LIBXSMM_DATATYPE_U8);
    constexpr bool off = false;
    const int nbl_per_wave = (a.bn > 0 && 64 % a.bn == 0) ? 64 / a.bn : 0;
    if (!off && i8 && a.c_type == LIBXSMM_DATATYPE_I32 && a.vnni_a && a.bk % 32 == 0 && (a.bn == 16 || a.bn == 32 || a.bn == 64) && a.M % 16 == 0 && a.N % a.bn == 0 &&
        (long long)nbl_per_wave * (a.K / a.bk) <= kBcscTbl && ((size_t)a.a % 4 == 0) && ((size_t)a.bvals % 8 == 0) && ((size_t)a.c % 16 == 0)) {
      const unsigned int tiles_i = (unsigned int)((a.M + 63) / 64), tiles_n = (unsigned int)((a.N + 63) / 64);
      const long long total = (long long)tiles_i * tiles_n * a.m_blocks;
      if (total < (1ll << 31)) {
        const dim3 grid((unsigned int)((total + 3) / 4));
        const bool ua = a.a_type == LIBXSMM_DATATYPE_U8;
        constexpr bool dma = true;
        if (dma && a.table != nullptr && (long long)nbl_per_wave * (a.K / a.bk) <= kBcscTblDma && ((size_t)a.a % 16 == 0) && (a.M % 4 == 0) && ((long long)(a.K / 4) * a.M < (1ll << 30))) {
          const int nkb = a.K / a.bk;
          const unsigned int* table = (const unsigned int*)a.table;
          if (!a.table_ready) hipLaunchKernelGGL(bcsc_invert_kernel, dim3((unsigned int)a.nblk_n), dim3(64), 0, st, a.colptr, a.rowidx, (unsigned int*)a.table, a.nblk_n, nkb);
          // host-resident / bound pattern, whole 64 x 64 tiles, beta = 0, B up to 8 KiB, many M-blocks per tile: waves streaming over M-blocks with one record per chunk
          const long long tt_count8 = (long long)tiles_i * tiles_n;
          if (a.beta0 && a.nnzb > 0 && (long long)a.nnzb * a.bn * a.bk <= 8192 && ((size_t)a.bvals % 16 == 0) && a.M % 64 == 0 && a.N % 64 == 0 && nkb <= 64 && nkb * (a.bk / 32) <= kBcscRecs &&
              tt_count8 <= 2048 && (long long)a.m_blocks * tt_count8 >= 4096) {
            long long mbg = std::min<long long>(a.m_blocks, std::max<long long>(1, 2048 / tt_count8));          // two waves per SIMD
            const long long per = (a.m_blocks + mbg - 1) / mbg;
            mbg = (a.m_blocks + per - 1) / per;
            const long long waves = mbg * tt_count8;
            const dim3 sgrid((unsigned int)((waves + 3) / 4));
            const bool early = tiles_n == 1;
#define LAUNCH_I8F3_(B_, U_, X_) do { if (early) hipLaunchKernelGGL((bcsc_mfma_i8_stream_full_kernel<B_, U_, X_, true>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); \
                                      else hipLaunchKernelGGL((bcsc_mfma_i8_stream_full_kernel<B_, U_, X_, false>), sgrid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)mbg, (unsigned int)waves, table); } while (0)
#define LAUNCH_I8F_(B_) do { if (ua) { if (a.nt_a) LAUNCH_I8F3_(B_, true, 2); else LAUNCH_I8F3_(B_, true, 0); } else { if (a.nt_a) LAUNCH_I8F3_(B_, false, 2); else LAUNCH_I8F3_(B_, false, 0); } } while (0)
            if (a.bn == 16) LAUNCH_I8F_(1); else if (a.bn == 32) LAUNCH_I8F_(2); else LAUNCH_I8F_(4);
#undef LAUNCH_I8F_
#undef LAUNCH_I8F3_
            if (name) *name = "bcsc_mfma_i8_stream_full_kernel";
            return (int)hipGetLastError();
          }
#define LAUNCH_I8D_(B_) do { \
    if (ua) { if (a.nt_a) hipLaunchKernelGGL((bcsc_mfma_i8_dma_kernel<B_, true, 2, 2, 3, 4>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total, table); \
              else hipLaunchKernelGGL((bcsc_mfma_i8_dma_kernel<B_, true, 0, 2, 3, 4>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total, table); } \
    else { if (a.nt_a) hipLaunchKernelGGL((bcsc_mfma_i8_dma_kernel<B_, false, 2, 2, (B_ == 1 ? 2 : 3), 4>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total, table); \
           else hipLaunchKernelGGL((bcsc_mfma_i8_dma_kernel<B_, false, 0, 2, (B_ == 1 ? 2 : 3), 4>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total, table); } } while (0)
          // ring depth 2, three waves per SIMD (141 VGPRs), 64 rows per wave -- measured on 8192 M-blocks of 64 x 256 (2:8, bn = 16): depth 2 / 3 / 4 at two waves
          // per SIMD 60.5 / 61.2 / 61.2 us, three waves per SIMD 56.0 us; 32 rows per wave (104 VGPRs, four waves per SIMD) 60.5 us: twice the B loads and
          // per-wave set-up outweigh the occupancy (profiles/r02_bcsc_counters.txt).  Waves streaming over M-blocks (the bf16 kernel's +9 %): 55.9 us here,
          // no gain -- half of this kernel's traffic is the int32 C it writes, not the operand stream the scheme keeps busy -- so it was not kept.
          // Signed A (i8 x u8) at bn = 16 carries 64 more correction accumulators (one set per N-block): under the three-waves cap (168 registers) the
          // compiler spilled 74 of them to scratch inside the loop (tools/kernel_resources.py), so that one variant is compiled for two waves per SIMD.
          if (a.bn == 16) LAUNCH_I8D_(1); else if (a.bn == 32) LAUNCH_I8D_(2); else LAUNCH_I8D_(4);
#undef LAUNCH_I8D_
          if (name) *name = "bcsc_mfma_i8_dma_kernel";
          return (int)hipGetLastError();
        }
#define LAUNCH_I8_(B_) do { if (ua) hipLaunchKernelGGL((bcsc_mfma_i8_kernel<B_, true>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total); \
                            else hipLaunchKernelGGL((bcsc_mfma_i8_kernel<B_, false>), grid, dim3(256), 0, st, a, tiles_i, tiles_n, (unsigned int)total); } while (0)
        if (a.bn == 16) LAUNCH_I8_(1); else if (a.bn == 32) LAUNCH_I8_(2); else LAUNCH_I8_(4);
#undef LAUNCH_I8_
        if (name) *name = "bcsc_mfma_i8_kernel";
        return (int)hipGetLastError();
      }
    }
  }
  const long long blocks = (long long)((a.M + 63) / 64) * a.N * a.m_blocks;
  hipLaunchKernelGGL(bcsc_generic_kernel, dim3((unsigned int)((blocks + 3) / 4)), dim3(64, 4), 0, st, a);
  if (name) *name = "bcsc_generic_kernel";
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Dense packed GEMM, general form (the specialised kernel comes from jit.cpp): one thread per C element,
// lanes along the packed axis.  [ref: samples/xgemm_packed/gemm_packed_kernel.c:35-72]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pgemm_generic_kernel(PgemmArgs p) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = p.P * p.M;
  if (gid >= per * p.N) return;
  const long long n = gid / per, rem = gid - n * per, m = rem / p.P, l = rem - m * p.P;
  GM const T* a = (GM const T*)p.a; GM const T* b = (GM const T*)p.b; GM T* c = (GM T*)p.c + (n * p.ldc + m) * p.P + l;
  T acc = p.beta0 ? (T)0 : *c;
  for (int k = 0; k < p.K; ++k) acc = fma(a[((long long)k * p.lda + m) * p.P + l], b[(n * p.ldb + k) * p.P + l], acc);
  *c = acc;
}
int launch_pgemm(const PgemmArgs& a, void* stream, const char** name) {
  hipStream_t st = (hipStream_t)stream;
  const long long total = a.P * a.M * a.N;
  if (total <= 0) { if (name) *name = "(empty)"; return 0; }
  if ((total + 255) / 256 >= (1ll << 31)) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned int)((total + 255) / 256));
  if (a.dtype == LIBXSMM_DATATYPE_F64) hipLaunchKernelGGL(pgemm_generic_kernel<double>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(pgemm_generic_kernel<float>, grid, dim3(256), 0, st, a);
  if (name) *name = "pgemm_generic_kernel";
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Packed SpGEMM with a sparse C (libxsmm_create_packed_spgemm_csc with ldc == 0): for every stored entry (m, n) of C
//   C_val[z] (+)= sum_k sum_p A[k][m][p] * B[k][n][p]          [ref: src/generator_packed_spgemm_csc_csparse_avx_avx2_avx512.c:17-195]
// -- the packed axis is REDUCED (a sampled dense-dense product).  One wave per stored entry, lanes along p, a wave reduction at the end.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csparse_kernel(CsparseArgs p) {
  const unsigned int z = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (z >= p.nnz) return;
  const int lane = threadIdx.x & 63;
  const unsigned int m = ((GM const unsigned int*)p.rows)[z], n = ((GM const unsigned int*)p.cols)[z];
  GM const float* a = (GM const float*)p.a + (long long)m * p.P;
  GM const float* b = (GM const float*)p.b + (long long)n * p.P;
  const long long sa = (long long)p.lda * p.P, sb = (long long)p.ldb * p.P;
  float acc = 0.0f;
  const bool vec = (p.P % 4 == 0) && ((((size_t)p.a | (size_t)p.b) & 15) == 0);
  for (int k = 0; k < p.K; ++k) {
    GM const float* ak = a + k * sa; GM const float* bk = b + k * sb;
    if (vec) {
      for (long long q = 4ll * lane; q < p.P; q += 256) {
        const f32x4v x = *(GM const f32x4v*)(ak + q), y = *(GM const f32x4v*)(bk + q);
        acc = fmaf(x[0], y[0], acc); acc = fmaf(x[1], y[1], acc); acc = fmaf(x[2], y[2], acc); acc = fmaf(x[3], y[3], acc);
      }
    } else {
      for (long long q = lane; q < p.P; q += 64) acc = fmaf(ak[q], bk[q], acc);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) { GM float* c = (GM float*)p.c + z; *c = p.beta0 ? acc : *c + acc; }
}
int launch_csparse(const CsparseArgs& a, void* stream, const char** name) {
  if (name) *name = "csparse_kernel";
  if (a.nnz == 0 || a.K <= 0 || a.P <= 0) return 0;
  hipLaunchKernelGGL(csparse_kernel, dim3((a.nnz + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // namespace xamd
