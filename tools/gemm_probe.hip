// gemm_probe.hip -- stand-alone A/B harness for the headline shape (f32 32x32x32, batch of independent
// problems, contiguous tiles).  Not part of the library: it exists to choose the memory path of
// gemm_mfma_f32_kernel by measurement (interleaved rounds in one process, HIP events, L3-cold rotation).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip -o gpurun_out/gemm_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/libxsmm.h"   // the library itself, through its C ABI, as one more variant

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int jl_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- V0: memory floor.  Each wave streams its A and B tile in and a C tile out with 16-byte accesses.
__global__ __launch_bounds__(256) void k_copy(const float* A, const float* B, float* C, int nb) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= nb) return;
  const f32x4* a = (const f32x4*)(A + (size_t)wid * 1024); const f32x4* b = (const f32x4*)(B + (size_t)wid * 1024);
  f32x4* c = (f32x4*)(C + (size_t)wid * 1024);
  f32x4 va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = a[lane + 64 * q]; vb[q] = b[lane + 64 * q]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) c[lane + 64 * q] = va[q] + vb[q];
}

// ---- V1: the library's current path: A via 16 coalesced dword loads, B via 4 x 16-byte loads at a
// 128-byte lane stride + permlane32_swap to natural k order, C via 16 dword stores (2 full lines each).
__global__ __launch_bounds__(256) void k_direct(const float* A, const float* B, float* C, int nb) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  if (wid >= nb) return;
  const float* a = A + (size_t)wid * 1024; const float* b = B + (size_t)wid * 1024; float* c = C + (size_t)wid * 1024;
  float af[16], v[16], bf[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) af[s] = a[li + (2 * s + h) * 32];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const f32x4 t = *(const f32x4*)(b + li * 32 + 16 * h + 4 * q); v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3]; }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
    bf[s] = __uint_as_float(r[0]); bf[s + 8] = __uint_as_float(r[1]);
  }
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) c[li + jl_of(r, h) * 32] = acc[r];
}

// ---- V2: every global access is a fully coalesced 16-byte/lane access; operands are redistributed through
// wave-private LDS (no barriers).  A: linear image, fragment reads ds_read_b32 (lanes along i, conflict free).
// B: 16-byte chunks XOR-swizzled by column so that the per-column ds_read_b128 is conflict free.
// STORE_LDS: also transpose C through LDS to write it with 4 x 16-byte stores instead of 16 x dword.
template <bool STORE_LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_lds(const float* A, const float* B, float* C, int nb) {
  __shared__ __attribute__((aligned(16))) float lds[WAVES][2048];
  const int w = threadIdx.x >> 6;
  const int wid = blockIdx.x * WAVES + w, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  if (wid >= nb) return;
  const float* a = A + (size_t)wid * 1024; const float* b = B + (size_t)wid * 1024; float* c = C + (size_t)wid * 1024;
  float* la = lds[w]; float* lb = lds[w] + 1024;
  f32x4 va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = ((const f32x4*)a)[lane + 64 * q]; vb[q] = ((const f32x4*)b)[lane + 64 * q]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ((f32x4*)la)[lane + 64 * q] = va[q];
    // B chunk index t = lane + 64q: column j = t / 8, chunk c = t % 8 (4 consecutive k); swizzle c ^= (j >> 1) & 7
    const int t = lane + 64 * q, j = t >> 3, cc = (t & 7) ^ ((j >> 1) & 7);
    ((f32x4*)lb)[j * 8 + cc] = vb[q];
  }
  float af[16], v[16], bf[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) af[s] = la[li + (2 * s + h) * 32];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cc = (4 * h + q) ^ ((li >> 1) & 7);
    const f32x4 t = ((const f32x4*)lb)[li * 8 + cc];
    v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
    bf[s] = __uint_as_float(r[0]); bf[s + 8] = __uint_as_float(r[1]);
  }
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
  if (!STORE_LDS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[li + jl_of(r, h) * 32] = acc[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) la[li + jl_of(r, h) * 32] = acc[r];     // column-major image of C, conflict free
#pragma unroll
    for (int q = 0; q < 4; ++q) ((f32x4*)c)[lane + 64 * q] = ((const f32x4*)la)[lane + 64 * q];
  }
}

// ---- V3: V2 with every wave processing TPW tiles back to back: the global loads of tile t+1 are issued
// before the MFMAs of tile t (register double buffer), so load, compute and store phases of different
// tiles overlap inside a wave instead of relying on other waves.
template <int TPW>
__global__ __launch_bounds__(256) void k_lds_pipe(const float* A, const float* B, float* C, int nb) {
  __shared__ __attribute__((aligned(16))) float lds[4][2048];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int wave = blockIdx.x * 4 + w, nwaves = gridDim.x * 4;
  float* la = lds[w]; float* lb = lds[w] + 1024;
  f32x4 va[4], vb[4];
  int t = wave;
  if (t < nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { va[q] = ((const f32x4*)(A + (size_t)t * 1024))[lane + 64 * q]; vb[q] = ((const f32x4*)(B + (size_t)t * 1024))[lane + 64 * q]; }
  }
  for (; t < nb; t += nwaves) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ((f32x4*)la)[lane + 64 * q] = va[q];
      const int tt = lane + 64 * q, j = tt >> 3, cc = (tt & 7) ^ ((j >> 1) & 7);
      ((f32x4*)lb)[j * 8 + cc] = vb[q];
    }
    const int tn = t + nwaves;
    if (tn < nb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { va[q] = ((const f32x4*)(A + (size_t)tn * 1024))[lane + 64 * q]; vb[q] = ((const f32x4*)(B + (size_t)tn * 1024))[lane + 64 * q]; }
    }
    float af[16], v[16], bf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) af[s] = la[li + (2 * s + h) * 32];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = (4 * h + q) ^ ((li >> 1) & 7);
      const f32x4 x = ((const f32x4*)lb)[li * 8 + cc];
      v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
      bf[s] = __uint_as_float(r[0]); bf[s + 8] = __uint_as_float(r[1]);
    }
    f32x16 acc = {0};
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
    float* c = C + (size_t)t * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[li + jl_of(r, h) * 32] = acc[r];
  }
  (void)TPW;
}

static libxsmm_gemmfunction g_lib_kernel = nullptr;
static void lib_launch(const float* A, const float* B, float* C, int nb) {
  static unsigned long long one = 1;
  libxsmm_gemm_param p; memset(&p, 0, sizeof(p));
  p.a.primary = (void*)A; p.b.primary = (void*)B; p.c.primary = C; p.op.tertiary = &one;
  libxsmm_hip_gemm_batch_strided(g_lib_kernel, &p, (size_t)nb, 4096, 4096, 4096);
}

struct Variant { const char* name; int waves_per_block; int tiles_per_wave; void (*fn)(const float*, const float*, float*, int); };

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 4096;
  const int rounds = argc > 2 ? atoi(argv[2]) : 20;
  const int inner = 20;
  const size_t tile = 1024, set_elems = (size_t)nb * tile;
  const int nsets = std::max(2, (int)((600ull << 20) / (set_elems * 4 * 3)) + 1);
  std::vector<float*> A(nsets), B(nsets), C(nsets);
  std::vector<float> ha(set_elems), hb(set_elems);
  for (size_t i = 0; i < set_elems; ++i) { ha[i] = (float)((int)(i * 7919u % 10) - 4) / 10.0f; hb[i] = (float)((int)(i * 104729u % 10) - 4) / 10.0f; }
  for (int s = 0; s < nsets; ++s) {
    CHECK(hipMalloc(&A[s], set_elems * 4)); CHECK(hipMalloc(&B[s], set_elems * 4)); CHECK(hipMalloc(&C[s], set_elems * 4));
    CHECK(hipMemcpy(A[s], ha.data(), set_elems * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(B[s], hb.data(), set_elems * 4, hipMemcpyHostToDevice));
  }
  Variant vs[] = {
    {"copy_floor", 4, 1, k_copy}, {"direct(lib)", 4, 1, k_direct}, {"lds", 4, 1, k_lds<false, 4>}, {"lds+cstore", 4, 1, k_lds<true, 4>},
    {"lds_w8", 8, 1, k_lds<false, 8>}, {"lds_pipe2", 4, 2, k_lds_pipe<2>}, {"lds_pipe4", 4, 4, k_lds_pipe<4>},
    {"LIBRARY", 0, 1, nullptr},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]);
  {
    const libxsmm_gemm_shape sh = libxsmm_create_gemm_shape(32, 32, 32, 32, 32, 32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
    g_lib_kernel = libxsmm_dispatch_brgemm(sh, LIBXSMM_GEMM_FLAG_BETA_0, 0, libxsmm_create_gemm_batch_reduce_config(LIBXSMM_GEMM_BATCH_REDUCE_STRIDE, 4096, 4096, 0));
    if (!g_lib_kernel) { printf("library dispatch failed\n"); return 1; }
    libxsmm_hip_set_stream(nullptr);   // stream-ordered launches on the null stream, like the other variants
  }
  // correctness cross-check of the GEMM variants against the library-style kernel
  std::vector<float> ref(set_elems), got(set_elems);
  hipLaunchKernelGGL(k_direct, dim3((nb + 3) / 4), dim3(256), 0, 0, A[0], B[0], C[0], nb);
  CHECK(hipMemcpy(ref.data(), C[0], set_elems * 4, hipMemcpyDeviceToHost));
  for (int v = 2; v < nv; ++v) {
    CHECK(hipMemset(C[0], 0, set_elems * 4));
    const int wpb = vs[v].waves_per_block, tpw = vs[v].tiles_per_wave;
    if (!vs[v].fn) lib_launch(A[0], B[0], C[0], nb);
    else { const int blocks = (nb + wpb * tpw - 1) / (wpb * tpw); hipLaunchKernelGGL(vs[v].fn, dim3(blocks), dim3(wpb * 64), 0, 0, A[0], B[0], C[0], nb); }
    CHECK(hipMemcpy(got.data(), C[0], set_elems * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < set_elems; ++i) if (got[i] != ref[i]) ++bad;
    printf("check %-12s mismatches=%zu\n", vs[v].name, bad);
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<std::vector<float>> us(nv);
  for (int mode = 0; mode < 2; ++mode) {          // 0: rotate sets (HBM), 1: same set (L3 resident)
    for (auto& u : us) u.clear();
    for (int r = 0; r < rounds; ++r) {
      for (int v = 0; v < nv; ++v) {
        const int wpb = vs[v].waves_per_block, tpw = vs[v].tiles_per_wave;
        const int blocks = vs[v].fn ? (nb + wpb * tpw - 1) / (wpb * tpw) : 0;
        CHECK(hipEventRecord(e0, 0));
        for (int it = 0; it < inner; ++it) {
          const int s = mode == 0 ? (r * inner + it) % nsets : 0;
          if (!vs[v].fn) lib_launch(A[s], B[s], C[s], nb);
          else hipLaunchKernelGGL(vs[v].fn, dim3(blocks), dim3(wpb * 64), 0, 0, A[s], B[s], C[s], nb);
        }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        us[v].push_back(ms * 1000.0f / inner);
      }
    }
    printf("---- nb=%d %s (per-launch us incl. launch gap; GB/s algorithmic = 12 KiB per tile)\n", nb, mode == 0 ? "HBM-cold (rotating sets)" : "L3-resident");
    for (int v = 0; v < nv; ++v) {
      std::sort(us[v].begin(), us[v].end());
      const float med = us[v][us[v].size() / 2], mn = us[v][0];
      printf("%-12s median %7.2f us  min %7.2f us  -> %7.1f GB/s (median)  %6.1f GFLOP/s\n", vs[v].name, med, mn,
             (double)nb * 12288.0 / med / 1e3, (double)nb * 65536.0 / med / 1e3);
    }
  }
  return 0;
}
