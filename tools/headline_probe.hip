// headline_probe.hip -- round-2 A/B harness for BASELINE config #2 (f32 32x32x32, 4096 independent problems, 12 KiB each).
// Question it answers: is the 11.25 us of gemm_f32_stream_kernel the floor of this footprint (32 MiB read + 16 MiB
// written in ONE round of waves), or do the lock-step phases (all waves load, then all multiply, then all store) cost time?
//   * copy_*: kernels with the same footprint and no arithmetic = the memory floor, in several launch geometries;
//   * gemm_occ<W,R>: the library's algorithm with W waves per workgroup and residency limited to R waves per CU by
//     an LDS pad, so that the launch becomes 16/R staggered rounds (loads of round r+1 overlap stores of round r);
//   * gemm_half: 8192 waves, each a 32x16 half of C on v_mfma_f32_16x16x4 (more memory-level parallelism, A read twice via L2);
//   * gemm_two: 2048 waves with two problems each, all 16 loads issued up front;
//   * LIBRARY: libxsmm_hip_gemm_batch_strided through the C ABI.
// Every kernel has its own name so that `rocprofv3 --kernel-trace --stats` of this binary gives per-variant durations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/headline_probe.hip -Llibxsmm_amd/lib -lxsmm_amd -Wl,-rpath,'$ORIGIN/../libxsmm_amd/lib' -o tools/headline_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/libxsmm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int jl_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
template <bool NT> __device__ __forceinline__ f32x4 ld4(const f32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st4(f32x4* p, f32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// ---- memory floor: one wave per problem, 8 x 16-byte loads and 4 x 16-byte stores per lane -------------------------------
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_tile(const float* A, const float* B, float* C, int nb) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= nb) return;
  const f32x4* a = (const f32x4*)(A + (size_t)wid * 1024); const f32x4* b = (const f32x4*)(B + (size_t)wid * 1024);
  f32x4* c = (f32x4*)(C + (size_t)wid * 1024);
  f32x4 va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = ld4<NTL>(a + lane + 64 * q); vb[q] = ld4<NTL>(b + lane + 64 * q); }
#pragma unroll
  for (int q = 0; q < 4; ++q) st4<NTS>(c + lane + 64 * q, va[q] + vb[q]);
}
// round 4: the same with every XCD (hardware workgroup b -> XCD b % 8) owning a CONTIGUOUS eighth of the problems instead of every eighth workgroup's
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_tile_xcd(const float* A, const float* B, float* C, int nb) {
  const unsigned int b = blockIdx.x, g = gridDim.x;
  const unsigned int lb = (g & 7u) == 0u ? (b & 7u) * (g >> 3) + (b >> 3) : b;
  const int wid = (int)(lb * 4u) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= nb) return;
  const f32x4* a = (const f32x4*)(A + (size_t)wid * 1024); const f32x4* bb = (const f32x4*)(B + (size_t)wid * 1024);
  f32x4* c = (f32x4*)(C + (size_t)wid * 1024);
  f32x4 va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = ld4<NTL>(a + lane + 64 * q); vb[q] = ld4<NTL>(bb + lane + 64 * q); }
#pragma unroll
  for (int q = 0; q < 4; ++q) st4<NTS>(c + lane + 64 * q, va[q] + vb[q]);
}
// fine-grained: one 16-byte C chunk per thread (2 loads, 1 store), 4x the waves
__global__ __launch_bounds__(256) void copy_fine(const float* A, const float* B, float* C, int nb) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)nb * 256) return;
  __builtin_nontemporal_store(((const f32x4*)A)[i] + ((const f32x4*)B)[i], (f32x4*)C + i);
}
// persistent: 2 workgroups per CU x 8 loads in flight per lane, grid-stride over the tiles
__global__ __launch_bounds__(256) void copy_persist(const float* A, const float* B, float* C, int nb) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  for (int wid = blockIdx.x * 4 + (threadIdx.x >> 6); wid < nb; wid += nw) {
    const f32x4* a = (const f32x4*)(A + (size_t)wid * 1024); const f32x4* b = (const f32x4*)(B + (size_t)wid * 1024);
    f32x4* c = (f32x4*)(C + (size_t)wid * 1024);
    f32x4 va[4], vb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { va[q] = a[lane + 64 * q]; vb[q] = b[lane + 64 * q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(va[q] + vb[q], c + lane + 64 * q);
  }
}

// ---- the library's algorithm (gemm_f32_stream_kernel, NN, contiguous tiles) with launch geometry as a parameter ---------------
__device__ __forceinline__ void frags_from_lds(float (&af)[16], float (&bf)[16], const float* la, const float* lb, int li, int h) {
  float v[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) af[s] = la[li + (2 * s + h) * 32];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = ((const f32x4*)lb)[li * 8 + ((4 * h + q) ^ ((li >> 1) & 7))];
    v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * s]), __float_as_uint(v[2 * s + 1]), false, false);
    bf[s] = __uint_as_float(r[0]); bf[s + 8] = __uint_as_float(r[1]);
  }
}
__device__ __forceinline__ void park(float* la, float* lb, const f32x4 (&va)[4], const f32x4 (&vb)[4], int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ((f32x4*)la)[lane + 64 * q] = va[q];
    const int t = lane + 64 * q, j = t >> 3, cc = (t & 7) ^ ((j >> 1) & 7);
    ((f32x4*)lb)[j * 8 + cc] = vb[q];
  }
}
// WAVES per workgroup; RES = resident waves per CU enforced by padding the static LDS (160 KiB / (RES / WAVES) per workgroup)
template <int WAVES, int RES, bool NTL>
__global__ __launch_bounds__(WAVES * 64) void gemm_occ(const float* A, const float* B, float* C, int nb) {
  constexpr int kWgPerCu = RES / WAVES;
  constexpr int kFloats = (160 * 1024 / kWgPerCu) / 4 - 64;          // whole budget of one residency slot
  static_assert(kFloats >= WAVES * 2048, "residency too high for 8 KiB per wave");
  __shared__ __attribute__((aligned(16))) float lds[kFloats];
  const int w = threadIdx.x >> 6;
  const int wid = blockIdx.x * WAVES + w, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  if (wid >= nb) return;
  const float* a = A + (size_t)wid * 1024; const float* b = B + (size_t)wid * 1024; float* c = C + (size_t)wid * 1024;
  float* la = lds + w * 2048; float* lb = la + 1024;
  f32x4 va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = ld4<NTL>((const f32x4*)a + lane + 64 * q); vb[q] = ld4<NTL>((const f32x4*)b + lane + 64 * q); }
  park(la, lb, va, vb, lane);
  float af[16], bf[16];
  frags_from_lds(af, bf, la, lb, li, h);
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[r], c + li + jl_of(r, h) * 32);
}

// two problems per wave, all sixteen loads in flight before the first use
template <int RES>
__global__ __launch_bounds__(256) void gemm_two(const float* A, const float* B, float* C, int nb) {
  constexpr int kFloats = (160 * 1024 / (RES / 4)) / 4 - 64;
  __shared__ __attribute__((aligned(16))) float lds[kFloats];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int wid = (blockIdx.x * 4 + w) * 2;
  if (wid >= nb) return;
  float* la = lds + w * 2048; float* lb = la + 1024;
  f32x4 va[2][4], vb[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) { va[t][q] = ((const f32x4*)(A + (size_t)(wid + t) * 1024))[lane + 64 * q]; vb[t][q] = ((const f32x4*)(B + (size_t)(wid + t) * 1024))[lane + 64 * q]; }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    park(la, lb, va[t], vb[t], lane);
    float af[16], bf[16];
    frags_from_lds(af, bf, la, lb, li, h);
    f32x16 acc = {0};
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
    float* c = C + (size_t)(wid + t) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[r], c + li + jl_of(r, h) * 32);
  }
}

// half tiles: wave (2p + hh) computes columns [16 hh, 16 hh + 16) of problem p with v_mfma_f32_16x16x4_f32.
// Operand roles as in the library (product formed transposed): SrcA <- B (rows of the MFMA = j), SrcB <- A (cols = i).
// 16x16x4: lane l supplies A-operand element (row = l & 15, k = l >> 4) and B-operand element (k = l >> 4, col = l & 15);
// D[4 regs]: row = 4 * (l >> 4) + r, col = l & 15.  With rows = j and cols = i a lane's four results are four columns j
// of one row i: each register is stored as 16 lanes x 4 bytes = 64 contiguous bytes per j.
typedef float f32x4v __attribute__((ext_vector_type(4)));
// The two waves of a problem share one LDS image of A (each fetches half of it: 2 KiB) and keep their own half of B:
// 4 loads per lane and wave, one workgroup barrier.  Row strides are padded (A: 48 floats per k, B: 36 floats per j) so that
// the fragment reads are (nearly) conflict free and the 16-byte image writes stay aligned.
__global__ __launch_bounds__(256) void gemm_half(const float* A, const float* B, float* C, int nb) {
  __shared__ __attribute__((aligned(16))) float lds[2][32 * 48 + 2 * 16 * 36];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pl = w >> 1, hh = w & 1;                        // problem inside the workgroup, half of C
  const int p = blockIdx.x * 2 + pl;
  const bool live = p < nb;
  float* la = lds[pl]; float* lb = la + 32 * 48 + hh * 16 * 36;
  if (live) {
    const float* a = A + (size_t)p * 1024 + hh * 512; const float* b = B + (size_t)p * 1024 + hh * 512;
    f32x4 va[2], vb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { va[q] = ((const f32x4*)a)[lane + 64 * q]; vb[q] = ((const f32x4*)b)[lane + 64 * q]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q, row = t >> 3, cc = t & 7;       // row: k (A, + 16 hh) or local j (B)
      *(f32x4*)(la + (16 * hh + row) * 48 + 4 * cc) = va[q];
      *(f32x4*)(lb + row * 36 + 4 * cc) = vb[q];
    }
  }
  __syncthreads();
  if (!live) return;
  float* c = C + (size_t)p * 1024 + hh * 512;
  const int l15 = lane & 15, kq = lane >> 4;
  f32x4v acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int k = 4 * ks + kq;
    const float bv = lb[l15 * 36 + k];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const float av = la[k * 48 + 16 * it + l15];
      acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[it], 0, 0, 0);
    }
  }
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(acc[it][r], c + (4 * kq + r) * 32 + 16 * it + l15);
}

static libxsmm_gemmfunction g_lib_kernel = nullptr;
static void lib_launch(const float* A, const float* B, float* C, int nb) {
  static unsigned long long one = 1;
  libxsmm_gemm_param p; memset(&p, 0, sizeof(p));
  p.a.primary = (void*)A; p.b.primary = (void*)B; p.c.primary = C; p.op.tertiary = &one;
  libxsmm_hip_gemm_batch_strided(g_lib_kernel, &p, (size_t)nb, 4096, 4096, 4096);
}

typedef void (*kfn)(const float*, const float*, float*, int);
struct Variant { const char* name; int threads; int problems_per_block_x2; kfn fn; bool is_gemm; int fixed_blocks; };
// problems_per_block_x2: twice the problems one workgroup covers (gemm_half: 4 waves = 2 problems -> 4)

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 4096;
  const int rounds = argc > 2 ? atoi(argv[2]) : 15;
  const int nmodes = argc > 3 ? atoi(argv[3]) : 2;      // 1: HBM-cold only (profiling: keeps the per-kernel statistics unmixed)
  const int inner = 20;
  const size_t tile = 1024, set_elems = (size_t)nb * tile;
  const int nsets = std::max(2, (int)((640ull << 20) / (set_elems * 4 * 3)) + 1);
  std::vector<float*> A(nsets), B(nsets), C(nsets);
  std::vector<float> ha(set_elems), hb(set_elems);
  for (size_t i = 0; i < set_elems; ++i) { ha[i] = (float)((int)(i * 7919u % 10) - 4) / 10.0f; hb[i] = (float)((int)(i * 104729u % 10) - 4) / 10.0f; }
  for (int s = 0; s < nsets; ++s) {
    CHECK(hipMalloc(&A[s], set_elems * 4)); CHECK(hipMalloc(&B[s], set_elems * 4)); CHECK(hipMalloc(&C[s], set_elems * 4));
    CHECK(hipMemcpy(A[s], ha.data(), set_elems * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(B[s], hb.data(), set_elems * 4, hipMemcpyHostToDevice));
  }
  const Variant vs[] = {
    {"copy_tile", 256, 8, copy_tile<false, false>, false, 0},
    {"copy_tile_nts", 256, 8, copy_tile<false, true>, false, 0},
    {"copy_tile_ntls", 256, 8, copy_tile<true, true>, false, 0},
    {"copy_tile_xcd", 256, 8, copy_tile_xcd<false, false>, false, 0},
    {"copy_tile_xcd_ntls", 256, 8, copy_tile_xcd<true, true>, false, 0},
    {"copy_fine", 256, 2, copy_fine, false, 0},
    {"copy_persist512", 256, 8, copy_persist, false, 512},
    {"copy_persist1024", 256, 8, copy_persist, false, 1024},
    {"gemm_occ<4,16>", 256, 8, gemm_occ<4, 16, false>, true, 0},
    {"gemm_occ<4,16,ntl>", 256, 8, gemm_occ<4, 16, true>, true, 0},
    {"gemm_occ<4,12>", 256, 8, gemm_occ<4, 12, false>, true, 0},
    {"gemm_occ<4,8>", 256, 8, gemm_occ<4, 8, false>, true, 0},
    {"gemm_occ<4,4>", 256, 8, gemm_occ<4, 4, false>, true, 0},
    {"gemm_occ<1,16>", 64, 2, gemm_occ<1, 16, false>, true, 0},
    {"gemm_occ<1,12>", 64, 2, gemm_occ<1, 12, false>, true, 0},
    {"gemm_occ<1,8>", 64, 2, gemm_occ<1, 8, false>, true, 0},
    {"gemm_occ<2,16>", 128, 4, gemm_occ<2, 16, false>, true, 0},
    {"gemm_occ<2,8>", 128, 4, gemm_occ<2, 8, false>, true, 0},
    {"gemm_two<8>", 256, 16, gemm_two<8>, true, 0},
    {"gemm_two<4>", 256, 16, gemm_two<4>, true, 0},
    {"gemm_half", 256, 4, gemm_half, true, 0},
    {"LIBRARY", 0, 0, nullptr, true, 0},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]);
  {
    const libxsmm_gemm_shape sh = libxsmm_create_gemm_shape(32, 32, 32, 32, 32, 32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32, LIBXSMM_DATATYPE_F32);
    g_lib_kernel = libxsmm_dispatch_brgemm(sh, LIBXSMM_GEMM_FLAG_BETA_0, 0, libxsmm_create_gemm_batch_reduce_config(LIBXSMM_GEMM_BATCH_REDUCE_STRIDE, 4096, 4096, 0));
    if (!g_lib_kernel) { printf("library dispatch failed\n"); return 1; }
    libxsmm_hip_set_stream(nullptr);
  }
  auto launch = [&](int v, int s) {
    if (!vs[v].fn) { lib_launch(A[s], B[s], C[s], nb); return; }
    const int blocks = vs[v].fixed_blocks ? vs[v].fixed_blocks : (2 * nb + vs[v].problems_per_block_x2 - 1) / vs[v].problems_per_block_x2;
    hipLaunchKernelGGL(vs[v].fn, dim3(blocks), dim3(vs[v].threads), 0, 0, A[s], B[s], C[s], nb);
  };
  // correctness: every GEMM variant must reproduce the library bit for bit (same k-ordered accumulation)
  std::vector<float> ref(set_elems), got(set_elems);
  lib_launch(A[0], B[0], C[0], nb); CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(ref.data(), C[0], set_elems * 4, hipMemcpyDeviceToHost));
  for (int v = 0; v < nv; ++v) {
    if (!vs[v].is_gemm || !vs[v].fn) continue;
    CHECK(hipMemset(C[0], 0xff, set_elems * 4));
    launch(v, 0); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(got.data(), C[0], set_elems * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < set_elems; ++i) if (memcmp(&got[i], &ref[i], 4) != 0) ++bad;
    printf("check %-20s mismatches=%zu\n", vs[v].name, bad);
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<std::vector<float>> us(nv);
  for (int mode = 0; mode < nmodes; ++mode) {          // 0: rotate sets (HBM), 1: same set (L3 resident)
    for (auto& u : us) u.clear();
    for (int r = 0; r < rounds; ++r) {
      for (int v = 0; v < nv; ++v) {
        CHECK(hipEventRecord(e0, 0));
        for (int it = 0; it < inner; ++it) launch(v, mode == 0 ? (r * inner + it) % nsets : 0);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        us[v].push_back(ms * 1000.0f / inner);
      }
    }
    printf("---- nb=%d %s (event time per launch incl. the launch gap; GB/s algorithmic = 12 KiB per problem)\n", nb, mode == 0 ? "HBM-cold (rotating sets)" : "L3-resident");
    for (int v = 0; v < nv; ++v) {
      std::sort(us[v].begin(), us[v].end());
      const float med = us[v][us[v].size() / 2], mn = us[v][0];
      printf("%-20s median %7.2f us  min %7.2f us  -> %7.1f GB/s  frac %.3f\n", vs[v].name, med, mn, (double)nb * 12288.0 / med / 1e3, (double)nb * 12288.0 / med / 1e3 / 8000.0);
    }
  }
  return 0;
}
