// policy_probe.hip -- which cache policy should the streaming kernels put on their operand loads and C stores?
// The headline footprint (32 MiB read + 16 MiB written per launch) is run as a copy kernel with every combination of the gfx950
// cache-policy bits (sc0, nt, sc1) on the 16-byte buffer loads / stores, in three data states:
//   cold     : inputs rotate over > 2x the 256 MiB Infinity Cache (every launch streams from HBM),
//   resident : the same set every launch, after it has been touched with PLAIN loads (it is in the Infinity Cache),
//   self     : the Infinity Cache is flushed, then the variant alone runs 30 times on one set: the last 20 launches show
//              whether the policy lets a re-read working set BECOME resident by itself.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/policy_probe.hip -o tools/policy_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, -1, 0x00020000);
}
// aux bits on gfx940+: 1 = sc0, 2 = nt, 16 = sc1
template <int LA, int SA>
__global__ __launch_bounds__(256) void copy_pol(const float* A, const float* B, float* C, int nb) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= nb) return;
  const __amdgpu_buffer_rsrc_t ra = rsrc_of(A + (size_t)wid * 1024), rb = rsrc_of(B + (size_t)wid * 1024), rc = rsrc_of(C + (size_t)wid * 1024);
  u32x4 va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, (lane + 64 * q) * 16, 0, LA); vb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, (lane + 64 * q) * 16, 0, LA); }
#pragma unroll
  for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(va[q] ^ vb[q], rc, (lane + 64 * q) * 16, 0, SA);
}
__global__ __launch_bounds__(256) void touch_plain(const float* A, const float* B, float* C, int nb) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)nb * 256) return;
  ((f32x4*)C)[i] = ((const f32x4*)A)[i] + ((const f32x4*)B)[i];
}

// ---- the headline GEMM (f32 32x32x32, one problem per wave; algorithm of gemm_f32_stream_kernel_lean) with the same policy bits:
// LA on the operand loads, SA on the C stores; CST = 1 passes the C tile through LDS so that it leaves as 4 x 16-byte stores per lane
// instead of 16 dword stores (policy bits on sub-16-byte stores cost one fabric write each, MI355X_MICROARCH.md).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int jl_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
template <int LA, int SA, int CST>
__global__ __launch_bounds__(256) void gemm_pol(const float* A, const float* B, float* C, int nb) {
  __shared__ __attribute__((aligned(16))) float lds[4][2048];
  const int w = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + w, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  if (wid >= nb) return;
  const __amdgpu_buffer_rsrc_t ra = rsrc_of(A + (size_t)wid * 1024), rb = rsrc_of(B + (size_t)wid * 1024), rc = rsrc_of(C + (size_t)wid * 1024);
  float* la = lds[w]; float* lb = la + 1024;
  u32x4 va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, (lane + 64 * q) * 16, 0, LA); vb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, (lane + 64 * q) * 16, 0, LA); }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ((u32x4*)la)[lane + 64 * q] = va[q];
    const int t = lane + 64 * q, j = t >> 3, cc = (t & 7) ^ ((j >> 1) & 7);
    ((u32x4*)lb)[j * 8 + cc] = vb[q];
  }
  float af[16], bf[16], v[16];
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
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
  if (CST == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), rc, (li + jl_of(r, h) * 32) * 4, 0, SA);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) la[li + jl_of(r, h) * 32] = acc[r];      // column-major image of C, conflict free
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(((const u32x4*)la)[lane + 64 * q], rc, (lane + 64 * q) * 16, 0, SA);
  }
}
#define G(LA, SA, CST, NAME) {NAME, gemm_pol<LA, SA, CST>}

typedef void (*kfn)(const float*, const float*, float*, int);
struct Variant { const char* name; kfn fn; };
#define V(LA, SA, NAME) {NAME, copy_pol<LA, SA>}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 4096;
  const int rounds = argc > 2 ? atoi(argv[2]) : 10;
  const int inner = 20;
  const size_t set_elems = (size_t)nb * 1024;
  const int nsets = std::max(2, (int)((640ull << 20) / (set_elems * 4 * 3)) + 1);
  std::vector<float*> A(nsets), B(nsets), C(nsets);
  std::vector<float> h(set_elems);
  for (size_t i = 0; i < set_elems; ++i) h[i] = (float)((int)(i * 7919u % 10) - 4) / 10.0f;
  for (int s = 0; s < nsets; ++s) {
    CHECK(hipMalloc(&A[s], set_elems * 4)); CHECK(hipMalloc(&B[s], set_elems * 4)); CHECK(hipMalloc(&C[s], set_elems * 4));
    CHECK(hipMemcpy(A[s], h.data(), set_elems * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(B[s], h.data(), set_elems * 4, hipMemcpyHostToDevice));
  }
  const bool gemm_mode = argc > 3 && atoi(argv[3]) == 1;
  const Variant vcopy[] = {
    V(0, 0, "ld plain   st plain"), V(0, 2, "ld plain   st nt"), V(0, 16, "ld plain   st sc1"), V(0, 17, "ld plain   st sc0sc1"), V(0, 18, "ld plain   st sc1nt"), V(0, 3, "ld plain   st sc0nt"),
    V(2, 2, "ld nt      st nt"), V(16, 2, "ld sc1     st nt"), V(17, 2, "ld sc0sc1  st nt"), V(18, 2, "ld sc1nt   st nt"), V(3, 2, "ld sc0nt   st nt"), V(1, 2, "ld sc0     st nt"),
    V(2, 0, "ld nt      st plain"), V(2, 18, "ld nt      st sc1nt"), V(18, 18, "ld sc1nt   st sc1nt"), V(2, 16, "ld nt      st sc1"),
    V(17, 16, "ld sc0sc1  st sc1"), V(17, 3, "ld sc0sc1  st sc0nt"), V(17, 17, "ld sc0sc1  st sc0sc1"), V(17, 0, "ld sc0sc1  st plain"), V(1, 16, "ld sc0     st sc1"), V(16, 16, "ld sc1     st sc1"),
    V(3, 16, "ld sc0nt   st sc1"), V(19, 2, "ld sc0sc1nt st nt"), V(19, 16, "ld sc0sc1nt st sc1"), V(17, 19, "ld sc0sc1  st sc0sc1nt"),
  };
  const Variant vgemm[] = {
    G(0, 2, 0, "g ld plain  st nt    dw"), G(0, 16, 0, "g ld plain  st sc1   dw"), G(0, 3, 0, "g ld plain  st sc0nt dw"), G(0, 0, 0, "g ld plain  st plain dw"),
    G(0, 2, 1, "g ld plain  st nt    x4"), G(0, 16, 1, "g ld plain  st sc1   x4"), G(0, 3, 1, "g ld plain  st sc0nt x4"), G(0, 0, 1, "g ld plain  st plain x4"),
    G(17, 2, 0, "g ld sc0sc1 st nt    dw"), G(17, 16, 0, "g ld sc0sc1 st sc1   dw"), G(17, 3, 0, "g ld sc0sc1 st sc0nt dw"),
    G(17, 2, 1, "g ld sc0sc1 st nt    x4"), G(17, 16, 1, "g ld sc0sc1 st sc1   x4"), G(17, 3, 1, "g ld sc0sc1 st sc0nt x4"),
    G(2, 2, 0, "g ld nt     st nt    dw"), G(2, 16, 0, "g ld nt     st sc1   dw"), G(2, 2, 1, "g ld nt     st nt    x4"), G(2, 16, 1, "g ld nt     st sc1   x4"),
    G(3, 2, 0, "g ld sc0nt  st nt    dw"), G(3, 16, 0, "g ld sc0nt  st sc1   dw"), G(3, 16, 1, "g ld sc0nt  st sc1   x4"), G(3, 3, 1, "g ld sc0nt  st sc0nt x4"),
  };
  const Variant* vs = gemm_mode ? vgemm : vcopy;
  const int nv = gemm_mode ? (int)(sizeof(vgemm) / sizeof(vgemm[0])) : (int)(sizeof(vcopy) / sizeof(vcopy[0]));
  const int blocks = (nb + 3) / 4;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto time_launches = [&](int v, int first_set, bool rotate, int n) {
    CHECK(hipEventRecord(e0, 0));
    for (int it = 0; it < n; ++it) { const int s = rotate ? (first_set + it) % nsets : first_set; hipLaunchKernelGGL(vs[v].fn, dim3(blocks), dim3(256), 0, 0, A[s], B[s], C[s], nb); }
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0f / n;
  };
  auto flush = [&]() { for (int s = 1; s < nsets; ++s) hipLaunchKernelGGL(touch_plain, dim3(nb), dim3(256), 0, 0, A[s], B[s], C[s], nb); CHECK(hipDeviceSynchronize()); };
  std::vector<std::vector<float>> cold(nv), res(nv), self(nv);
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < nv; ++v) {
      cold[v].push_back(time_launches(v, r * inner, true, inner));
      hipLaunchKernelGGL(touch_plain, dim3(nb), dim3(256), 0, 0, A[0], B[0], C[0], nb);      // make set 0 resident with plain accesses
      hipLaunchKernelGGL(touch_plain, dim3(nb), dim3(256), 0, 0, A[0], B[0], C[0], nb);
      res[v].push_back(time_launches(v, 0, false, inner));
      if (r < 3) { flush(); (void)time_launches(v, 0, false, 10); self[v].push_back(time_launches(v, 0, false, inner)); }
    }
  printf("nb=%d: us per launch (event time incl. launch gap), median over rounds; footprint 48 MiB per launch\n", nb);
  printf("%-22s %10s %10s %10s\n", "policy", "cold", "resident", "self");
  for (int v = 0; v < nv; ++v) {
    std::sort(cold[v].begin(), cold[v].end()); std::sort(res[v].begin(), res[v].end()); std::sort(self[v].begin(), self[v].end());
    printf("%-22s %10.2f %10.2f %10.2f\n", vs[v].name, cold[v][cold[v].size() / 2], res[v][res[v].size() / 2], self[v][self[v].size() / 2]);
  }
  return 0;
}
