// width_probe.hip -- what does the ACCESS WIDTH cost a streaming kernel over small contiguous blocks?
// nb blocks of `dw` dwords in each of three arrays (A, B read; C written), one wave per block, all loads of a block issued before the
// first store: exactly the traffic of a batched small GEMM without its arithmetic.  VEC = 1: dword accesses (any block size, any
// alignment: what the ragged kernel does), VEC = 4: 16-byte accesses (block size a multiple of 4 dwords, 16-byte aligned).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/width_probe.hip -o tools/width_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base, unsigned int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)bytes, 0x00020000);
}
template <int VEC, int R, int AUX>
__global__ __launch_bounds__(256) void copy_blocks(const unsigned int* A, const unsigned int* B, unsigned int* C, unsigned int nb, unsigned int dw) {
  const unsigned int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= nb) return;
  const size_t off = (size_t)wid * dw;
  const __amdgpu_buffer_rsrc_t ra = rsrc_of(A + off, dw * 4), rb = rsrc_of(B + off, dw * 4), rc = rsrc_of(C + off, dw * 4);
  if (VEC == 1) {
    unsigned int va[R], vb[R];
#pragma unroll
    for (int q = 0; q < R; ++q) { va[q] = __builtin_amdgcn_raw_buffer_load_b32(ra, (lane + 64 * q) * 4, 0, AUX); vb[q] = __builtin_amdgcn_raw_buffer_load_b32(rb, (lane + 64 * q) * 4, 0, AUX); }
#pragma unroll
    for (int q = 0; q < R; ++q) __builtin_amdgcn_raw_buffer_store_b32(va[q] ^ vb[q], rc, (lane + 64 * q) * 4, 0, 2);
  } else {
    u32x4 va[R], vb[R];
#pragma unroll
    for (int q = 0; q < R; ++q) { va[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, (lane + 64 * q) * 16, 0, AUX); vb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, (lane + 64 * q) * 16, 0, AUX); }
#pragma unroll
    for (int q = 0; q < R; ++q) __builtin_amdgcn_raw_buffer_store_b128(va[q] ^ vb[q], rc, (lane + 64 * q) * 16, 0, 2);
  }
}
// the ragged kernel's access pattern without its arithmetic: blocks are m x m (column = m dwords), a load / store instruction covers
// 64 / m whole columns (m = 23: 46 of 64 lanes, 184 contiguous bytes); PERSIST: grid-stride loop of resident workgroups (one wave each)
template <int R, bool PERSIST>
__global__ __launch_bounds__(64) void copy_rounds(const unsigned int* A, const unsigned int* B, unsigned int* C, unsigned int nb, unsigned int m) {
  const unsigned int lane = threadIdx.x, cpr = 64u / m, sub = lane / m, i = lane - sub * m;
  const unsigned int vo = sub < cpr ? 4u * (sub * m + i) : 0x7ffffff0u, dw = m * m;
  for (unsigned int wid = blockIdx.x; wid < nb; wid += PERSIST ? gridDim.x : nb) {
    const size_t off = (size_t)wid * dw;
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(A + off, dw * 4), rb = rsrc_of(B + off, dw * 4), rc = rsrc_of(C + off, dw * 4);
    unsigned int va[R], vb[R];
#pragma unroll
    for (int q = 0; q < R; ++q) { va[q] = __builtin_amdgcn_raw_buffer_load_b32(ra, vo, q * cpr * m * 4, 0); vb[q] = __builtin_amdgcn_raw_buffer_load_b32(rb, vo, q * cpr * m * 4, 0); }
#pragma unroll
    for (int q = 0; q < R; ++q) __builtin_amdgcn_raw_buffer_store_b32(va[q] ^ vb[q], rc, vo, q * cpr * m * 4, 2);
  }
}
template <int R, bool PERSIST>
static void run_rounds(const char* name, unsigned int m, size_t total_bytes_per_array, int occ) {
  const unsigned int dw = m * m, nb = (unsigned int)(total_bytes_per_array / (dw * 4));
  const int nsets = 2;
  std::vector<unsigned int*> A(nsets), B(nsets), C(nsets);
  for (int s = 0; s < nsets; ++s) {
    CHECK(hipMalloc(&A[s], (size_t)nb * dw * 4 + 64)); CHECK(hipMalloc(&B[s], (size_t)nb * dw * 4 + 64)); CHECK(hipMalloc(&C[s], (size_t)nb * dw * 4 + 64));
    CHECK(hipMemset(A[s], 1, (size_t)nb * dw * 4)); CHECK(hipMemset(B[s], 2, (size_t)nb * dw * 4));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const dim3 grid(PERSIST ? 256u * occ : nb);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((copy_rounds<R, PERSIST>), grid, dim3(64), 0, 0, A[i % nsets], B[i % nsets], C[i % nsets], nb, m);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_rounds<R, PERSIST>), grid, dim3(64), 0, 0, A[i % nsets], B[i % nsets], C[i % nsets], nb, m);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps, gbs = 3.0 * nb * dw * 4 / (us * 1e-6) / 1e9;
  printf("%-44s blocks %8u x %5u B  %9.2f us  %8.1f GB/s  frac %.3f\n", name, nb, dw * 4, us, gbs, gbs / 8000.0);
  for (int s = 0; s < nsets; ++s) { CHECK(hipFree(A[s])); CHECK(hipFree(B[s])); CHECK(hipFree(C[s])); }
}
template <int VEC, int R, int AUX>
static void run(const char* name, unsigned int dw, size_t total_bytes_per_array) {
  const unsigned int nb = (unsigned int)(total_bytes_per_array / (dw * 4));
  const int nsets = 2;
  std::vector<unsigned int*> A(nsets), B(nsets), C(nsets);
  for (int s = 0; s < nsets; ++s) {
    CHECK(hipMalloc(&A[s], (size_t)nb * dw * 4 + 64)); CHECK(hipMalloc(&B[s], (size_t)nb * dw * 4 + 64)); CHECK(hipMalloc(&C[s], (size_t)nb * dw * 4 + 64));
    CHECK(hipMemset(A[s], 1, (size_t)nb * dw * 4)); CHECK(hipMemset(B[s], 2, (size_t)nb * dw * 4));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const dim3 grid((nb + 3) / 4);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((copy_blocks<VEC, R, AUX>), grid, dim3(256), 0, 0, A[i % nsets], B[i % nsets], C[i % nsets], nb, dw);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_blocks<VEC, R, AUX>), grid, dim3(256), 0, 0, A[i % nsets], B[i % nsets], C[i % nsets], nb, dw);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps, gbs = 3.0 * nb * dw * 4 / (us * 1e-6) / 1e9;
  printf("%-44s blocks %8u x %5u B  %9.2f us  %8.1f GB/s  frac %.3f\n", name, nb, dw * 4, us, gbs, gbs / 8000.0);
  for (int s = 0; s < nsets; ++s) { CHECK(hipFree(A[s])); CHECK(hipFree(B[s])); CHECK(hipFree(C[s])); }
}
int main() {
  const size_t big = 280u << 20;
  run_rounds<12, false>("rounds 23^2, one wave-workgroup per block", 23, big, 0);
  run_rounds<12, true>("rounds 23^2, persistent 8 / CU", 23, big, 8);
  run_rounds<12, true>("rounds 23^2, persistent 16 / CU", 23, big, 16);
  run_rounds<12, true>("rounds 23^2, persistent 24 / CU", 23, big, 24);
  run_rounds<4, false>("rounds 13^2, one wave-workgroup per block", 13, big, 0);
  run_rounds<4, true>("rounds 13^2, persistent 24 / CU", 13, big, 24);
  run<1, 3, 0>("dword  13^2 blocks (676 B)", 169, big);
  run<1, 9, 0>("dword  23^2 blocks (2116 B, 4-byte aligned)", 529, big);
  run<1, 9, 2>("dword  23^2 blocks, nt loads", 529, big);
  run<1, 9, 0>("dword  2112 B blocks (64-byte aligned)", 528, big);
  run<4, 3, 0>("x4     2112 B blocks", 528, big);
  run<4, 3, 2>("x4     2112 B blocks, nt loads", 528, big);
  run<1, 16, 0>("dword  32^2 blocks (4096 B)", 1024, big);
  run<4, 4, 0>("x4     32^2 blocks (4096 B)", 1024, big);
  run<4, 4, 2>("x4     32^2 blocks, nt loads", 1024, big);
  run<1, 25, 0>("dword  40^2 blocks (6400 B)", 1600, big);
  run<4, 7, 0>("x4     40^2 blocks (6400 B)", 1600, big);
  return 0;
}
