// bperm_probe.hip -- what a ds_bpermute_b32 / v_readlane_b32 / buffer_load_ushort costs when every SIMD issues them back to back:
//   hipcc --offload-arch=gfx950 -O3 tools/bperm_probe.hip -o tools/bperm_probe && tools/bperm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned int* out, const unsigned short* src, int iters) {
  const unsigned int lane = threadIdx.x & 63u;
  unsigned int v[8];
  for (int e = 0; e < 8; ++e) v[e] = lane * 7u + e;
  unsigned int acc = 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 1 << 24, 0x00020000);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (MODE == 0) v[e] = (unsigned int)__builtin_amdgcn_ds_bpermute((int)((v[e] & 63u) << 2), (int)(v[e] + acc));                         // dependent chain per register, 8 independent
      else if (MODE == 1) acc += (unsigned int)__builtin_amdgcn_readlane((int)v[e], (e * 8 + it) & 63);
      else if (MODE == 2) v[e] += (unsigned int)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)(lane * 2u), (int)(((blockIdx.x * 16 + (threadIdx.x >> 6)) * 4096 + it * 1024 + e * 128) & ((1 << 24) - 256)), 0);
      else v[e] = __builtin_amdgcn_mbcnt_hi(v[e], __builtin_amdgcn_mbcnt_lo(acc + e, 0u)) + (v[e] << 2);
    }
  }
  for (int e = 0; e < 8; ++e) acc += v[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> static void run(const char* name, unsigned int* out, unsigned short* src, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<256, 1024>>>(out, src, iters); hipDeviceSynchronize();
  hipEventRecord(a); probe<MODE><<<256, 1024>>>(out, src, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double per = ms * 1e-3 * 2.4e9 / ((double)iters * 8.0 * 4.0);       // cycles per instruction and SIMD (4 waves per SIMD issue it in turn)
  printf("%-28s %8.1f us  %6.1f cycles per wave-instruction per SIMD (16 waves / CU)\n", name, ms * 1e3, per);
}
int main() {
  unsigned int* out; unsigned short* src;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&src, 1 << 24); hipMemset(src, 1, 1 << 24);
  run<0>("ds_bpermute_b32", out, src, 2000);
  run<1>("v_readlane_b32", out, src, 2000);
  run<2>("buffer_load_ushort (L2)", out, src, 500);
  run<3>("v_mbcnt pair + shl_add", out, src, 2000);
  return 0;
}
