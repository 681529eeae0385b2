// What the bf16 matrix pipe of an MI355X sustains on THIS chip under THIS power budget: every wave (one per SIMD, 256 accumulator registers like the
// macro-tile GEMM kernel) issues v_mfma_f32_32x32x16_bf16 back to back on register operands -- no LDS, no memory.  Operands: zeros, or the value
// distribution of bench.py (multiples of 0.1 in [-0.4, 0.5], truncated to bf16), or random bit patterns.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 1) void mfma_loop(const bf16x8* __restrict__ ops, float* __restrict__ out, int iters) {
  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = ops[(threadIdx.x + 256 * i) & 4095]; b[i] = ops[(threadIdx.x + 256 * (i + 4)) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) acc[ti * 4 + tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[tj], a[ti], acc[ti * 4 + tj], 0, 0, 0);
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  const int blocks = 256, iters = 20000;             // 256 workgroups of 4 waves: one wave per SIMD on every CU
  std::vector<unsigned short> h(4096 * 8);
  bf16x8* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"zeros", "bench.py values (multiples of 0.1 in [-0.4, 0.5], bf16 by truncation)", "random bf16 in [-1, 1)"};
  for (int mode = 0; mode < 3; ++mode) {
    srand(555);
    for (auto& x : h) {
      float f = 0.0f;
      if (mode == 1) f = (float)((rand() % 10) - 4) / 10.0f;
      if (mode == 2) f = (float)rand() / (float)RAND_MAX * 2.0f - 1.0f;
      unsigned int u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16);
    }
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, o, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int launches = 10;
    for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * 16 * iters * launches, flops = mfmas * 32768.0;
    const double cyc_per_simd = 32.0 * 16 * iters * launches;          // 32 cycles per MFMA at full rate
    printf("{\"operands\": \"%s\", \"ms\": %.3f, \"TFLOP/s\": %.1f, \"pct_of_2500\": %.1f, \"implied_clock_GHz_if_pipe_full\": %.3f}\n",
           names[mode], ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100.0, cyc_per_simd / (ms * 1e-3) / 1e9);
  }
  return 0;
}
