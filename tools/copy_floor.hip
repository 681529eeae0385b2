// copy_floor.hip -- what the memory system gives a kernel that does NOTHING but move a workload's bytes: R read streams and W write streams of a given size, 16 bytes per
// lane, whole lines, as many bytes in flight as the register file allows.  The fraction of the 8 TB/s figure this reaches is the roof of every streaming kernel with the
// same read : write mix and footprint (round 6: the configs of BASELINE.json against their own copy floors, profiles/r06_copy_floor.txt).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/copy_floor.hip -o tools/copy_floor
//   tools/copy_floor  <name> <reads> <writes> <MiB per stream> [<name> ...]      (sets rotate so that at least 1 GiB is touched between two uses of a buffer)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GM __attribute__((address_space(1)))
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Streams { const u32x4* in[4]; u32x4* out[4]; };

// one workgroup moves consecutive 4 KiB pieces (256 lanes x 16 bytes) of every stream, U pieces per stream in flight; NT: non-temporal loads and stores
template <int NR, int NW, int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(Streams s, unsigned long long pieces) {
  const unsigned long long first = (unsigned long long)blockIdx.x * U;
  if (first >= pieces) return;
  u32x4 v[NR > 0 ? NR : 1][U];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long p = std::min(first + u, pieces - 1);
      GM const u32x4* src = (GM const u32x4*)s.in[r] + p * 256ull + threadIdx.x;
      v[r][u] = NT ? __builtin_nontemporal_load(src) : *src;
    }
  if (NW == 0) {            // read only: the loaded values reach a store that never executes
    u32x4 x = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) x ^= v[r][u];
    if (x[0] == 0x12345678u && x[1] == 0x9abcdef0u && x[2] == 0x0fedcba9u) *((GM u32x4*)s.in[0] + threadIdx.x) = x;
  }
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (first + u >= pieces) continue;
      u32x4 x = (u32x4){1u + (unsigned int)w, 2u, 3u, 4u};
#pragma unroll
      for (int r = 0; r < NR; ++r) x ^= v[r][u];                 // every read stream reaches a store: none of the loads is dead
      GM u32x4* dst = (GM u32x4*)s.out[w] + (first + u) * 256ull + threadIdx.x;
      if (NT) __builtin_nontemporal_store(x, dst); else *dst = x;
    }
}

// the packed sparse kernels' pattern: a lane owns 16 bytes of a COLUMN chunk and touches it in every one of R input rows and W output rows (row pitch = the row length):
// R + W streams a whole row apart instead of 2-3 contiguous ones
template <int R, int W, bool NT, int BS = 256>
__global__ __launch_bounds__(BS) void rows_kernel(const u32x4* in, u32x4* out, unsigned long long row_vec) {
  const unsigned long long c = (unsigned long long)blockIdx.x * BS + threadIdx.x;
  if (c >= row_vec) return;
  u32x4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { GM const u32x4* src = (GM const u32x4*)in + (unsigned long long)r * row_vec + c; v[r] = NT ? __builtin_nontemporal_load(src) : *src; }
#pragma unroll
  for (int w = 0; w < W; ++w) {
    u32x4 x = v[w % R] ^ v[(w + 7) % R];
    GM u32x4* dst = (GM u32x4*)out + (unsigned long long)w * row_vec + c;
    if (NT) __builtin_nontemporal_store(x, dst); else *dst = x;
  }
}

// the BCSC streaming kernel's pattern (config #4): `waves` waves; wave w walks M-blocks w, w + waves, ...; an M-block = eight 4 KiB chunks of A of which seven are read
// (the 2 : 8 pattern touches 7 of 8 K-blocks), three chunks in flight per wave, then 8 KiB of C are written
template <int NT>          // bit 0: non-temporal loads, bit 1: non-temporal stores
__global__ __launch_bounds__(256) void bcsc_pattern_kernel(const u32x4* a, u32x4* c, unsigned int m_blocks, unsigned int waves) {
  const unsigned int w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (w >= waves) return;
  u32x4 acc = (u32x4){0u, 0u, 0u, 0u};
  for (unsigned int mb = w; mb < m_blocks; mb += waves) {
    GM const u32x4* src = (GM const u32x4*)a + (unsigned long long)mb * 2048ull + lane;       // 32 KiB per M-block
    u32x4 v[3][4];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int x = 0; x < 4; ++x) { GM const u32x4* q = src + (ch * 4 + x) * 64; v[ch][x] = (NT & 1) ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) {
#pragma unroll
      for (int x = 0; x < 4; ++x) acc ^= v[ch % 3][x];
      if (ch + 3 < 7) {
#pragma unroll
        for (int x = 0; x < 4; ++x) { GM const u32x4* q = src + ((ch + 3) * 4 + x) * 64; v[ch % 3][x] = (NT & 1) ? __builtin_nontemporal_load(q) : *q; }
      }
    }
    GM u32x4* dst = (GM u32x4*)c + (unsigned long long)mb * 512ull + lane;                     // 8 KiB per M-block
#pragma unroll
    for (int x = 0; x < 8; ++x) { u32x4 o = acc; o[0] += (unsigned int)x; if (NT & 2) __builtin_nontemporal_store(o, dst + x * 64); else dst[x * 64] = o; }
  }
}

// the same pattern with the reads as LDS-DMA requests (global_load_lds, 16 bytes per lane, a ring of three 4 KiB chunks per wave in LDS; nothing reads the LDS: what
// the requests themselves cost).  s_waitcnt counts in issue order: a chunk is waited for with the two younger chunks (and, on a tile's first chunks, the 8 stores) behind it.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int AUX, bool NTS>
__global__ __launch_bounds__(256) void bcsc_pattern_dma_kernel(const u32x4* a, u32x4* c, unsigned int m_blocks, unsigned int waves) {
  __shared__ __attribute__((aligned(16))) unsigned int ring[4][3][1024];
  const unsigned int wv = threadIdx.x >> 6, w = blockIdx.x * 4u + wv, lane = threadIdx.x & 63u;
  if (w >= waves) return;
  unsigned int* r = ring[__builtin_amdgcn_readfirstlane((int)wv)][0];
  const unsigned int tiles = (m_blocks - w + waves - 1u) / waves, total = tiles * 7u;
  auto issue = [&](unsigned int f) __attribute__((always_inline)) {
    const unsigned int t = f / 7u, ch = f - 7u * t;
    GM const u32x4* src = (GM const u32x4*)a + (unsigned long long)(w + t * waves) * 2048ull + ch * 256u + lane;
#pragma unroll
    for (int x = 0; x < 4; ++x) __builtin_amdgcn_global_load_lds((GM const void*)(src + 64 * x), (lds_ptr_t)((char*)r + 4096u * (f % 3u) + 1024 * x), 16, 0, AUX);
  };
  for (unsigned int f = 0; f < 3u && f < total; ++f) issue(f);
  unsigned int cc = 0, cj = 0;
  for (unsigned int f = 0; f < total; ++f) {
    const unsigned int left = total - 1u - f;
    const bool stored = cj > 0 && cc < 3;
    if (left >= 2) { if (stored) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else if (left == 1) { if (stored) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else if (stored) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (left >= 3) issue(f + 3u);
    if (++cc == 7u) {
      GM u32x4* dst = (GM u32x4*)c + (unsigned long long)(w + cj * waves) * 512ull + lane;
#pragma unroll
      for (int x = 0; x < 8; ++x) { u32x4 o = (u32x4){f, lane, (unsigned int)x, cj}; if (NTS) __builtin_nontemporal_store(o, dst + x * 64); else dst[x * 64] = o; }
      cc = 0; ++cj;
    }
  }
}

template <int NR, int NW, bool NT> static void launch(const Streams& s, unsigned long long pieces, hipStream_t st) {
  constexpr int U = 4;
  hipLaunchKernelGGL((copy_kernel<NR, NW, U, NT>), dim3((unsigned int)((pieces + U - 1) / U)), dim3(256), 0, st, s, pieces);
}
static void launch_any(int nr, int nw, bool nt, const Streams& s, unsigned long long pieces, hipStream_t st) {
#define C_(R_, W_) if (nr == R_ && nw == W_) { if (nt) launch<R_, W_, true>(s, pieces, st); else launch<R_, W_, false>(s, pieces, st); return; }
  C_(1, 1) C_(2, 1) C_(4, 1) C_(1, 0) C_(0, 1) C_(3, 1)
#undef C_
  printf("unsupported mix %d:%d\n", nr, nw); exit(1);
}

int main(int argc, char** argv) {
  if (argc < 5 || (argc - 1) % 4) { printf("usage: copy_floor <name> <reads> <writes> <MiB per stream> ...\n"); return 1; }
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int a = 1; a + 3 < argc; a += 4) {
    const char* name = argv[a]; const int nr = atoi(argv[a + 1]), nw = atoi(argv[a + 2]); const double mib = atof(argv[a + 3]);
    if (nr == 7 && nw == 1) {             // "<name> 7 1 <M-blocks>": the BCSC streaming pattern, 2048 and 4096 waves
      const unsigned int m_blocks = (unsigned int)mib;
      const size_t a_bytes = (size_t)m_blocks * 32768, c_bytes = (size_t)m_blocks * 8192;
      const double moved = (double)m_blocks * (7.0 * 4096 + 8192);
      const int ns = (int)std::max<size_t>(2, ((size_t)1 << 30) / (a_bytes + c_bytes) + 1);
      std::vector<void*> as_((size_t)ns), cs_((size_t)ns);
      for (int q = 0; q < ns; ++q) { CHECK(hipMalloc(&as_[(size_t)q], a_bytes)); CHECK(hipMemset(as_[(size_t)q], 0x31, a_bytes)); CHECK(hipMalloc(&cs_[(size_t)q], c_bytes)); }
      for (unsigned int waves = 2048; waves <= 2048; waves *= 2)      // (4096 and 8192 waves measure the same: profiles/r06_copy_floor.txt history)
        for (int nt = 0; nt < 6; ++nt) {
          auto go = [&](int i) {
            const u32x4* ap = (const u32x4*)as_[(size_t)(i % ns)]; u32x4* cp = (u32x4*)cs_[(size_t)(i % ns)];
            if (nt == 0) hipLaunchKernelGGL((bcsc_pattern_kernel<0>), dim3(waves / 4), dim3(256), 0, st, ap, cp, m_blocks, waves);
            else if (nt == 1) hipLaunchKernelGGL((bcsc_pattern_kernel<1>), dim3(waves / 4), dim3(256), 0, st, ap, cp, m_blocks, waves);
            else if (nt == 2) hipLaunchKernelGGL((bcsc_pattern_kernel<2>), dim3(waves / 4), dim3(256), 0, st, ap, cp, m_blocks, waves);
            else if (nt == 3) hipLaunchKernelGGL((bcsc_pattern_kernel<3>), dim3(waves / 4), dim3(256), 0, st, ap, cp, m_blocks, waves);
            else if (nt == 4) hipLaunchKernelGGL((bcsc_pattern_dma_kernel<0, false>), dim3(waves / 4), dim3(256), 0, st, ap, cp, m_blocks, waves);
            else hipLaunchKernelGGL((bcsc_pattern_dma_kernel<2, true>), dim3(waves / 4), dim3(256), 0, st, ap, cp, m_blocks, waves); };
          for (int i = 0; i < 2 * ns; ++i) go(i);
          CHECK(hipStreamSynchronize(st));
          const int reps = std::max(3 * ns, (int)(0.05 / (moved / 5e12)));
          CHECK(hipEventRecord(e0, st));
          for (int i = 0; i < reps; ++i) go(i);
          CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
          float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
          const double us = ms * 1e3 / reps, tbs = moved / us / 1e6;
          printf("{\"copy_floor\": \"%s\", \"pattern\": \"bcsc 7 of 8 chunks in, 8 KiB out per M-block\", \"m_blocks\": %u, \"waves\": %u, \"sets\": %d, \"policy\": \"%s\", \"us\": %.2f, \"TB/s\": %.3f, \"frac_of_8TBs\": %.4f}\n",
                 name, m_blocks, waves, ns, nt == 0 ? "default" : nt == 1 ? "nt loads" : nt == 2 ? "nt stores" : nt == 3 ? "nt" : nt == 4 ? "LDS-DMA reads, default" : "LDS-DMA reads, nt", us, tbs, tbs / 8.0);
        }
      for (int q = 0; q < ns; ++q) { CHECK(hipFree(as_[(size_t)q])); CHECK(hipFree(cs_[(size_t)q])); }
      continue;
    }
    if (nr == 35 && nw == 35) {           // "<name> 35 35 <MiB per row>": the packed CSR / FsSpMDM access pattern (35 rows in, 35 rows out)
      const unsigned long long row_vec = (unsigned long long)(mib * 65536.0);
      const size_t row_bytes = (size_t)row_vec * 16, set_bytes = row_bytes * 70;
      const int ns = (int)std::max<size_t>(2, ((size_t)1 << 30) / set_bytes + 1);
      std::vector<void*> ins((size_t)ns), outs((size_t)ns);
      for (int q = 0; q < ns; ++q) { CHECK(hipMalloc(&ins[(size_t)q], row_bytes * 35)); CHECK(hipMemset(ins[(size_t)q], 0x21, row_bytes * 35)); CHECK(hipMalloc(&outs[(size_t)q], row_bytes * 35)); }
      for (int nt = 0; nt < 2; ++nt) {
        static const int bs = []() { const char* e = getenv("ROWS_BLOCK"); return e ? atoi(e) : 256; }();      // experiment: threads per workgroup (64 / 256 / 1024)
        const dim3 grid((unsigned int)((row_vec + bs - 1) / bs));
        auto go = [&](int i) { const u32x4* a_ = (const u32x4*)ins[(size_t)(i % ns)]; u32x4* o_ = (u32x4*)outs[(size_t)(i % ns)];
          if (bs == 1024) { if (nt) hipLaunchKernelGGL((rows_kernel<35, 35, true, 1024>), grid, dim3(1024), 0, st, a_, o_, row_vec); else hipLaunchKernelGGL((rows_kernel<35, 35, false, 1024>), grid, dim3(1024), 0, st, a_, o_, row_vec); }
          else if (bs == 64) { if (nt) hipLaunchKernelGGL((rows_kernel<35, 35, true, 64>), grid, dim3(64), 0, st, a_, o_, row_vec); else hipLaunchKernelGGL((rows_kernel<35, 35, false, 64>), grid, dim3(64), 0, st, a_, o_, row_vec); }
          else { if (nt) hipLaunchKernelGGL((rows_kernel<35, 35, true>), grid, dim3(256), 0, st, a_, o_, row_vec); else hipLaunchKernelGGL((rows_kernel<35, 35, false>), grid, dim3(256), 0, st, a_, o_, row_vec); } };
        for (int i = 0; i < 2 * ns; ++i) go(i);
        CHECK(hipStreamSynchronize(st));
        const int reps = std::max(3 * ns, (int)(0.05 / (set_bytes / 5e12)));
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) go(i);
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tbs = set_bytes / us / 1e6;
        printf("{\"copy_floor\": \"%s\", \"reads\": 35, \"writes\": 35, \"MiB_per_stream\": %.1f, \"sets\": %d, \"policy\": \"%s\", \"us\": %.2f, \"TB/s\": %.3f, \"frac_of_8TBs\": %.4f}\n",
               name, mib, ns, nt ? "nt" : "default", us, tbs, tbs / 8.0);
      }
      for (int q = 0; q < ns; ++q) { CHECK(hipFree(ins[(size_t)q])); CHECK(hipFree(outs[(size_t)q])); }
      continue;
    }
    const unsigned long long pieces = (unsigned long long)(mib * 256.0);                       // 4 KiB pieces per stream
    const size_t bytes = (size_t)pieces * 4096;
    const size_t per_set = bytes * (size_t)(nr + nw);
    const int nsets = (int)std::max<size_t>(2, ((size_t)1 << 30) / per_set + 1);
    std::vector<Streams> sets((size_t)nsets);
    std::vector<void*> all;
    for (int q = 0; q < nsets; ++q) {
      memset(&sets[(size_t)q], 0, sizeof(Streams));
      for (int r = 0; r < nr; ++r) { void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0x11 * (r + 1), bytes)); sets[(size_t)q].in[r] = (const u32x4*)p; all.push_back(p); }
      for (int w = 0; w < nw; ++w) { void* p; CHECK(hipMalloc(&p, bytes)); sets[(size_t)q].out[w] = (u32x4*)p; all.push_back(p); }
    }
    for (int nt = 0; nt < 2; ++nt) {
      for (int i = 0; i < 2 * nsets; ++i) launch_any(nr, nw, nt != 0, sets[(size_t)(i % nsets)], pieces, st);
      CHECK(hipStreamSynchronize(st));
      const int reps = std::max(3 * nsets, (int)(0.05 / (per_set / 5e12)));
      CHECK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) launch_any(nr, nw, nt != 0, sets[(size_t)(i % nsets)], pieces, st);
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps, tbs = per_set / us / 1e6;
      printf("{\"copy_floor\": \"%s\", \"reads\": %d, \"writes\": %d, \"MiB_per_stream\": %.1f, \"sets\": %d, \"policy\": \"%s\", \"us\": %.2f, \"TB/s\": %.3f, \"frac_of_8TBs\": %.4f}\n",
             name, nr, nw, mib, nsets, nt ? "nt" : "default", us, tbs, tbs / 8.0);
    }
    for (void* p : all) CHECK(hipFree(p));
  }
  return 0;
}
