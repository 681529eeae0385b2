// jit.cpp -- run-time specialisation of the fixed-pattern sparse kernels.
//
// The reference is a JIT: libxsmm_create_packed_spgemm_csr/_csc and libxsmm_create_spgemm_csr_areg emit machine code
// with the sparsity pattern unrolled into the instruction stream [ref: src/generator_packed_spgemm_csr_asparse_avx_
// avx2_avx512.c:336-470 (one FMA per non-zero, B rows addressed by immediate), src/generator_spgemm_csr_asparse_reg.c].
// The gfx950 analogue: generate HIP source for exactly this pattern and geometry, compile it with hiprtc for the
// device's ISA and load it as a module.  What specialisation buys on this machine:
//   * every X row a lane needs is fetched by ONE vector load into registers, all of them in flight at once
//     (the pattern tells which rows are touched; untouched rows are never read);
//   * a non-zero costs exactly one v_(pk_)fma: the X operand is a register picked at code-generation time, the
//     value is a scalar register filled by batched s_load_dwordx16 from the run-time values array;
//   * no LDS, no index loads, no loop: the kernel is a straight line of loads, FMAs and coalesced stores, so the
//     hardware streams X in and Y out at HBM rate.
// Shapes that do not fit the register budget fall back to the precompiled LDS-staged kernels (sparse_kernels.hip).
// hiprtc is bound with dlopen so that the library loads on hosts without it (the precompiled kernels then serve).
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "internal.hpp"

namespace xamd {

struct JitKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  int device = -1;
  long long total_threads = 0;
  int vec = 1, elem = 4;
  size_t code_size = 0;
  int refs = 1;
  std::string key, name;
};

namespace {

struct Rtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  bool tried = false, ok = false;
};
Rtc g_rtc;
std::mutex g_jit_lock;
std::unordered_map<std::string, JitKernel*> g_jit_cache;

bool rtc_ready() {
  if (g_rtc.tried) return g_rtc.ok;
  g_rtc.tried = true;
  const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
  for (const char* n : names) { g_rtc.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rtc.lib) break; }
  if (!g_rtc.lib) return false;
#define BIND_(field, sym) g_rtc.field = (decltype(g_rtc.field))dlsym(g_rtc.lib, sym)
  BIND_(create, "hiprtcCreateProgram"); BIND_(compile, "hiprtcCompileProgram"); BIND_(log_size, "hiprtcGetProgramLogSize");
  BIND_(log, "hiprtcGetProgramLog"); BIND_(code_size, "hiprtcGetCodeSize"); BIND_(code, "hiprtcGetCode"); BIND_(destroy, "hiprtcDestroyProgram");
#undef BIND_
  g_rtc.ok = g_rtc.create && g_rtc.compile && g_rtc.log_size && g_rtc.log && g_rtc.code_size && g_rtc.code && g_rtc.destroy;
  return g_rtc.ok;
}

void append(std::string& s, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void append(std::string& s, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt);
  const int n = std::vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (n > 0) s.append(buf, (size_t)(n < (int)sizeof(buf) ? n : (int)sizeof(buf) - 1));
}

// Pick the widest per-lane vector that (a) keeps every row segment aligned, (b) keeps the touched X rows in registers,
// and (c) still leaves enough waves to cover the chip when the column axis is short.
// Results are written once and not read again by the kernel: with beta = 0 they go out as non-temporal stores (measured on the packed CSR /
// FsSpMDM kernels: +3..6 % at P = 65 536, +19 % at P = 4096; with beta = 1 -- C is read first -- plain stores are slightly better; non-temporal
// LOADS changed nothing).  LIBXSMM_HIP_JIT_NT=0 switches both policies back.
static bool nt_stores();
// Round 6: the packed operand is read once, too -- non-temporal LOADS measured +3 % (packed CSR @10 % / @15 %, P = 65 536: 0.678 -> 0.700, 0.670 -> 0.689), +4 % (FsSpMDM
// N = 10^6: 0.696 -> 0.725), +1.5 % at P = 4096 (profiles/r06_jit_nt_loads.jsonl); a bare copy of the same footprint gains 7 % from the policy (tools/copy_floor.hip).
static bool nt_loads() { return nt_stores(); }
static bool nt_stores() { static const bool on = []() { const char* e = getenv("LIBXSMM_HIP_JIT_NT"); return !(e && e[0] == '0'); }(); return on; }
int choose_vec(const SpmmJitSpec& s, int touched, int elem) {
  const int words = elem / 4;
  // f64: ONE double per lane.  Round-3 sweep (profiles/r03_csr_width_sweep.jsonl, LIBXSMM_HIP_JIT_VEC): FsSpMDM f64 N = 2^20 108 us at one double per lane against
  // 117-123 us at two (beta = 1: 181 vs 193 us), packed CSR f64 the same either way (221 vs 221-226 us); f32 keeps four floats per lane (FsSpMDM 53.8 us vs
  // 57.2 / 62.3 at one / two; CSR @10 % 109 vs 114 us).
  const int cands[3] = {elem == 8 ? 1 : 16 / elem, elem == 8 ? 1 : 8 / elem, 1};
  int best = 0;
  // LIBXSMM_HIP_JIT_VEC=<elements per lane>: experiments on the register budget / occupancy trade (taken if the geometry admits it)
  static const int forced = []() { const char* e = getenv("LIBXSMM_HIP_JIT_VEC"); return e ? atoi(e) : 0; }();
  if (forced >= 1 && forced * elem <= 16 && !(s.ncols % forced || s.ld_x % forced || s.ld_y % forced || s.outer_x % forced || s.outer_y % forced) &&
      (long long)touched * forced * words <= 176) return forced;
  for (int ci = 0; ci < 3; ++ci) {
    const int e = cands[ci];
    if (e < 1 || (ci > 0 && e == cands[ci - 1])) continue;
    if (s.ncols % e || s.ld_x % e || s.ld_y % e || s.outer_x % e || s.outer_y % e) continue;
    if ((long long)touched * e * words > 176) continue;
    if (!best) best = e;
    const long long waves = (s.ncols / e) * (long long)s.nouter / 64;
    if (waves >= 2048) return e;           // enough parallelism at this width
  }
  if (!best) return 0;
  // short axis: the narrowest admissible width maximises the number of waves
  for (int ci = 2; ci >= 0; --ci) {
    const int e = cands[ci];
    if (e < 1) continue;
    if (s.ncols % e || s.ld_x % e || s.ld_y % e || s.outer_x % e || s.outer_y % e) continue;
    if ((long long)touched * e * words > 176) continue;
    return e;
  }
  return best;
}

std::string generate_spmm(const SpmmJitSpec& s, int vec, long long* total_threads, const std::string& fname) {
  const bool f64 = (s.dtype == LIBXSMM_DATATYPE_F64);
  const char* T = f64 ? "double" : "float";
  const unsigned int nnz = s.ptr[s.rows];
  std::vector<char> touched((size_t)s.inner, 0);
  for (unsigned int z = 0; z < nnz; ++z) touched[s.idx[z]] = 1;
  const long long tpo = s.ncols / vec;
  *total_threads = tpo;                       // per slab; the launch multiplies by the slab count
  std::string src;
  src.reserve(4096 + (size_t)nnz * 64);
  append(src, "// generated by libxsmm_amd: rows=%d inner=%d nnz=%u ncols=%lld outer=%d vec=%d %s beta0=%d\n", s.rows, s.inner, nnz, s.ncols, s.nouter, vec, T, s.beta0);
  append(src, "typedef %s T;\n", T);
  if (vec > 1) append(src, "typedef T V __attribute__((ext_vector_type(%d)));\n", vec); else src += "typedef T V;\n";
  src += "#define GM __attribute__((address_space(1)))\n#define CM __attribute__((address_space(4)))\n";
  // the slab (outer) count and strides are run-time arguments: a batched launch (libxsmm_hip_gemm_batch_strided on a packed
  // kernel) reuses the same code with the caller's element loop as the slab axis; bslabs = slabs per batch element
  src += "extern \"C\" __global__ __launch_bounds__(256) void " + fname + "(const void* vals_, const void* x_, void* y_, long long nslab, long long bslabs, long long outer_x, long long outer_y, long long batch_x, long long batch_y) {\n";
  append(src, "  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;\n  if (t >= %lldLL * nslab) return;\n", tpo);
  append(src, "  const long long o = t / %lldLL, c = t - o * %lldLL, ob = o / bslabs, oi = o - ob * bslabs;\n", tpo, tpo);
  append(src, "  GM const T* x = (GM const T*)x_ + ob * batch_x + oi * outer_x + c * %d;\n  GM T* y = (GM T*)y_ + ob * batch_y + oi * outer_y + c * %d;\n", vec, vec);
  src += "  CM const T* v = (CM const T*)vals_;\n";
  for (int k = 0; k < s.inner; ++k) if (touched[k]) append(src, nt_loads() ? "  const V x%d = __builtin_nontemporal_load((GM const V*)(x + %lldLL));\n" : "  const V x%d = *(GM const V*)(x + %lldLL);\n", k, (long long)k * s.ld_x);
  // rows that are written: non-empty ones, and empty ones when beta=0 demands zeros [ref: asparse generator :336-345 skips them]
  std::vector<int> live;
  for (int r = 0; r < s.rows; ++r) {
    const bool empty = s.ptr[r] == s.ptr[r + 1];
    if (empty && (s.skip_empty || !s.beta0)) continue;
    live.push_back(r);
  }
  const int ahead = 6;                       // beta=1: old C rows are requested this many rows before they are needed
  if (!s.beta0) for (size_t i = 0; i < live.size() && i < (size_t)ahead; ++i) append(src, "  V c%d = *(GM const V*)(y + %lldLL);\n", live[i], (long long)live[i] * s.ld_y);
  src += "  V acc;\n";
  for (size_t i = 0; i < live.size(); ++i) {
    const int r = live[i];
    if (!s.beta0 && i + ahead < live.size()) append(src, "  V c%d = *(GM const V*)(y + %lldLL);\n", live[i + ahead], (long long)live[i + ahead] * s.ld_y);
    const unsigned int z0 = s.ptr[r], z1 = s.ptr[r + 1];
    if (!s.beta0) append(src, "  acc = c%d;\n", r);
    else if (z0 == z1) src += "  acc = (V)(T)0;\n";
    for (unsigned int z = z0; z < z1; ++z) {
      const unsigned int vz = s.vmap ? s.vmap[z] : z;
      // (beta = 0: the first product is ADDED to +0 as the reference's loop does -- a bare product would keep the sign of a zero product, -0 where the reference has +0;
      //  one instruction either way)
      if (s.beta0 && z == z0) append(src, "  acc = __builtin_elementwise_fma((V)v[%u], x%u, (V)(T)0);\n", vz, s.idx[z]);
      else append(src, "  acc = __builtin_elementwise_fma((V)v[%u], x%u, acc);\n", vz, s.idx[z]);
    }
    append(src, (nt_stores() && s.beta0) ? "  __builtin_nontemporal_store(acc, (GM V*)(y + %lldLL));\n" : "  *(GM V*)(y + %lldLL) = acc;\n", (long long)r * s.ld_y);
  }
  src += "}\n";
  return src;
}

}  // namespace

// compile `src` (entry point `fname`) for the current device and load it; cached by (device, source)
static JitKernel* build_module(const std::string& src, const std::string& fname, long long total, int vec, int elem, std::string* why) {
  auto fail = [&](const char* msg) -> JitKernel* { if (why) *why = msg; return nullptr; };
  if ((total + 255) / 256 >= (1ll << 31)) return fail("grid too large");
  std::lock_guard<std::mutex> guard(g_jit_lock);
  if (!rtc_ready()) return fail("hiprtc is not available");
  // Dry run (LIBXSMM_HIP_DRYRUN=1, no device): the source is compiled for gfx950 and kept (optionally dumped), never loaded or launched --
  // the CPU test-suite compiles what the generators emit and reads the code objects' register / scratch budget.
  const bool dry = rt_dryrun();
  int dev = dry ? -1 : 0;
  std::string arch_name = "gfx950";
  if (!dry) {
    if (hipGetDevice(&dev) != hipSuccess) return fail("no current device");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail("device properties unavailable");
    arch_name = prop.gcnArchName;
  }
  const std::string key = std::to_string(dev) + ":" + src;
  auto it = g_jit_cache.find(key);
  if (it != g_jit_cache.end()) { ++it->second->refs; return it->second; }
  const auto t_start = std::chrono::steady_clock::now();
  hiprtcProgram prog = nullptr;
  if (g_rtc.create(&prog, src.c_str(), "libxsmm_amd_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return fail("hiprtcCreateProgram failed");
  const std::string arch = std::string("--offload-arch=") + arch_name;
  const char* opts[] = {arch.c_str(), "-O3", "-ffp-contract=off"};
  const hiprtcResult rc = g_rtc.compile(prog, 3, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0; (void)g_rtc.log_size(prog, &n);
    std::string log(n, '\0'); if (n) (void)g_rtc.log(prog, &log[0]);
    if (why) *why = "hiprtc compile failed: " + log;
    (void)g_rtc.destroy(&prog);
    return nullptr;
  }
  const auto t_compiled = std::chrono::steady_clock::now();
  size_t csz = 0; (void)g_rtc.code_size(prog, &csz);
  std::vector<char> code(csz);
  (void)g_rtc.code(prog, code.data());
  (void)g_rtc.destroy(&prog);
  if (const char* dump = std::getenv("LIBXSMM_HIP_JIT_DUMP")) {          // <dir>/<kernel>.hip and .co of everything that is generated
    const std::string base = std::string(dump) + "/" + fname;
    if (FILE* f = std::fopen((base + ".hip").c_str(), "wb")) { std::fwrite(src.data(), 1, src.size(), f); std::fclose(f); }
    if (FILE* f = std::fopen((base + ".co").c_str(), "wb")) { std::fwrite(code.data(), 1, code.size(), f); std::fclose(f); }
  }
  JitKernel* k = new JitKernel();
  if (dry) {
    k->device = -1; k->total_threads = total; k->vec = vec; k->elem = elem; k->code_size = csz; k->key = key; k->name = fname;
    g_jit_cache.emplace(key, k);
    return k;
  }
  if (hipModuleLoadData(&k->mod, code.data()) != hipSuccess || hipModuleGetFunction(&k->fn, k->mod, fname.c_str()) != hipSuccess) {
    (void)hipGetLastError();
    if (k->mod) (void)hipModuleUnload(k->mod);
    delete k;
    return fail("hipModuleLoadData failed");
  }
  if (libxsmm_verbosity >= 3 || libxsmm_verbosity < 0) {
    const auto t_loaded = std::chrono::steady_clock::now();
    std::fprintf(stderr, "LIBXSMM-AMD: hiprtc %.0f ms, module load %.0f ms, %zu bytes of source\n", std::chrono::duration<double, std::milli>(t_compiled - t_start).count(),
                 std::chrono::duration<double, std::milli>(t_loaded - t_compiled).count(), src.size());
  }
  k->device = dev; k->total_threads = total; k->vec = vec; k->elem = elem; k->code_size = csz; k->key = key; k->name = fname;
  g_jit_cache.emplace(key, k);
  return k;
}

JitKernel* jit_spmm_create(const SpmmJitSpec& s, std::string* why) {
  auto fail = [&](const char* msg) -> JitKernel* { if (why) *why = msg; return nullptr; };
  const unsigned int nnz = s.ptr[s.rows];
  if (nnz == 0 || nnz > 16384u || s.rows > 4096) return fail("pattern outside the JIT envelope");
  const int elem = (s.dtype == LIBXSMM_DATATYPE_F64) ? 8 : 4;
  std::vector<char> seen((size_t)s.inner, 0); int touched = 0;
  for (unsigned int z = 0; z < nnz; ++z) if (!seen[s.idx[z]]) { seen[s.idx[z]] = 1; ++touched; }
  const int vec = choose_vec(s, touched, elem);
  if (vec == 0) return fail("touched X rows exceed the register budget");
  long long total = 0;
  // the symbol carries the specialisation so that profiles tell the kernels apart
  const std::string fname = std::string("spmm_jit_") + (elem == 8 ? "f64" : "f32") + "_v" + std::to_string(vec) + "_r" + std::to_string(s.rows) + "_k" + std::to_string(s.inner) +
                            "_z" + std::to_string(nnz) + "_b" + std::to_string(s.beta0 ? 0 : 1);
  const std::string src = generate_spmm(s, vec, &total, fname);
  return build_module(src, fname, total, vec, elem, why);
}

// Dense packed GEMM  C[n][m][p] (+)= sum_k A[k][m][p] * B[n][k][p]  [ref: src/generator_packed_gemm_avx_avx512.c; gold
// samples/xgemm_packed/gemm_packed_kernel.c:35-72]: a lane owns `vec` packed positions, fetches its K*M + N*K operand
// vectors once (all loads in flight) and produces the M*N results from registers.
JitKernel* jit_pgemm_create(const PgemmArgs& g, std::string* why) {
  auto fail = [&](const char* msg) -> JitKernel* { if (why) *why = msg; return nullptr; };
  const int elem = (g.dtype == LIBXSMM_DATATYPE_F64) ? 8 : 4, words = elem / 4;
  const long long operands = (long long)g.K * g.M + (long long)g.N * g.K;
  if (operands <= 0 || (long long)g.M * g.N > 4096) return fail("shape outside the JIT envelope");
  int vec = 0;
  for (int e = 16 / elem; e >= 1; e >>= 1) {
    if (g.P % e) continue;
    if (operands * e * words > 176) continue;
    vec = e;
    if ((g.P / e) / 64 >= 2048) break;          // enough waves at this width; otherwise keep narrowing for parallelism
  }
  if (vec == 0) return fail("operands exceed the register budget");
  const char* T = elem == 8 ? "double" : "float";
  const long long total = g.P / vec;
  const std::string fname = std::string("pgemm_jit_") + (elem == 8 ? "f64" : "f32") + "_v" + std::to_string(vec) + "_m" + std::to_string(g.M) + "_n" + std::to_string(g.N) +
                            "_k" + std::to_string(g.K) + "_b" + std::to_string(g.beta0 ? 0 : 1);
  std::string src;
  append(src, "// generated by libxsmm_amd: packed GEMM m=%d n=%d k=%d lda=%d ldb=%d ldc=%d P=%lld vec=%d %s beta0=%d\n", g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.P, vec, T, g.beta0);
  append(src, "typedef %s T;\n", T);
  if (vec > 1) append(src, "typedef T V __attribute__((ext_vector_type(%d)));\n", vec); else src += "typedef T V;\n";
  src += "#define GM __attribute__((address_space(1)))\n";
  src += "extern \"C\" __global__ __launch_bounds__(256) void " + fname + "(const void* a_, const void* b_, void* c_) {\n";
  append(src, "  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;\n  if (t >= %lldLL) return;\n", total);
  append(src, "  GM const T* a = (GM const T*)a_ + t * %d;\n  GM const T* b = (GM const T*)b_ + t * %d;\n  GM T* c = (GM T*)c_ + t * %d;\n", vec, vec, vec);
  for (int k = 0; k < g.K; ++k) for (int m = 0; m < g.M; ++m) append(src, "  const V a%d_%d = *(GM const V*)(a + %lldLL);\n", k, m, ((long long)k * g.lda + m) * g.P);
  for (int n = 0; n < g.N; ++n) for (int k = 0; k < g.K; ++k) append(src, "  const V b%d_%d = *(GM const V*)(b + %lldLL);\n", n, k, ((long long)n * g.ldb + k) * g.P);
  const int ahead = 6, outs = g.M * g.N;
  auto coff = [&](int o) { const int n = o / g.M, m = o % g.M; return ((long long)n * g.ldc + m) * g.P; };
  if (!g.beta0) for (int o = 0; o < outs && o < ahead; ++o) append(src, "  V c%d = *(GM const V*)(c + %lldLL);\n", o, coff(o));
  src += "  V acc;\n";
  for (int o = 0; o < outs; ++o) {
    const int n = o / g.M, m = o % g.M;
    if (!g.beta0 && o + ahead < outs) append(src, "  V c%d = *(GM const V*)(c + %lldLL);\n", o + ahead, coff(o + ahead));
    if (!g.beta0) append(src, "  acc = c%d;\n", o);
    for (int k = 0; k < g.K; ++k) {
      if (g.beta0 && k == 0) append(src, "  acc = __builtin_elementwise_fma(a%d_%d, b%d_%d, (V)(T)0);\n", k, m, n, k);      // (added to +0, as in generate_spmm)
      else append(src, "  acc = __builtin_elementwise_fma(a%d_%d, b%d_%d, acc);\n", k, m, n, k);
    }
    append(src, (nt_stores() && g.beta0) ? "  __builtin_nontemporal_store(acc, (GM V*)(c + %lldLL));\n" : "  *(GM V*)(c + %lldLL) = acc;\n", coff(o));
  }
  src += "}\n";
  return build_module(src, fname, total, vec, elem, why);
}

JitKernel* jit_compile(const std::string& src, const std::string& fname, long long total_threads, int align_bytes, std::string* why) {
  return build_module(src, fname, total_threads, align_bytes, 1, why);     // vec*elem = required pointer alignment
}
int jit_launch(JitKernel* k, void** args, void* stream) {
  const unsigned int grid = (unsigned int)((k->total_threads + 255) / 256);
  return (int)hipModuleLaunchKernel(k->fn, grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr);
}
bool jit_on_current_device(const JitKernel* k) {
  int dev = -1;
  return k && hipGetDevice(&dev) == hipSuccess && dev == k->device;
}

bool jit_pgemm_usable(const JitKernel* k, const void* a, const void* b, const void* c) {
  return jit_spmm_usable(k, a, b) && jit_spmm_usable(k, b, c);
}
bool jit_spmm_usable(const JitKernel* k, const void* x, const void* y) {
  if (!k) return false;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != k->device) return false;
  const size_t mask = (size_t)k->vec * k->elem - 1;
  return (((size_t)x | (size_t)y) & mask) == 0;
}

int jit_spmm_launch(JitKernel* k, const void* vals, const void* x, void* y, void* stream) {     // packed GEMM kernels: three pointers
  void* args[3] = {(void*)&vals, (void*)&x, (void*)&y};
  const unsigned int grid = (unsigned int)((k->total_threads + 255) / 256);
  return (int)hipModuleLaunchKernel(k->fn, grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr);
}
// fixed-pattern kernels: `batch` elements of `bslabs` slabs each; strides in elements.  total_threads holds threads per slab.
int jit_spmm_launch_slabs(JitKernel* k, const void* vals, const void* x, void* y, long long batch, long long bslabs,
                          long long outer_x, long long outer_y, long long batch_x, long long batch_y, void* stream) {
  long long nslab = batch * bslabs;
  void* args[9] = {(void*)&vals, (void*)&x, (void*)&y, (void*)&nslab, (void*)&bslabs, (void*)&outer_x, (void*)&outer_y, (void*)&batch_x, (void*)&batch_y};
  const long long blocks = (k->total_threads * nslab + 255) / 256;
  if (blocks >= (1ll << 31)) return (int)hipErrorInvalidValue;
  return (int)hipModuleLaunchKernel(k->fn, (unsigned int)blocks, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr);
}

void jit_release(JitKernel* k) {
  if (!k) return;
  std::lock_guard<std::mutex> guard(g_jit_lock);
  if (--k->refs > 0) return;
  g_jit_cache.erase(k->key);
  if (k->mod) (void)hipModuleUnload(k->mod);
  delete k;
}

const char* jit_name(const JitKernel* k) { return k ? k->name.c_str() : ""; }
size_t jit_code_size(const JitKernel* k) { return k ? k->code_size : 0; }

}  // namespace xamd
