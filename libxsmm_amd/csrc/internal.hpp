// internal.hpp -- shared between the host runtime (dispatch.cpp, frontend.cpp) and the
// device translation units (*.hip).  Nothing in here is visible to users of include/libxsmm.h.
//
// Design in one paragraph: a dispatch call normalises its arguments into one of the two
// opaque descriptors below, looks the descriptor bytes up in the registry and returns a
// *trampoline*: one of N ahead-of-time instantiated C functions `tramp<I>` whose only job is
// to call `invoke(I, param)`.  That is how a plain `void(*)(const libxsmm_gemm_param*)` can
// carry per-kernel state without JIT-emitting x86 thunks (the reference's closure trick,
// src/generator_x86_reference.c:25-98).  `invoke` decodes the param struct on the host
// (slots the reference's kernels read inside machine code), builds a by-value argument
// block and launches the matching hand-written gfx950 kernel on the calling thread's stream.
#pragma once

#include "../../include/libxsmm.h"
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>

// ---- opaque descriptors (users only ever hold pointers into a libxsmm_descriptor_blob) ----
// Own layout; only the *behaviour* of the reference's init functions is reproduced
// [ref: src/libxsmm_generator.c:36-321].  Both fit LIBXSMM_DESCRIPTOR_MAXSIZE and are
// fully initialised (no padding garbage) because their bytes are the registry key.
struct libxsmm_gemm_descriptor {
  uint32_t m, n, k, lda, ldb, ldc;
  uint32_t flags;
  uint8_t prefetch, a_type, b_type, c_type, comp_type, d_type, br_unroll, store_mask;  // store_mask: bit0 ap, bit1 bp, bit2 cp
  int64_t br_stride_a, br_stride_b;        // bytes, stride mode only
  uint32_t ldd;
  uint16_t bin_type, bin_flags;            // fused binary post-op on C (d operand)
  uint16_t ap_type, bp_type, cp_type;      // fused unary ops
  uint16_t ap_flags, bp_flags, cp_flags;
  uint32_t ldap, ldbp, ldcp;
  uint32_t reserved;
} __attribute__((packed));   // packed + byte-aligned: a caller's blob may sit at any address (the reference's descriptors are packed too)
static_assert(sizeof(libxsmm_gemm_descriptor) <= LIBXSMM_DESCRIPTOR_MAXSIZE, "gemm descriptor too large");

struct libxsmm_meltw_descriptor {
  uint32_t m, n, ldi, ldo, ldi2, ldi3;
  uint8_t in0_type, in1_type, in2_type, comp_type, out_type, operation;
  uint16_t flags, param;
  uint32_t reserved;
} __attribute__((packed));
static_assert(sizeof(libxsmm_meltw_descriptor) <= LIBXSMM_DESCRIPTOR_MAXSIZE, "meltw descriptor too large");

namespace xamd {

// ---- kernel kinds ---------------------------------------------------------------------------
enum Kind : int {
  K_GEMM = 0,            // dense (BR)GEMM, optional fused epilogue
  K_TILECFG,             // AMX tile-config handles: no-ops on this backend
  K_MELTW,               // unary / binary / ternary TPP
  K_SPMM_ASPARSE,        // packed CSR, A sparse (also the FsSpMDM inner kernel)
  K_SPMM_BSPARSE,        // packed CSR/CSC, B sparse
  K_SPMM_CSPARSE,        // packed CSC with a sparse C: the packed axis is reduced
  K_BCSC,                // block-sparse B, pattern at run time
  K_PGEMM,               // dense packed GEMM (A, B and C in SOA layout)
  K_MEQN                 // matrix equation (tree of TPPs)
};

// ---- device argument blocks (passed by value as kernel arguments) ---------------------------
struct GemmArgs {
  const char* a; const char* b; char* c;            // `primary` slots of batch element 0
  const char* d; unsigned char* relu_mask;          // fused bias operand, ReLU bitmask output
  const long long* offs_a; const long long* offs_b; // OFFSET mode (bytes), shared by the batch
  const void* const* list_a; const void* const* list_b; void* const* list_c;  // pointer-list batch
  long long bs_a, bs_b, bs_c, bs_d, bs_mask;        // batch byte strides
  // 2-D strided batch (blocked GEMM out of BRGEMM tiles): element e = (i, j), i = e % batch_inner fastest;
  // A steps with i (bs_a), B with j (bs_b), C / the bitmask with both (bs_c, bs_mask along i; bs_c2, bs_mask2 along j), the bias with i.
  unsigned int batch_inner;                         // 0: 1-D batch
  int stream_hint;                                  // libxsmm_hip_set_streaming_hint of the calling thread: 0 auto, 1 operands re-read / cache-resident, 2 read once from HBM
  unsigned int map2d_shift;                         // set by launch_gemm: 0 = linear element order, s = super-tiles of 2^s x 2^s elements per XCD
  long long bs_c2, bs_mask2;
  long long br_stride_a, br_stride_b;               // bytes
  unsigned long long br_count;
  unsigned int nbatch;
  int m, n, k, lda, ldb, ldc;
  unsigned int flags;                               // libxsmm_gemm_flags
  int br_mode;                                      // 0 none, 1 address, 2 offset, 3 stride
  int a_type, b_type, c_type;
  int colbias, act;                                 // act: 0 none, 1 relu, 2 relu+bitmask, 3 sigmoid
  int vnni_c;
  int tiles_m, tiles_n;                             // set by launch_gemm for the tile size of the chosen kernel
  int comp_f16;                                     // F16 GEMM with comp_type F16: the running sum is rounded to f16 after every product (generic kernel)
  float scf;                                        // 8-bit GEMM with f32 output: scale read from c.tertiary on the host
  const char* a_scf; long long bs_scf;              // MXFP4 A: E8M0 scales (a.tertiary; a pointer list in ADDRESS mode) and their batch stride
  const char* b_scf; long long bs_bscf;             // MX x MX: the scales of B (b.tertiary)
  int lists_aligned16;                              // pointer-list batch whose every pointer is known to be 16-byte aligned (lists built by the library: the coalescing queue)
  int tune;                                         // experiment switches set by launch_gemm from the environment (bit 0: 6-bit MX operands gathered per lane instead of staged through LDS)
};

struct MeltwArgs {
  const char* in0; const char* in1; const char* in2; char* out;
  const void* aux_in; void* aux_out;                // secondary slots (masks, indices, offsets)
  long long bs_in0, bs_in1, bs_in2, bs_out, bs_aux; // batch byte strides
  unsigned long long scalar_u64;                    // op.primary payloads read on the host
  unsigned long long scalar_u64b;                   // a second one (DECOMP_FP32_TO_BF16X3: the byte offset of the third piece)
  float scalar_f32;
  unsigned int nbatch;
  int m, n, ldi, ldi1, ldi2, ldo;
  int in0_type, in1_type, in2_type, out_type, comp_type;
  void* ws; size_t ws_bytes;                          // device workspace for two-pass kernels (may be NULL)
  unsigned int flags;
  int type, operation;
  int nt;                                             // round 6: the launch's operands cannot be cache resident (footprint, streaming hint): non-temporal loads and stores in the streaming kernels
};

// sparse operator S (rows x inner) applied to a packed panel:
//   Y[r][q] (+)= sum_z val[z] * X[idx[z]][q],   q = 0..ncols-1 contiguous, for `nouter` slabs
struct SpmmArgs {
  const unsigned int* ptr; const unsigned int* idx;   // device copies of the pattern
  const void* vals;                                   // run-time values (device-accessible)
  const unsigned int* vmap;                           // optional: value position of pattern entry z
  const char* x; char* y;
  long long ld_x, ld_y;                               // row strides in elements
  long long outer_x, outer_y;                         // slab strides in elements
  long long ncols; int rows, inner, nouter; unsigned int nnz;
  int dtype, beta0, skip_empty, vals_are_f64;
};

// C[n][m][p] (+)= sum_k A[k][m][p] * B[n][k][p]; leading dimensions in packed elements
struct PgemmArgs {
  const char* a; const char* b; char* c;
  int M, N, K, lda, ldb, ldc, dtype, beta0;
  long long P;
};

// C_val[z] (+)= sum_k sum_p A[(k*lda + rows[z])*P + p] * B[(k*ldb + cols[z])*P + p]
struct CsparseArgs {
  const char* a; const char* b; char* c;
  const unsigned int* rows; const unsigned int* cols;
  unsigned int nnz; int K, lda, ldb, beta0; long long P;
};

struct BcscArgs {
  const char* a; const char* bvals; char* c;
  const unsigned int* colptr; const unsigned int* rowidx;
  int M, N, K, m_blocks, bk, bn, nblk_n;
  int a_type, b_type, c_type, vnni_a, beta0;      // 8-bit integers: a_type / b_type in {I8, U8}, exactly one unsigned, c_type I32, A VNNI-4
  void* table;                        // workspace for the inverted pattern: nblk_n * (K / bk) words (may be NULL)
  int nt_a;                           // stream the A operand with non-temporal loads (set by launch_bcsc: launch larger than the Infinity Cache, or the caller's hint)
  int stream_hint;                    // libxsmm_hip_set_streaming_hint of the calling thread
  int table_ready;                    // the table already holds this call's inverted pattern (host-resident pattern, cached per kernel)
  unsigned long long kmask0;          // host-resident pattern: bit kb set when some block of the first 64 columns lies in k-block kb (valid when nnzb > 0 and K / bk <= 64) --
                                      // a kernel whose waves all work on those columns knows its first A requests without a look at the table
  int nnzb;                           // number of blocks of B when the host knows it (host-resident or bound pattern), else 0: a B of a few KiB is kept in LDS by the streaming kernel (round 6)
};

// ---- run-time specialised sparse kernels (jit.cpp) ------------------------------------------------------
struct SpmmJitSpec {              // everything the generated kernel bakes in; pointers are HOST arrays
  int dtype, rows, inner, nouter, beta0, skip_empty;
  const unsigned int* ptr; const unsigned int* idx; const unsigned int* vmap;
  long long ld_x, ld_y, outer_x, outer_y, ncols;   // elements
};
struct JitKernel;
JitKernel* jit_spmm_create(const SpmmJitSpec& spec, std::string* why);
JitKernel* jit_pgemm_create(const PgemmArgs& geometry, std::string* why);   // pointers of `geometry` are ignored
bool jit_spmm_usable(const JitKernel* k, const void* x, const void* y);
bool jit_pgemm_usable(const JitKernel* k, const void* a, const void* b, const void* c);
int jit_spmm_launch(JitKernel* k, const void* vals, const void* x, void* y, void* stream);
int jit_spmm_launch_slabs(JitKernel* k, const void* vals, const void* x, void* y, long long batch, long long bslabs,
                          long long outer_x, long long outer_y, long long batch_x, long long batch_y, void* stream);
void jit_release(JitKernel* k);
JitKernel* jit_compile(const std::string& src, const std::string& fname, long long total_threads, int align_bytes, std::string* why);
int jit_launch(JitKernel* k, void** args, void* stream);
bool jit_on_current_device(const JitKernel* k);
int rt_jit_mode();
bool rt_dryrun();            // LIBXSMM_HIP_DRYRUN=1 and no device: dispatch / code generation work, nothing can be launched
const char* jit_name(const JitKernel* k);
size_t jit_code_size(const JitKernel* k);

// ---- host-side kernel context ------------------------------------------------------------------
struct KernelCtx {
  int slot = -1;
  Kind kind = K_GEMM;
  bool registered = false;          // owned by the registry (dispatch_*) vs caller-owned (create_*)
  libxsmm_gemm_descriptor g{};
  libxsmm_meltw_descriptor e{};
  unsigned int nflops = 0;
  // sparse creators
  int packed_width = 0, bk = 0, bn = 0;
  int sp_rows = 0, sp_inner = 0; unsigned int sp_nnz = 0;
  unsigned int* d_ptr = nullptr; unsigned int* d_idx = nullptr; void* d_vals = nullptr;  // device pattern (+ baked values)
  unsigned int* d_vmap = nullptr;   // value position per pattern entry (B-sparse CSR regrouped by column)
  int sp_ncols = 0, sp_skip_empty = 0;
  JitKernel* jit = nullptr;         // pattern-specialised kernel (nullptr: precompiled kernels serve)
  std::vector<unsigned int> h_ptr, h_idx, h_vmap;   // host pattern kept while specialisation is deferred to the first batched launch
  struct EqnPlan* eqn = nullptr;    // K_MEQN: the evaluation plan (meqn.cpp)
  // K_BCSC: patterns that arrived in HOST memory, inverted on the host once and kept on the device ([colptr | rowidx | table] per entry)
  // Entries are immutable once published and live until the kernel is released: the hit path reads `bcsc_last` without a lock.
  struct BcscCached { std::vector<unsigned int> pattern; unsigned int* d_block = nullptr; unsigned long long kmask0 = 0ull; };
  std::vector<BcscCached*> bcsc_cache;            // guarded by the cache lock (miss path only); at most 4 current entries, evicted ones go to bcsc_old
  std::vector<BcscCached*> bcsc_old;
  std::atomic<const BcscCached*> bcsc_last{nullptr};
  // libxsmm_hip_bcsc_bind_pattern: a DEVICE-resident pattern the caller promises not to change: its inverted table is built once
  struct BcscBound { const void* colptr = nullptr; const void* rowidx = nullptr; unsigned long long nblk_n = 0; unsigned int* d_table = nullptr; int nkb = 0;
                     int nnzb = 0; unsigned long long kmask0 = 0ull; };      // read back once at bind time (0: not read -- the bind was captured into a graph)
  BcscBound bcsc_bound;
  int device = 0;
  const char* kname_single = "";                 // static strings or strings owned by a never-shrinking table
  const char* kname_batched = "";
};

// ---- matrix equations (meqn.cpp) and the runtime services they use (runtime.cpp) ------------------------
struct EqnPlan;
void run_meqn(EqnPlan* plan, const void* param);
void free_meqn_plan(EqnPlan* plan);
void free_meqn_equations();                      // libxsmm_finalize: drop every equation object
const char* meqn_plan_name(const EqnPlan* plan);
const void* rt_new_meqn_handle(EqnPlan* plan);   // caller-independent handle owned by the equation registry
void rt_finish_launch(int err, const char* kernel_name);
void* rt_workspace(size_t nbytes);
void rt_workspace_reserve(size_t nbytes);   // nested workspace() requests are placed behind this many bytes (0: off)
bool rt_ready();
const void* rt_small_host_input(const void* p, size_t nbytes);   // tiny operands (scalars) may live in host memory: staged if they do
void* rt_small_host_output(void* p, size_t nbytes);              // ... and so may a 1 x 1 result: staged and copied back when the call is synchronous
void rt_scratch_reset();
void rt_nest(int delta);     // +1 / -1 around a kernel handle invoked from inside another invocation (equation GEMM nodes)
void rt_note(const char* what, int a, int b, int c);      // verbosity >= 1: why a request was refused
void* rt_stream();

// ---- per-thread execution state -----------------------------------------------------------------
struct ThreadState {
  void* stream = nullptr;     // hipStream_t
  int async = -1;             // -1: not initialised from the environment yet
  int stream_hint = 0;        // libxsmm_hip_set_streaming_hint
  int device = -1;
  int last_error = 0;
  std::string last_error_msg;
  unsigned long long launches = 0;
  // libxsmm_hip_pipeline_begin / _end: launches between the two are declared independent and rotate over `pipe_lanes` internal streams
  int pipe_lanes = 0, pipe_cur = 0, pipe_device = -1;
  void* pipe_user = nullptr;             // the caller's stream while a pipeline section is open
  void* pipe_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void* pipe_event[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [8]: the fork point
};
ThreadState& tls();
void set_error(int code, const char* fmt, ...);

// ---- launch entry points implemented in the .hip translation units ------------------------------
// Each returns a HIP error code (0 == success) and the name of the kernel it picked.
int launch_gemm(const GemmArgs& args, void* stream, const char** kernel_name);
int launch_gemm_f64(const GemmArgs& args, void* stream, const char** kernel_name);     // gemm_f64_kernels.hip: v_mfma_f64_16x16x4_f64
const char* gemm_f64_kernel_name(const libxsmm_gemm_descriptor& d);
int launch_gemm_f32_wg64_sharedb(const GemmArgs& args, bool nt, void* stream, const char** kernel_name, int* taken);   // gemm_sharedb_kernels.hip: 64^3 f32, B shared by the batch
int launch_gemm_p16w(const GemmArgs& args, bool nt, void* stream, const char** kernel_name, int* taken);
int launch_gemm_wgp16(const GemmArgs& args, void* stream, const char** kernel_name, int* taken);
int launch_gemm_16bit_w64(const GemmArgs& args, bool nt, void* stream, const char** kernel_name, int* taken);   // gemm_w64_kernels.hip: bf16 / f16 64^3, one problem per wave, whole-line requests
int launch_gemm_wgp_f32(const GemmArgs& args, void* stream, const char** kernel_name, int* taken);      // gemm_wgp_f32_kernels.hip
int launch_gemm_wgp8(const GemmArgs& args, int kind, bool ua, bool ub, void* stream, const char** kernel_name, int* taken);   // 8-bit x 8-bit (integers / BF8 / HF8), packed blocks
int launch_gemm_wgp16_w8(const GemmArgs& args, int kind, void* stream, const char** kernel_name, int* taken);   // the same form for 8-bit weights x bf16 (kind: gemm_w8_bf16_kernel's KIND)
const char* gemm_kernel_name(const libxsmm_gemm_descriptor& d, bool batched);
bool gemm_supported(const libxsmm_gemm_descriptor& d);
int launch_meltw(const MeltwArgs& args, void* stream, const char** kernel_name);
int launch_mx_out_quant(const float* src, void* dst, void* scf, int m, int n, int ldc, int fp4, unsigned int nbatch, long long bs_dst, long long bs_scf, void* stream);
int launch_bitmask_expand(const void* bitmap, const void* vals, void* dense, unsigned int* rows_scratch, int rows, int row_bytes, int elem_size, void* stream);
int launch_gemm_bitmask16(const GemmArgs& args, const void* bitmap, unsigned int* scratch, size_t scratch_bytes, void* stream, const char** kernel_name, int* taken);
size_t gemm_bitmask_reg_workspace(const GemmArgs& args);      // gemm_bitmask_kernels.hip (round 4): bytes of workspace of the register-expanding bitmask GEMM, 0 = shape not taken
int launch_gemm_bitmask_reg(const GemmArgs& args, const void* bitmap, void* ws, size_t ws_bytes, void* stream, const char** kernel_name, int* taken);
int launch_stochastic_bf8(const MeltwArgs& args, void* stream);     // second pass of a TPP with *_STOCHASTIC_ROUND: f32 results -> BF8
bool meltw_supported(const libxsmm_meltw_descriptor& d);
int launch_mfma_probe(int bf16, const void* operands, int iterations, void* stream, double* flop);
int launch_brchain_f32(const GemmArgs& args, float* partial, size_t partial_capacity_tiles, int* nslices, void* stream, const char** kernel_name);
int launch_brsplit_reduce(const GemmArgs& args, const float* partial, int nsplit, void* stream);
int launch_spmm(const SpmmArgs& args, void* stream, const char** kernel_name);
int launch_bcsc(const BcscArgs& args, void* stream, const char** kernel_name);
// Automatic streaming decision (libxsmm_hip_set_streaming_hint(0)): a launch whose own operands exceed the Infinity Cache streams -- and so does a launch whose operands
// TOGETHER WITH those of the calling thread's recent launches on other operands do (a caller that walks over more input sets than the cache holds re-reads nothing
// from it either).  `key` identifies the launch's operand set (its first operand's address), `bytes` is what the launch moves.  32 sets are remembered, each forgotten after
// 96 decisions without being seen again, so a caller that settles on one resident set gets cacheable loads back.  (runtime.cpp, thread-local.)
// `out`: what the launch writes.  A launch whose first operand IS what one of the last eight decisions' launches wrote is a hand-over inside a chain (GEMM -> TPP on its C): its
// operand may well still be cached, it stays cacheable whatever the sum says.
bool rt_recent_operands_exceed_cache(const void* key, unsigned long long bytes, const void* out = nullptr);
int rt_window_verdict();
int launch_bcsc_invert(const unsigned int* colptr, const unsigned int* rowidx, unsigned int* table, int nblk_n, int nkb, void* stream);
int launch_csparse(const CsparseArgs& args, void* stream, const char** kernel_name);
int launch_pgemm(const PgemmArgs& args, void* stream, const char** kernel_name);

// ---- runtime services (dispatch.cpp) ---------------------------------------------------------------
KernelCtx* ctx_from_handle(const void* fn);
const void* handle_for_slot(int slot);
void invoke(int slot, const void* param);
bool runtime_ready();              // library initialised and a device is present
int typesize(int t);

}  // namespace xamd
