// runtime.cpp -- host runtime of libxsmm_amd: library state, descriptor construction, the
// descriptor -> handle registry, trampolines, and the per-call argument decoding that turns a
// libxsmm_*_param into a kernel launch.  Replaces the reference's L2 layer
// [ref: src/libxsmm_main.c:1234 (init), :2730 internal_find_code, :2132 libxsmm_build,
//  :3323-3511 dispatchers; src/libxsmm_generator.c:36-321 descriptor init].
// There is no CPU fallback anywhere in this file: without a HIP device every dispatch returns NULL
// and says why (stderr, once) -- the library fails loudly instead of silently computing on the host.
#include <hip/hip_runtime_api.h>
#include "internal.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <utility>

using namespace xamd;

static int jit_mode();   // LIBXSMM_HIP_JIT / libxsmm_hip_set_jit, defined with the sparse creators

// ---- exported global state [ref: include/libxsmm_generator.h:213-222] ---------------------------
extern "C" {
__attribute__((visibility("default"))) unsigned int libxsmm_ninit = 0;
__attribute__((visibility("default"))) int libxsmm_verbosity = 0;
__attribute__((visibility("default"))) int libxsmm_target_archid = LIBXSMM_X86_GENERIC;
__attribute__((visibility("default"))) int libxsmm_stdio_handle = 0;   // [ref: include/libxsmm_generator.h:214] 0: no user lock on I/O
__attribute__((visibility("default"))) int libxsmm_se = 0;             // security-enhanced environment: never the case here
}

namespace {

// ---- handle table ------------------------------------------------------------------------------------
// A handle is a plain C function pointer that must carry per-kernel state.  The reference gets that closure by
// JIT-emitting the whole kernel; here a handle is a 32-byte x86-64 THUNK in an executable pool that loads its slot
// number and tail-calls xamd_invoke(slot, param) -- the same trick the reference plays for its C reference kernels
// [ref: src/generator_x86_reference.c:52-96].  Slot <-> handle is pure address arithmetic in both directions.
// Capacity follows the reference's registry [ref: src/libxsmm_main.h:18-22]: 131072 REGISTERED kernels (dispatch_*),
// plus as many caller-owned ones (create_*).  Thunk pages are written once, 128 thunks at a time, then flipped to
// read+execute (W^X).  If the process may not map executable memory (hardened kernels; the reference cannot JIT there
// either) a small table of ahead-of-time instantiated trampolines serves instead.
constexpr int kRegistryCapacity = 131072;
constexpr int kSlots = 2 * kRegistryCapacity;
constexpr int kStaticSlots = 256;
constexpr size_t kThunkBytes = 32, kThunkPage = 4096, kThunksPerPage = kThunkPage / kThunkBytes;
std::mutex g_lock;
KernelCtx* g_slots[kSlots];
std::vector<int> g_free_slots;
int g_next_slot = 0;
int g_slot_limit = kSlots;            // LIBXSMM_HIP_MAX_HANDLES lowers it (tests of the exhaustion path)
int g_registered_limit = kRegistryCapacity;
int g_device_count = -1;
bool g_warned_nodevice = false;
bool g_dryrun = false;                // LIBXSMM_HIP_DRYRUN=1: dispatch works without a device (registry tests); calling a kernel is an error
std::atomic<unsigned int> g_generation{1};   // bumped by libxsmm_finalize: invalidates every thread's dispatch cache

unsigned char* g_thunk_pool = nullptr;       // kSlots * kThunkBytes, nullptr: static trampolines only
std::vector<bool> g_thunk_page_ready;

template <int I> void tramp(const void* param) { xamd::invoke(I, param); }
using tramp_fn = void (*)(const void*);
template <int... Is> constexpr std::array<tramp_fn, sizeof...(Is)> make_tramps(std::integer_sequence<int, Is...>) {
  return {{&tramp<Is>...}};
}
const std::array<tramp_fn, kStaticSlots> g_tramps = make_tramps(std::make_integer_sequence<int, kStaticSlots>{});

extern "C" void xamd_invoke_thunk(int slot, const void* param) { xamd::invoke(slot, param); }

void thunk_pool_init_locked() {
  if (g_thunk_pool) return;
#if defined(__x86_64__)
  const char* off = std::getenv("LIBXSMM_HIP_THUNKS");
  if (off && off[0] == '0') return;
  void* p = mmap(nullptr, (size_t)kSlots * kThunkBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return;
  // probe once that a page may become executable at all
  if (mprotect(p, kThunkPage, PROT_READ | PROT_EXEC) != 0) { munmap(p, (size_t)kSlots * kThunkBytes); return; }
  (void)mprotect(p, kThunkPage, PROT_READ | PROT_WRITE);
  g_thunk_pool = (unsigned char*)p;
  g_thunk_page_ready.assign(((size_t)kSlots + kThunksPerPage - 1) / kThunksPerPage, false);
#endif
}
// make the thunk of `slot` callable; false if its page could not be made executable
bool thunk_ready_locked(int slot) {
  if (!g_thunk_pool) return slot < kStaticSlots;
  const size_t page = (size_t)slot / kThunksPerPage;
  if (g_thunk_page_ready[page]) return true;
  unsigned char* base = g_thunk_pool + page * kThunkPage;
  const unsigned long long target = (unsigned long long)(size_t)&xamd_invoke_thunk;
  for (size_t t = 0; t < kThunksPerPage; ++t) {
    unsigned char* c = base + t * kThunkBytes;
    const unsigned int id = (unsigned int)(page * kThunksPerPage + t);
    size_t o = 0;
    c[o++] = 0x48; c[o++] = 0x89; c[o++] = 0xfe;                                   // mov rsi, rdi   (param -> 2nd argument)
    c[o++] = 0xbf; std::memcpy(c + o, &id, 4); o += 4;                             // mov edi, slot
    c[o++] = 0x48; c[o++] = 0xb8; std::memcpy(c + o, &target, 8); o += 8;          // movabs rax, &xamd_invoke_thunk
    c[o++] = 0xff; c[o++] = 0xe0;                                                  // jmp rax
    while (o < kThunkBytes) c[o++] = 0xcc;                                         // int3 padding
  }
  if (mprotect(base, kThunkPage, PROT_READ | PROT_EXEC) != 0) return false;
  g_thunk_page_ready[page] = true;
  return true;
}

// ---- registry key: kind + the zero-padded descriptor bytes; hashed once per dispatch, never heap-allocated ----------
struct RegKey {
  unsigned long long w[(LIBXSMM_DESCRIPTOR_MAXSIZE + 7) / 8 + 1];   // [0] = kind, [1..] = descriptor bytes
  bool operator==(const RegKey& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
inline unsigned long long hash_key(const RegKey& k) {
  unsigned long long h = 0x9e3779b97f4a7c15ull;
  for (unsigned long long x : k.w) { h ^= x; h *= 0xff51afd7ed558ccdull; h ^= h >> 29; }
  return h ^ (h >> 32);
}
struct RegKeyHash { size_t operator()(const RegKey& k) const { return (size_t)hash_key(k); } };
std::unordered_map<RegKey, KernelCtx*, RegKeyHash> g_registry;
std::vector<KernelCtx*> g_meqn_ctxs;         // equation handles: registry-owned, freed at finalize

const char* const kTypeNames[] = {
#define X_(NAME, SIZE) #NAME,
  LIBXSMM_DATATYPE_TABLE(X_)
#undef X_
  "" };
const unsigned char kTypeSizes[] = {
#define X_(NAME, SIZE) SIZE,
  LIBXSMM_DATATYPE_TABLE(X_)
#undef X_
  0 };

void vlog(int level, const char* fmt, ...) {
  if (libxsmm_verbosity == 0 || (libxsmm_verbosity > 0 && libxsmm_verbosity < level)) return;   // library code is mute by default
  va_list ap; va_start(ap, fmt);
  std::fprintf(stderr, "LIBXSMM-AMD: "); std::vfprintf(stderr, fmt, ap); std::fprintf(stderr, "\n");
  va_end(ap);
}

bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  set_error((int)e, "%s failed: %s", what, hipGetErrorString(e));
  return false;
}

hipStream_t cur_stream() { return (hipStream_t)tls().stream; }

void copy_back_staged();
// > 0 while a kernel handle is invoked from inside another invocation (a MATMUL / BRGEMM node of a matrix equation): the nested call must
// neither rewind the caller's staging scratch (its staged operands and pending copy-backs live there) nor synchronise / copy back early
thread_local int t_nest = 0;
void finish_launch(int err, const char* kname) {
  ThreadState& t = tls();
  ++t.launches;
  if (t.pipe_lanes > 1 && t_nest == 0) {           // an open pipeline section: the next independent launch goes to the next lane
    t.pipe_cur = (t.pipe_cur + 1) % t.pipe_lanes;
    t.stream = t.pipe_stream[t.pipe_cur];
  }
  if (err != 0) { set_error(err, "launch of %s failed: %s", kname ? kname : "?", hipGetErrorString((hipError_t)err)); return; }
  if (!t.async && t_nest == 0) {
    hipError_t e = hipStreamSynchronize(cur_stream());
    if (e != hipSuccess) set_error((int)e, "kernel %s faulted: %s", kname ? kname : "?", hipGetErrorString(e));
    copy_back_staged();
  }
}

// Index arrays (BR offsets / address lists, BCSC pattern) are dereferenced on the device.  Arrays in
// plain host memory are staged through a per-thread device scratch; device-visible ones pass through.
// Device blocks that were outgrown are RETIRED, not freed: a captured hipGraph (or a kernel still in flight on another
// stream) may hold their address.  They are released at libxsmm_finalize.
struct Retired { std::mutex lock; std::vector<void*> blocks; };
Retired& retired() { static Retired* r = new Retired(); return *r; }   // never destroyed: a thread may still retire its blocks while the process exits
void retire_block(void* p) { if (p) { Retired& r = retired(); std::lock_guard<std::mutex> guard(r.lock); r.blocks.push_back(p); } }
static int cur_device() { const int d = tls().device; return d < 0 ? 0 : d; }
struct Scratch { char* base = nullptr; size_t cap = 0, used = 0; int device = 0; ~Scratch() { retire_block(base); } };   // a thread that exits hands its block to finalize
thread_local Scratch t_scratch;
// Inside a pipeline section every LANE stages into a scratch of its own: a lane's uploads and kernels are ordered on that lane's stream, so the lane's scratch is rewound
// at the start of each of its calls like the thread's own outside a section.  (Until round 5 a section appended to the one shared scratch -- call N + 1's upload on
// another lane is not ordered behind call N's kernel -- and a long section grew it, retiring block after block until libxsmm_finalize.)  [advisor, round 4]
thread_local Scratch t_scratch_lane[8];
Scratch& cur_scratch();
struct CopyBack { void* host; const void* dev; size_t width, height, pitch; };   // height rows of `width` bytes, `pitch` bytes apart (height 1: plain)
thread_local std::vector<CopyBack> t_copyback;       // staged outputs of the current synchronous call
// Stages a host-resident operand: `height` rows of `width` bytes that lie `pitch` bytes apart (a panel of a wider matrix: only the bytes the
// kernel touches are copied, in either direction -- the rows' gaps belong to the caller).  The device image keeps the pitch, so kernels see
// the caller's leading dimension.  `copy_in`: upload the current contents; `copy_back`: download after the kernel (finish_launch, synchronous mode).
static void* stage2d(const void* p, size_t width, size_t height, size_t pitch, bool copy_in, bool copy_back) {
  if (p == nullptr || width == 0 || height == 0) return const_cast<void*>(p);
  hipPointerAttribute_t attr;
  const hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e == hipSuccess && attr.type != hipMemoryTypeUnregistered) return const_cast<void*>(p);
  (void)hipGetLastError();   // clear the sticky "invalid value" of an unregistered pointer
  if (height == 1 || pitch == width) { width *= height; height = 1; pitch = width; }
  const size_t nbytes = (height - 1) * pitch + width;
  Scratch& s = cur_scratch();
  const size_t need = (nbytes + 255) & ~(size_t)255;
  if (s.base && s.device != cur_device()) { retire_block(s.base); s.base = nullptr; s.cap = s.used = 0; }   // the thread switched device
  if (s.used + need > s.cap) {
    // Pointers handed out earlier in this call (staged operands already written into the argument block, pending copy-backs)
    // must stay valid, so the old block is neither moved nor freed: it is RETIRED (released at libxsmm_finalize) and the
    // overflowing request starts a fresh, larger block.  Growth is geometric, so a thread retires O(log size) blocks.
    const size_t ncap = std::max<size_t>((s.used + need) * 2, 1 << 20);
    char* nb = nullptr;
    if (!hip_ok(hipMalloc((void**)&nb, ncap), "hipMalloc(scratch)")) return nullptr;
    if (s.base) retire_block(s.base);
    s.base = nb; s.cap = ncap; s.used = 0; s.device = cur_device();
  }
  char* dst = s.base + s.used; s.used += need;
  if (copy_in) {
    const hipError_t ce = height == 1 ? hipMemcpyAsync(dst, p, width, hipMemcpyHostToDevice, cur_stream())
                                      : hipMemcpy2DAsync(dst, pitch, p, pitch, width, height, hipMemcpyHostToDevice, cur_stream());
    if (!hip_ok(ce, "hipMemcpyAsync(host operand)")) return nullptr;
  }
  if (copy_back) t_copyback.push_back(CopyBack{const_cast<void*>(p), dst, width, height, pitch});
  return dst;
}
static void* stage(const void* p, size_t nbytes, bool copy_in, bool copy_back) { return stage2d(p, nbytes, 1, nbytes, copy_in, copy_back); }
const void* device_visible(const void* p, size_t nbytes) { return stage(p, nbytes, true, false); }
// an array KNOWN to live in plain host memory (the coalescing queue's pointer lists): uploaded into the scratch without the pointer query.
// The upload does NOT read the caller's (pageable, soon overwritten) storage asynchronously: the bytes are first copied into a PINNED slot owned by
// this thread, the device copy leaves from there, and a slot is reused only after the event behind its last upload has completed.  A pointer-list flush
// stages three lists (A, B, C), so the ring has eight slots: two flushes in flight before the host waits on the oldest [advisor, round 5].  A slot's event belongs
// to the device it was created on: a thread that moved to another device (libxsmm_hip_set_device) gets a fresh event for the slot, the pinned bytes are portable.
// [advisor, round 4: hipMemcpyAsync straight out of the queue's std::vector is a race once the runtime copies truly asynchronously]
struct PinnedSlot { char* base = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; int device = -1; };
struct PinnedRing { static constexpr int NSLOT = 8; PinnedSlot slot[NSLOT]; int next = 0;
  ~PinnedRing() { for (PinnedSlot& p : slot) { if (p.base) (void)hipHostFree(p.base); if (p.done) (void)hipEventDestroy(p.done); } } };
thread_local PinnedRing t_pinned;
static void* stage_host(const void* p, size_t nbytes) {
  if (!p || nbytes == 0) return nullptr;
  Scratch& s = cur_scratch();
  const size_t need = (nbytes + 255) & ~(size_t)255;
  if (s.base && s.device != cur_device()) { retire_block(s.base); s.base = nullptr; s.cap = s.used = 0; }
  if (s.used + need > s.cap) {
    const size_t ncap = std::max<size_t>((s.used + need) * 2, 1 << 20);
    char* nb = nullptr;
    if (!hip_ok(hipMalloc((void**)&nb, ncap), "hipMalloc(scratch)")) return nullptr;
    if (s.base) retire_block(s.base);
    s.base = nb; s.cap = ncap; s.used = 0; s.device = cur_device();
  }
  char* dst = s.base + s.used; s.used += need;
  PinnedSlot& ps = t_pinned.slot[t_pinned.next]; t_pinned.next = (t_pinned.next + 1) % PinnedRing::NSLOT;
  if (ps.busy) { (void)hipEventSynchronize(ps.done); ps.busy = false; }          // the upload that last used this slot has left it
  if (ps.done && ps.device != cur_device()) { (void)hipEventDestroy(ps.done); ps.done = nullptr; }      // the event of another device cannot be recorded on this one's stream
  if (ps.cap < nbytes) {
    if (ps.base) { (void)hipHostFree(ps.base); ps.base = nullptr; ps.cap = 0; }
    const size_t ncap = std::max<size_t>(nbytes * 2, 64 << 10);
    if (!hip_ok(hipHostMalloc((void**)&ps.base, ncap, hipHostMallocPortable), "hipHostMalloc(pointer lists)")) { ps.base = nullptr; return nullptr; }
    ps.cap = ncap;
  }
  if (!ps.done) { if (!hip_ok(hipEventCreateWithFlags(&ps.done, hipEventDisableTiming), "hipEventCreate(pointer lists)")) return nullptr; ps.device = cur_device(); }
  std::memcpy(ps.base, p, nbytes);
  if (!hip_ok(hipMemcpyAsync(dst, ps.base, nbytes, hipMemcpyHostToDevice, cur_stream()), "hipMemcpyAsync(pointer lists)")) return nullptr;
  if (hip_ok(hipEventRecord(ps.done, cur_stream()), "hipEventRecord(pointer lists)")) ps.busy = true;
  else (void)hipStreamSynchronize(cur_stream());
  return dst;
}
// Operands of a SYNCHRONOUS call may live in plain host memory (the reference's contract: any pointer, result valid on return);
// an MI355X cannot see such memory, so it is staged.  Stream-ordered (async) and batched launches take device-accessible memory only:
// no pointer query, no copy on the fast path.
static bool staging_allowed(size_t batch_count) { return !tls().async && batch_count <= 1; }
static const void* host_input(const void* p, size_t nbytes, size_t batch_count) { return staging_allowed(batch_count) ? stage(p, nbytes, true, false) : p; }
static void* host_inout(void* p, size_t nbytes, size_t batch_count) { return staging_allowed(batch_count) ? stage(p, nbytes, true, true) : p; }
// Rewinds the staging scratch at the start of a call.  NOT inside an open pipeline section: its launches run on different lane streams, so call N + 1's
// upload of a staged operand (an OFFSET / ADDRESS list, a gather index list, a BCSC pattern) is not ordered behind call N's kernel, which may not have
// read the same bytes yet -- the scratch keeps growing until the first call after libxsmm_hip_pipeline_end (ordered behind every lane by the join).
Scratch& cur_scratch() { ThreadState& t = tls(); return t.pipe_lanes > 1 ? t_scratch_lane[t.pipe_cur & 7] : t_scratch; }
void scratch_reset() {
  if (t_nest > 0) return;
  if (tls().pipe_lanes > 1) { t_scratch_lane[tls().pipe_cur & 7].used = 0; return; }      // this lane's scratch: its previous call is ahead of this one on the lane's stream
  t_scratch.used = 0; t_copyback.clear();
}
void copy_back_staged() {
  for (const CopyBack& c : t_copyback)
    (void)hip_ok(c.height == 1 ? hipMemcpy(c.host, c.dev, c.width, hipMemcpyDeviceToHost) : hipMemcpy2D(c.host, c.pitch, c.dev, c.pitch, c.width, c.height, hipMemcpyDeviceToHost),
                 "hipMemcpy(staged result)");
  t_copyback.clear();
}
// per-thread device workspace for partial results; grows monotonically, reused in stream order
struct Workspace { void* base = nullptr; size_t cap = 0; int device = 0; ~Workspace() { retire_block(base); } };
thread_local Workspace t_workspace[8];    // one per pipeline lane: launches that overlap must not share partial-result buffers ([0] outside a section)
thread_local size_t t_ws_reserved = 0;   // front part owned by an enclosing call (the slots of a matrix equation around a GEMM node)
void* workspace(size_t nbytes_wanted);
// an OPTIONAL workspace (a fast path that has a fallback): a failed allocation is not an error of the call
void* workspace_try(size_t nbytes_wanted) {
  ThreadState& t = tls();
  const int e0 = t.last_error; const std::string m0 = t.last_error_msg;
  Workspace& w = t_workspace[t.pipe_lanes > 1 ? t.pipe_cur : 0];
  if (nbytes_wanted + t_ws_reserved <= w.cap && w.device == cur_device()) return (char*)w.base + t_ws_reserved;
  void* nb = nullptr;
  const size_t ncap = std::max<size_t>(nbytes_wanted + t_ws_reserved, 4u << 20);
  if (hipMalloc(&nb, ncap) != hipSuccess) { (void)hipGetLastError(); t.last_error = e0; t.last_error_msg = m0; return nullptr; }
  if (w.base) retire_block(w.base);
  w.base = nb; w.cap = ncap; w.device = cur_device();
  return (char*)w.base + t_ws_reserved;
}
void* workspace(size_t nbytes_wanted) {
  Workspace& w = t_workspace[tls().pipe_lanes > 1 ? tls().pipe_cur : 0];
  const size_t nbytes = nbytes_wanted + t_ws_reserved;
  if (w.base && w.device != cur_device()) { retire_block(w.base); w.base = nullptr; w.cap = 0; }
  if (nbytes > w.cap) {
    if (w.base) { retire_block(w.base); w.base = nullptr; w.cap = 0; }
    const size_t ncap = std::max<size_t>(nbytes, 4u << 20);
    if (!hip_ok(hipMalloc(&w.base, ncap), "hipMalloc(workspace)")) { w.base = nullptr; return nullptr; }
    w.cap = ncap; w.device = cur_device();
  }
  return (char*)w.base + t_ws_reserved;
}   // stream order protects data of the previous call

int alloc_slot_locked() {
  // A slot whose thunk page cannot be made executable is PARKED (never returned to the free list: every later allocation would pop it
  // again and fail although other pages work); the search goes on with the next slot.
  for (;;) {
    int slot = -1;
    if (!g_free_slots.empty()) { slot = g_free_slots.back(); g_free_slots.pop_back(); }
    else if (g_next_slot < std::min(g_slot_limit, g_thunk_pool ? kSlots : kStaticSlots)) slot = g_next_slot++;
    if (slot < 0) return -1;
    if (thunk_ready_locked(slot)) return slot;
    vlog(1, "thunk page of handle slot %d cannot be made executable: slot parked", slot);
  }
}

KernelCtx* new_ctx_locked(Kind kind) {
  const int slot = alloc_slot_locked();
  if (slot < 0) { vlog(1, "out of kernel handles (%d in use)", g_next_slot - (int)g_free_slots.size()); return nullptr; }
  KernelCtx* c = new KernelCtx();
  c->slot = slot; c->kind = kind; c->device = cur_device();
  g_slots[slot] = c;
  return c;
}

// Dry run (no device): pattern "uploads" are host copies (to_device below), released accordingly.
void dev_free(void* p) { if (!p) return; if (g_dryrun) std::free(p); else (void)hipFree(p); }
void free_ctx_locked(KernelCtx* c) {
  if (!c) return;
  dev_free(c->d_ptr); dev_free(c->d_idx); dev_free(c->d_vals); dev_free(c->d_vmap);
  if (c->jit) jit_release(c->jit);
  if (c->eqn) free_meqn_plan(c->eqn);
  for (auto* e : c->bcsc_cache) { if (e->d_block) (void)hipFree(e->d_block); delete e; }
  for (auto* e : c->bcsc_old) { if (e->d_block) (void)hipFree(e->d_block); delete e; }
  if (c->bcsc_bound.d_table) (void)hipFree(c->bcsc_bound.d_table);
  g_slots[c->slot] = nullptr; g_free_slots.push_back(c->slot);
  delete c;
}

bool tilecfg_halfset(unsigned int f) {   // [ref: src/libxsmm_generator.c:154-157]
  const bool a = (f & LIBXSMM_GEMM_FLAG_NO_RESET_TILECONFIG) != 0, b = (f & LIBXSMM_GEMM_FLAG_NO_SETUP_TILECONFIG) != 0;
  return a != b;
}

// Dry run (no device): pattern "uploads" are host copies, so that the creators get as far as code generation (compiled, never launched).
template <typename T> T* to_device(const T* host, size_t count) {
  if (count == 0) count = 1;
  if (g_dryrun) { T* h = (T*)std::calloc(count, sizeof(T)); if (h && host) std::memcpy(h, host, count * sizeof(T)); return h; }
  T* d = nullptr;
  if (hipMalloc((void**)&d, count * sizeof(T)) != hipSuccess) return nullptr;
  if (host && hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  return d;
}

}  // namespace

namespace xamd {
static thread_local int g_window_verdict = 0;          // libxsmm_hip_streaming_window_verdict
bool rt_recent_operands_exceed_cache(const void* key, unsigned long long bytes, const void* out) {
  constexpr int kEntries = 32;                          // 32 operand sets, each forgotten after 96 decisions without being seen again
  struct Entry { const void* key; const void* out; unsigned long long bytes, gen; };
  struct Window { Entry e[kEntries]; unsigned long long gen; };
  static thread_local Window w = {};
  ++w.gen;
  int slot = -1, oldest = 0;
  bool handed_over = false;                             // the first operand is what a remembered launch wrote
  for (int i = 0; i < kEntries; ++i) {
    const bool live = w.e[i].gen != 0 && w.gen - w.e[i].gen < 96ull;
    if (w.e[i].key == key && w.e[i].gen != 0) slot = i;
    if (live && key != nullptr && w.e[i].out == key && w.e[i].key != key && w.gen - w.e[i].gen <= 8ull) handed_over = true;      // (written within the last few launches: an address a finished phase of the caller wrote to and freed is not a hand-over)
    if (w.e[i].gen < w.e[oldest].gen) oldest = i;
  }
  if (slot < 0) slot = oldest;
  w.e[slot] = Entry{key, out, bytes, w.gen};
  unsigned long long sum = 0;
  for (int i = 0; i < kEntries; ++i) if (w.e[i].gen != 0 && w.gen - w.e[i].gen < 96ull) sum += w.e[i].bytes;
  g_window_verdict = (!handed_over && sum > (256ull << 20)) ? 1 : 0;
  return g_window_verdict != 0;
}
int rt_window_verdict() { return g_window_verdict; }

ThreadState& tls() {
  thread_local ThreadState st;
  if (st.async < 0) {
    const char* a = std::getenv("LIBXSMM_HIP_ASYNC"); const char* s = std::getenv("LIBXSMM_HIP_SYNC");
    st.async = (a && std::atoi(a) != 0) ? (std::atoi(a) == 2 ? 2 : 1) : 0;
    if (const char* co = std::getenv("LIBXSMM_HIP_COALESCE")) { if (std::atoi(co) != 0) st.async = 2; }
    if (s && std::atoi(s) != 0) st.async = 0;
    const char* h = std::getenv("LIBXSMM_HIP_STREAMING");
    if (h) { const int v = std::atoi(h); st.stream_hint = (v >= 0 && v <= 2) ? v : 0; }
  }
  return st;
}

void set_error(int code, const char* fmt, ...) {
  ThreadState& t = tls();
  char buf[512];
  va_list ap; va_start(ap, fmt); std::vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  t.last_error = code ? code : -1; t.last_error_msg = buf;
  // kernels have no error channel: a failed launch is reported even when the library is otherwise mute
  std::fprintf(stderr, "LIBXSMM-AMD ERROR: %s\n", buf);
}

int typesize(int t) { return (t >= 0 && t < (int)LIBXSMM_DATATYPE_COUNT_) ? (int)kTypeSizes[t] : 0; }

bool runtime_ready() {
  if (libxsmm_ninit < 2) libxsmm_init();
  if (g_device_count > 0 || g_dryrun) return true;
  if (!g_warned_nodevice) {
    g_warned_nodevice = true;
    std::fprintf(stderr, "LIBXSMM-AMD ERROR: no HIP device visible -- this backend has no CPU path; every dispatch returns NULL\n");
  }
  return false;
}

static int slot_from_handle(const void* fn) {
  if (g_thunk_pool) {
    const size_t off = (size_t)((const unsigned char*)fn - g_thunk_pool);
    if ((const unsigned char*)fn >= g_thunk_pool && off < (size_t)kSlots * kThunkBytes && off % kThunkBytes == 0) return (int)(off / kThunkBytes);
    return -1;
  }
  for (int i = 0; i < kStaticSlots; ++i) if ((const void*)g_tramps[i] == fn) return i;
  return -1;
}
KernelCtx* ctx_from_handle(const void* fn) {
  if (!fn) return nullptr;
  const int slot = slot_from_handle(fn);
  return slot >= 0 ? g_slots[slot] : nullptr;
}
const void* handle_for_slot(int slot) { return g_thunk_pool ? (const void*)(g_thunk_pool + (size_t)slot * kThunkBytes) : (const void*)g_tramps[slot]; }

}  // namespace xamd

// =====================================================================================================
// invocation: param struct -> argument block -> launch
// =====================================================================================================
static void attach_jit(KernelCtx* c, const unsigned int* ptr, const unsigned int* idx, const unsigned int* vmap, long long batch = 1);
namespace {

struct BatchSpec {
  size_t count = 1;
  long long s[5] = {0, 0, 0, 0, 0};                 // kind specific byte strides
  size_t inner = 0; long long c2 = 0, mask2 = 0;    // 2-D GEMM batch: count = inner * outer, second strides of C / bitmask
  const void* const* la = nullptr; const void* const* lb = nullptr; void* const* lc = nullptr;
  bool lists_on_host = false;                       // the coalescing queue's lists: plain host vectors, staged without a pointer query
  bool lists_aligned16 = false;                     // ... and every pointer in them is known to be 16-byte aligned (the fast kernels take pointer lists then)
};

// LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK [ref: gemm ref :535-556,857-948]: a.primary = the non-zeros of A, a.secondary = one bit per element.
// The dense image is rebuilt in the workspace (three small launches), then the dense kernel of the equivalent descriptor runs on it.
void coalesce_flush();       // the coalescing queue (further down): every launch path drains it first, so launches keep the caller's order
static void run_gemm_bitmask(KernelCtx* k, const libxsmm_gemm_param* p, const BatchSpec& b) {
  const libxsmm_gemm_descriptor& d = k->g;
  if (b.count != 1 || b.la) { set_error(-2, "a GEMM with bitmask-compressed A cannot be batched (its operand size differs per problem)"); return; }
  if (!p->a.primary || !p->a.secondary || !p->b.primary || !p->c.primary) { set_error(-2, "GEMM with bitmask-compressed A: a.primary / a.secondary / b.primary / c.primary is NULL"); return; }
  scratch_reset();
  const int es = typesize(d.a_type), kb = (d.a_type == LIBXSMM_DATATYPE_F32) ? 1 : 2;
  const int rows = (int)d.k / kb, row_bytes = (int)d.m * kb / 8;
  const size_t bitmap_bytes = (size_t)rows * (size_t)row_bytes, dense_bytes = bitmap_bytes * 8 * (size_t)es;
  GemmArgs a{};
  const void* bitmap = p->a.secondary; const void* vals = p->a.primary;
  a.b = (const char*)p->b.primary; a.c = (char*)p->c.primary;
  if (staging_allowed(1)) {
    hipPointerAttribute_t attr;
    const bool host_bits = hipPointerGetAttributes(&attr, bitmap) != hipSuccess || attr.type == hipMemoryTypeUnregistered;
    (void)hipGetLastError();
    if (host_bits) {     // plain host memory: the length of the value array is the number of set bits, countable right here
      size_t nnz = 0;
      const unsigned char* hb = (const unsigned char*)bitmap;
      for (size_t i = 0; i < bitmap_bytes; ++i) nnz += (size_t)__builtin_popcount(hb[i]);
      bitmap = stage(bitmap, bitmap_bytes, true, false);
      vals = stage(vals, std::max<size_t>(nnz, 1) * (size_t)es, true, false);
    }
    a.b = (const char*)stage(a.b, (size_t)d.ldb * (size_t)d.n * (size_t)typesize(d.b_type), true, false);
    a.c = (char*)stage(a.c, (size_t)d.ldc * (size_t)d.n * (size_t)typesize(d.c_type), true, true);
    if (!bitmap || !vals || !a.b || !a.c) return;
  }
  const char* kname = nullptr;
  // round 4: the expansion in registers (gemm_bitmask_kernels.hip).  The shape decides first, then exactly the workspace that shape needs is requested;
  // an allocation that fails only loses the fast path (no sticky error: the other paths follow)
  {
    GemmArgs q{};
    q.a = (const char*)vals; q.b = a.b; q.c = a.c;
    q.nbatch = 1; q.m = (int)d.m; q.n = (int)d.n; q.k = (int)d.k; q.lda = (int)d.m; q.ldb = (int)d.ldb; q.ldc = (int)d.ldc;
    q.flags = d.flags; q.a_type = d.a_type; q.b_type = d.b_type; q.c_type = d.c_type; q.br_count = 1; q.br_mode = 0;
    const size_t need = gemm_bitmask_reg_workspace(q);
    void* rws = need ? workspace_try(need) : nullptr;
    if (rws) {
      int taken = 0;
      const int ferr = launch_gemm_bitmask_reg(q, bitmap, rws, need, tls().stream, &kname, &taken);
      if (taken) { if (kname) k->kname_single = kname; finish_launch(ferr, kname); return; }
    }
  }
  // 16-bit operands: the fused kernel multiplies straight out of (non-zeros, bitmap): no dense image is written or read (round 3)
  {
    // counts / offsets per (bit row, tile of 128 rows), then room for the f32 partial tiles of up to 32 slices of k (fewer slices if the workspace is short)
    const size_t tiles = ((size_t)d.m + 127) / 128, table_bytes = ((((size_t)rows * (tiles + 2)) + 63) & ~(size_t)63) * sizeof(unsigned int);
    const size_t part_bytes = (size_t)d.m * (size_t)d.n * sizeof(float);
    size_t want_bytes = table_bytes + 32 * part_bytes;
    if (want_bytes > ((size_t)1 << 30)) want_bytes = std::max(table_bytes + 2 * part_bytes, (size_t)1 << 30);
    unsigned int* fs = (es == 2) ? (unsigned int*)workspace(want_bytes) : nullptr;
    if (fs) {
      a.a = (const char*)vals;
      a.nbatch = 1; a.stream_hint = tls().stream_hint;
      a.m = (int)d.m; a.n = (int)d.n; a.k = (int)d.k; a.lda = (int)d.m; a.ldb = (int)d.ldb; a.ldc = (int)d.ldc;
      a.flags = d.flags; a.a_type = d.a_type; a.b_type = d.b_type; a.c_type = d.c_type; a.br_count = 1; a.br_mode = 0;
      int taken = 0;
      const int ferr = launch_gemm_bitmask16(a, bitmap, fs, want_bytes, tls().stream, &kname, &taken);
      if (taken) { if (kname) k->kname_single = kname; finish_launch(ferr, kname); return; }
    }
  }
  const size_t scratch_bytes = ((size_t)rows * 2 * sizeof(unsigned int) + 255) & ~(size_t)255;
  char* ws = (char*)workspace(scratch_bytes + dense_bytes);
  if (!ws) return;
  int err = launch_bitmask_expand(bitmap, vals, ws + scratch_bytes, (unsigned int*)ws, rows, row_bytes, es, tls().stream);
  if (err == 0) {
    a.a = ws + scratch_bytes;
    a.nbatch = 1; a.stream_hint = tls().stream_hint;
    a.m = (int)d.m; a.n = (int)d.n; a.k = (int)d.k; a.lda = (int)d.m; a.ldb = (int)d.ldb; a.ldc = (int)d.ldc;
    a.flags = (d.flags & ~(unsigned int)LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK) | (kb == 2 ? (unsigned int)LIBXSMM_GEMM_FLAG_VNNI_A : 0u);
    a.a_type = d.a_type; a.b_type = d.b_type; a.c_type = d.c_type;
    a.br_count = 1; a.br_mode = 0;
    rt_workspace_reserve(scratch_bytes + dense_bytes);     // a kernel that needs partial sums of its own places them behind the dense image
    err = launch_gemm(a, tls().stream, &kname);
    rt_workspace_reserve(0);
  }
  if (kname) k->kname_single = kname;
  finish_launch(err, kname ? kname : "bitmask_expand");
}

// Flags that mean nothing for the descriptor's operand types are IGNORED, as the reference does (its JIT and its C loop alike; tests/test_dispatch_differential_cpu.py):
// VNNI_A / VNNI_B on f32 / f64 operands (no VNNI layout exists for them [ref: gemm ref :1359-1426]), INTLV_A_FORMAT on types that have no interleaved format
// (it belongs to 4-bit / 2-bit weights [ref: gemm ref :467-486]).  The handle still reports the caller's flags (libxsmm_get_mmkernel_info).
unsigned int effective_gemm_flags(const libxsmm_gemm_descriptor& d) {
  unsigned int f = d.flags;
  if ((d.a_type == LIBXSMM_DATATYPE_F32 && d.b_type == LIBXSMM_DATATYPE_F32) || (d.a_type == LIBXSMM_DATATYPE_F64 && d.b_type == LIBXSMM_DATATYPE_F64))
    f &= ~(unsigned int)(LIBXSMM_GEMM_FLAG_VNNI_A | LIBXSMM_GEMM_FLAG_VNNI_B);
  const bool has_intlv = d.a_type == LIBXSMM_DATATYPE_I4X2 || d.a_type == LIBXSMM_DATATYPE_U4X2 || d.a_type == LIBXSMM_DATATYPE_MXFP4X2 || d.a_type == LIBXSMM_DATATYPE_I2X4 || d.a_type == LIBXSMM_DATATYPE_I1X8;
  if (!has_intlv) f &= ~(unsigned int)LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT;
  // 8-bit integers with an f32 result: the reference's loop reads A as VNNI-4 whether or not the caller's flags say so [ref: gemm ref :1556-1683, l_k_block = 4] (round 6)
  const auto is_i8 = [](int t) { return t == LIBXSMM_DATATYPE_I8 || t == LIBXSMM_DATATYPE_U8; };
  if (is_i8(d.a_type) && is_i8(d.b_type) && d.c_type == LIBXSMM_DATATYPE_F32) f |= (unsigned int)LIBXSMM_GEMM_FLAG_VNNI_A;
  return f;
}
void run_gemm(KernelCtx* k, const void* param, const BatchSpec& b) {
  coalesce_flush();
  const libxsmm_gemm_descriptor& d = k->g;
  if (d.flags & LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK) { run_gemm_bitmask(k, (const libxsmm_gemm_param*)param, b); return; }
  const bool ext = (d.flags & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI) != 0;
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;          // {op,a,b,c} prefix is common
  const libxsmm_gemm_ext_param* pe = (const libxsmm_gemm_ext_param*)param;
  GemmArgs a{};
  scratch_reset();
  a.a = (const char*)p->a.primary; a.b = (const char*)p->b.primary; a.c = (char*)p->c.primary;
  a.list_a = b.la; a.list_b = b.lb; a.list_c = b.lc;
  if (b.lists_on_host) {
    a.list_a = (const void* const*)stage_host(b.la, b.count * sizeof(void*)); a.list_b = (const void* const*)stage_host(b.lb, b.count * sizeof(void*));
    a.list_c = (void* const*)stage_host(b.lc, b.count * sizeof(void*));
    if (!a.list_a || !a.list_b || !a.list_c) return;
  }
  a.lists_aligned16 = (b.la && b.lists_aligned16) ? 1 : 0;
  a.bs_a = b.s[0]; a.bs_b = b.s[1]; a.bs_c = b.s[2]; a.bs_d = b.s[3]; a.bs_mask = b.s[4];
  a.nbatch = (unsigned int)b.count;
  a.batch_inner = (unsigned int)b.inner; a.bs_c2 = b.c2; a.bs_mask2 = b.mask2;
  a.stream_hint = tls().stream_hint;
  a.m = (int)d.m; a.n = (int)d.n; a.k = (int)d.k; a.lda = (int)d.lda; a.ldb = (int)d.ldb; a.ldc = (int)d.ldc;
  a.flags = d.flags; a.a_type = d.a_type; a.b_type = d.b_type; a.c_type = d.c_type;
  a.flags = effective_gemm_flags(d);
  a.vnni_c = (d.flags & LIBXSMM_GEMM_FLAG_VNNI_C) ? 1 : 0;
  a.comp_f16 = (d.a_type == LIBXSMM_DATATYPE_F16 && d.comp_type == LIBXSMM_DATATYPE_F16) ? 1 : 0;
  a.br_count = 1; a.br_mode = 0;
  if (d.flags & (LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE)) {
    // the count is re-read on every call [ref: gemm ref :490-492; SURVEY Appendix B.4]
    if (!p->op.tertiary) { set_error(-2, "BRGEMM kernel called without op.tertiary (batch-reduce count)"); return; }
    a.br_count = *(const unsigned long long*)p->op.tertiary;
  }
  if (d.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS) {
    a.br_mode = 1;
    if (b.la == nullptr) {
      // a/b.primary are pointer lists; with a strided batch each element has its own list
      const size_t span = (size_t)a.br_count * sizeof(void*) + (size_t)(b.count - 1) * (size_t)std::max<long long>(std::max(b.s[0], b.s[1]), 0);
      a.a = (const char*)device_visible(p->a.primary, span); a.b = (const char*)device_visible(p->b.primary, span);
      if (!a.a || !a.b) return;
    }
  } else if (d.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET) {
    a.br_mode = 2;
    a.offs_a = (const long long*)device_visible(p->a.secondary, (size_t)a.br_count * sizeof(long long));
    a.offs_b = (const long long*)device_visible(p->b.secondary, (size_t)a.br_count * sizeof(long long));
    if ((!a.offs_a || !a.offs_b) && a.br_count) { set_error(-2, "OFFSET-BRGEMM kernel called without a/b.secondary offset arrays"); return; }
  } else if (d.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE) {
    a.br_mode = 3; a.br_stride_a = d.br_stride_a; a.br_stride_b = d.br_stride_b;
  }
  if ((d.a_type == LIBXSMM_DATATYPE_I8 || d.a_type == LIBXSMM_DATATYPE_U8) && (d.b_type == LIBXSMM_DATATYPE_I8 || d.b_type == LIBXSMM_DATATYPE_U8) && d.c_type == LIBXSMM_DATATYPE_F32) {
    if (!p->c.tertiary) { set_error(-2, "8-bit GEMM with f32 output needs the scale in c.tertiary"); return; }   // [ref: gemm ref :591-592]
    a.scf = *(const float*)p->c.tertiary;
  }
  const bool a_fp6 = d.a_type == LIBXSMM_DATATYPE_MXBF6 || d.a_type == LIBXSMM_DATATYPE_MXHF6;
  char* c_scf = nullptr;                       // MX-typed C: where its block scales go
  bool i8_rows = false;                        // i8 weights x bf16: a_scf = one f32 per row
  bool intlv4 = false;                         // interleaved 4-bit weights x 8-bit activations: a_scf = zero points or E8M0 scales, b_scf = f32 scales
  if ((d.a_type == LIBXSMM_DATATYPE_MXFP4X2 || d.a_type == LIBXSMM_DATATYPE_MXBF8 || d.a_type == LIBXSMM_DATATYPE_MXHF8 || a_fp6) && d.b_type == d.a_type) {
    // MX x MX: scales of A in a.tertiary, of B in b.tertiary [ref: gemm ref :577-583]; a batched launch steps them with their operand:
    // one scale byte per 32 elements
    if (!p->a.tertiary || !p->b.tertiary) { set_error(-2, "MX x MX GEMM needs the E8M0 scales in a.tertiary and b.tertiary"); return; }
    if (b.la) { set_error(-3, "MX x MX GEMM: pointer-list batches carry no scale lists; use the strided batch"); return; }
    const long long epb = (d.a_type == LIBXSMM_DATATYPE_MXFP4X2) ? 2 : 1;
    if (b.count > 1 && (a_fp6 ? ((b.s[0] % 24) != 0 || (b.s[1] % 24) != 0) : ((((b.s[0] | b.s[1]) * epb) % 32) != 0))) { set_error(-3, "MX x MX GEMM: batch strides must cover whole 32-element scale blocks"); return; }
    if (d.c_type == d.a_type && (d.a_type == LIBXSMM_DATATYPE_MXFP4X2 || d.a_type == LIBXSMM_DATATYPE_MXBF8)) {
      if (!p->c.tertiary) { set_error(-2, "MX-typed C needs room for its E8M0 scales in c.tertiary"); return; }
      if (b.la || b.inner) { set_error(-3, "MX-typed C: strided batches only"); return; }
      c_scf = (char*)p->c.tertiary;
    }
    a.a_scf = (const char*)p->a.tertiary; a.bs_scf = a_fp6 ? b.s[0] / 24 : b.s[0] * epb / 32;       // 6-bit: 32 elements are 24 bytes
    a.b_scf = (const char*)p->b.tertiary; a.bs_bscf = a_fp6 ? b.s[1] / 24 : b.s[1] * epb / 32;
  } else if ((d.flags & LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT) && (d.a_type == LIBXSMM_DATATYPE_I4X2 || d.a_type == LIBXSMM_DATATYPE_U4X2 || d.a_type == LIBXSMM_DATATYPE_MXFP4X2)) {
    // interleaved 4-bit weights x 8-bit activations [ref: gemm ref :565-575, :1009-1088, :1272-1330].  I4X2: one zero point per row in
    // a.quaternary.  MXFP4: E8M0 scales of A in a.tertiary, one f32 per (column, 32-deep block) of B in b.tertiary.  Batches step them with A / B.
    intlv4 = true;
    if (b.la) { set_error(-3, "interleaved 4-bit GEMM: strided batches only"); return; }
    if (d.a_type != LIBXSMM_DATATYPE_MXFP4X2) {
      if (!p->a.quaternary) { set_error(-2, "I4X2 GEMM needs the zero points in a.quaternary"); return; }
      a.a_scf = (const char*)p->a.quaternary; a.bs_scf = b.s[0] * 2 / std::max<long long>(d.k, 1);
    } else {
      if (!p->a.tertiary || !p->b.tertiary) { set_error(-2, "MXFP4 x I8 GEMM needs the E8M0 scales of A in a.tertiary and the f32 scales of B in b.tertiary"); return; }
      if (b.count > 1 && ((b.s[0] % 16) != 0 || (b.s[1] % 32) != 0)) { set_error(-3, "MXFP4 x I8 GEMM: batch strides must cover whole 32-element scale blocks"); return; }
      a.a_scf = (const char*)p->a.tertiary; a.bs_scf = b.s[0] / 16;
      a.b_scf = (const char*)p->b.tertiary; a.bs_bscf = b.s[1] / 8;            // one f32 per 32 bytes of B
    }
  } else if (d.a_type == LIBXSMM_DATATYPE_I8 && d.b_type == LIBXSMM_DATATYPE_BF16) {
    // one f32 scale per row of the i8 weights in a.tertiary [ref: gemm ref :1684-1730]; a strided batch steps them with A (lda floats per k * lda weights)
    if (!p->a.tertiary) { set_error(-2, "I8 x BF16 GEMM needs the row scales in a.tertiary"); return; }
    if (b.la) { set_error(-3, "I8 x BF16 GEMM: strided batches only"); return; }
    a.a_scf = (const char*)p->a.tertiary; a.bs_scf = (b.s[0] / std::max<long long>(d.k, 1)) * 4;
    i8_rows = true;
  } else if (d.a_type == LIBXSMM_DATATYPE_MXFP4X2) {
    // E8M0 scales of the MXFP4 weights travel in a.tertiary (a list of per-block pointers in ADDRESS mode) [ref: gemm ref :565-569].
    // A batched launch steps them like A: the pointer list by sa, the scale bytes by sa * 2 / 32 (one byte per 32 weights).
    if (!p->a.tertiary) { set_error(-2, "MXFP4 GEMM needs the E8M0 scales in a.tertiary"); return; }
    if (b.la) { set_error(-3, "MXFP4 GEMM: pointer-list batches carry no scale list; use the strided batch"); return; }
    if (a.br_mode == 1) {
      const size_t span = (size_t)a.br_count * sizeof(void*) + (size_t)(b.count - 1) * (size_t)std::max<long long>(b.s[0], 0);
      a.a_scf = (const char*)device_visible(p->a.tertiary, span); a.bs_scf = b.s[0];
      if (!a.a_scf) return;
    } else {
      if (b.count > 1 && (b.s[0] % 16) != 0) { set_error(-3, "MXFP4 GEMM: the batch stride of A must cover whole 32-weight scale blocks (multiple of 16 bytes)"); return; }
      a.a_scf = (const char*)p->a.tertiary; a.bs_scf = b.s[0] / 16;
    }
  }
  if (ext) {
    // fused epilogue decoded as the reference does [ref: gemm ref :404-428]
    if (d.bin_type == LIBXSMM_MELTW_TYPE_BINARY_ADD &&
        (d.bin_flags & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0 | LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_1))) {
      a.colbias = 1; a.d = (const char*)pe->d.primary;
      if (!a.d) { set_error(-2, "fused column bias requested but d.primary is NULL"); return; }
    }
    if (d.cp_type == LIBXSMM_MELTW_TYPE_UNARY_RELU) {
      a.act = (d.cp_flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) ? 2 : 1;
      if (a.act == 2) { a.relu_mask = (unsigned char*)pe->c.secondary; if (!a.relu_mask) { set_error(-2, "ReLU bitmask requested but c.secondary is NULL"); return; } }
    } else if (d.cp_type == LIBXSMM_MELTW_TYPE_UNARY_SIGMOID) a.act = 3;
  }
  // A synchronous single call may be handed plain host memory (the reference's hello-world mallocs its matrices): operands whose
  // extent follows from the descriptor alone (no pointer lists, no offset arrays) are staged.  One pointer query per operand.
  if (staging_allowed(b.count) && !a.list_a && (a.br_mode == 0 || a.br_mode == 3) && a.br_count >= 1) {
    const bool ta = (d.flags & LIBXSMM_GEMM_FLAG_TRANS_A) != 0, tb = (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B) != 0;
    const bool fp6 = d.a_type == LIBXSMM_DATATYPE_MXBF6 || d.a_type == LIBXSMM_DATATYPE_MXHF6;
    const bool mxmx = (d.a_type == LIBXSMM_DATATYPE_MXFP4X2 || d.a_type == LIBXSMM_DATATYPE_MXBF8 || d.a_type == LIBXSMM_DATATYPE_MXHF8 || fp6) && d.b_type == d.a_type;
    const auto bytes_of = [](int type, size_t elems) { return (type == LIBXSMM_DATATYPE_MXFP4X2 || type == LIBXSMM_DATATYPE_I4X2 || type == LIBXSMM_DATATYPE_U4X2) ? elems / 2 : (type == LIBXSMM_DATATYPE_MXBF6 || type == LIBXSMM_DATATYPE_MXHF6) ? elems * 3 / 4 :
                                                             type == LIBXSMM_DATATYPE_I1X8 ? elems / 8 : type == LIBXSMM_DATATYPE_I2X4 ? elems / 4 : elems * (size_t)typesize(type); };
    const size_t ea = bytes_of(d.a_type, (size_t)a.lda * (size_t)(ta ? a.m : a.k));
    const size_t eb = bytes_of(d.b_type, (size_t)a.ldb * (size_t)((tb || mxmx) ? a.k : a.n));
    const size_t ec = bytes_of(d.c_type, (size_t)a.ldc * (size_t)(a.n + (a.vnni_c ? (a.n & 1) : 0)));
    const size_t span = (a.br_mode == 3) ? (size_t)(a.br_count - 1) : 0;
    if (a.br_mode != 3 || (a.br_stride_a >= 0 && a.br_stride_b >= 0)) {
      a.a = (const char*)stage(a.a, span * (size_t)a.br_stride_a + ea, true, false);
      a.b = (const char*)stage(a.b, span * (size_t)a.br_stride_b + eb, true, false);
      a.c = (char*)stage(a.c, ec, true, true);
      if (a.d) a.d = (const char*)stage(a.d, (size_t)a.m * (size_t)typesize(d.c_type), true, false);
      if (a.relu_mask) a.relu_mask = (unsigned char*)stage(a.relu_mask, (size_t)(((a.ldc + 15) / 16) * 16 / 8) * (size_t)a.n, true, true);
      // E8M0 scales: one byte per 32 elements, so a batch-reduce element is (stride * elements-per-byte / 32) bytes further on
      const size_t epb_a = (d.a_type == LIBXSMM_DATATYPE_MXFP4X2) ? 2 : 1, epb_b = (d.b_type == LIBXSMM_DATATYPE_MXFP4X2) ? 2 : 1;
      const size_t sc_step_a = fp6 ? (size_t)a.br_stride_a / 24 : (size_t)a.br_stride_a * epb_a / 32, sc_step_b = fp6 ? (size_t)a.br_stride_b / 24 : (size_t)a.br_stride_b * epb_b / 32;
      if (i8_rows) a.a_scf = (const char*)stage(a.a_scf, (size_t)a.lda * sizeof(float), true, false);
      else if (intlv4 && d.a_type != LIBXSMM_DATATYPE_MXFP4X2) a.a_scf = (const char*)stage(a.a_scf, span * ((size_t)a.br_stride_a * 2 / (size_t)a.k) + (size_t)a.lda, true, false);
      else if (intlv4) {
        a.a_scf = (const char*)stage(a.a_scf, span * ((size_t)a.br_stride_a / 16) + (size_t)a.lda * (size_t)(a.k / 32), true, false);
        a.b_scf = (const char*)stage(a.b_scf, (span * ((size_t)a.br_stride_b / 32) + (size_t)(a.ldb / 32) * (size_t)a.n) * sizeof(float), true, false);
      } else {
      if (a.a_scf) a.a_scf = (const char*)stage(a.a_scf, span * sc_step_a + (size_t)a.lda * (size_t)(a.k / 32), true, false);
      if (a.b_scf) a.b_scf = (const char*)stage(a.b_scf, span * sc_step_b + (size_t)a.ldb * (size_t)(a.k / 32), true, false);
      }
      if (c_scf) c_scf = (char*)stage(c_scf, (size_t)(a.ldc / 32) * (size_t)a.n, true, true);     // copied in as well: the entries of padded rows stay the caller's
      if (!a.a || !a.b || !a.c) return;
    }
  }
  const char* kname = nullptr;
  if (c_scf) {
    // MX-typed C [ref: gemm ref :2666-2678, :2787-2798]: the product goes to an f32 image in the workspace, a second kernel quantises it in
    // 32-row blocks (data to c.primary, E8M0 scales to c.tertiary); a batched launch steps the scales with C (one byte per 32 elements)
    const bool fp4 = d.c_type == LIBXSMM_DATATYPE_MXFP4X2;
    const size_t img = (size_t)a.ldc * (size_t)a.n * sizeof(float);
    float* ws = (float*)workspace(img * a.nbatch);
    if (!ws) return;
    if (a.nbatch > 1 && ((a.bs_c * (fp4 ? 2 : 1)) % 32) != 0) { set_error(-3, "MX-typed C: the batch stride must cover whole 32-element scale blocks"); return; }
    GemmArgs pa = a;
    pa.c = (char*)ws; pa.c_type = LIBXSMM_DATATYPE_F32; pa.bs_c = (long long)img; pa.bs_c2 = 0; pa.batch_inner = 0; pa.list_c = nullptr;
    rt_workspace_reserve(img * a.nbatch);
    int err = launch_gemm(pa, tls().stream, &kname);
    rt_workspace_reserve(0);
    if (err == 0) err = launch_mx_out_quant(ws, a.c, c_scf, a.m, a.n, a.ldc, fp4 ? 1 : 0, a.nbatch, a.bs_c, a.bs_c * (fp4 ? 2 : 1) / 32, tls().stream);
    if (kname) { if (b.count > 1) k->kname_batched = kname; else k->kname_single = kname; }
    finish_launch(err, kname);
    return;
  }
  // One (or very few) problems with a long STRIDE batch-reduce chain would run on a handful of waves: split the chain
  // into `nsplit` segments that run as a batch of partial products (f32 tiles in a workspace), then add them up and
  // apply beta / bias / activation in a second pass (SURVEY 8(d) config #2 variant B: one BRGEMM with br = 4096).
  constexpr bool split_off = false;
  const long long tiles = (long long)((a.m + 31) / 32) * ((a.n + 31) / 32);
  if (!split_off && a.br_mode == 3 && a.nbatch == 1 && !a.list_a && a.br_count >= 16 && tiles * 8 <= 1024 && (a.a_type == LIBXSMM_DATATYPE_F32 || a.a_type == LIBXSMM_DATATYPE_BF16) && a.b_type == a.a_type && a.m > 0 && a.n > 0) {
    const size_t tile_bytes = (size_t)a.m * a.n * sizeof(float);
    // f32 chains of whole 32 x 32 tiles: eight waves per slice add their accumulators up on chip (one partial tile per 8 * chunk blocks)
    {
      const size_t cap = (size_t)((a.br_count + 7) / 8);
      float* ws8 = (float*)workspace(cap * tile_bytes);
      if (ws8) {
        int nslices = 0;
        int err = launch_brchain_f32(a, ws8, cap, &nslices, tls().stream, &kname);
        if (nslices > 0) {
          if (err == 0) err = launch_brsplit_reduce(a, ws8, nslices, tls().stream);
          if (kname) k->kname_single = kname;
          finish_launch(err, kname);
          return;
        }
      }
    }
    // everything else (bf16, transposes, ragged tiles): the chain as a batch of partial products on the tile kernels, as many waves as a launch of
    // 4096 problems has (round 2 stopped at 2048 / tiles segments of >= 4 blocks: a quarter of the chip's waves)
    unsigned long long nsplit = std::min<unsigned long long>(a.br_count / 2, (unsigned long long)(4096 / tiles));
    const unsigned long long chunk = (a.br_count + nsplit - 1) / nsplit;
    const unsigned long long nfull = a.br_count / chunk, tail = a.br_count - nfull * chunk;
    nsplit = nfull + (tail ? 1 : 0);
    float* ws = (float*)workspace(nsplit * tile_bytes);
    if (ws) {
      GemmArgs pa = a;
      pa.c = (char*)ws; pa.ldc = a.m; pa.c_type = LIBXSMM_DATATYPE_F32; pa.vnni_c = 0;
      pa.flags = (a.flags | LIBXSMM_GEMM_FLAG_BETA_0) & ~(unsigned int)LIBXSMM_GEMM_FLAG_VNNI_C;
      pa.d = nullptr; pa.relu_mask = nullptr; pa.colbias = 0; pa.act = 0;
      pa.batch_inner = 0; pa.bs_c2 = 0; pa.bs_mask2 = 0;      // the partial products are a plain 1-D batch whatever the caller's (1 x 1) batch form was
      pa.br_count = chunk; pa.nbatch = (unsigned int)nfull;
      pa.bs_a = (long long)chunk * a.br_stride_a; pa.bs_b = (long long)chunk * a.br_stride_b; pa.bs_c = (long long)tile_bytes; pa.bs_d = 0; pa.bs_mask = 0;
      int err = launch_gemm(pa, tls().stream, &kname);
      if (err == 0 && tail) {
        pa.a = a.a + (long long)nfull * pa.bs_a; pa.b = a.b + (long long)nfull * pa.bs_b; pa.c = (char*)ws + nfull * tile_bytes;
        pa.br_count = tail; pa.nbatch = 1;
        err = launch_gemm(pa, tls().stream, nullptr);
      }
      if (err == 0) err = launch_brsplit_reduce(a, ws, (int)nsplit, tls().stream);
      if (kname) k->kname_single = kname;
      finish_launch(err, kname);
      return;
    }
  }
  const int err = launch_gemm(a, tls().stream, &kname);
  if (kname) { if (b.count > 1 || b.la) k->kname_batched = kname; else k->kname_single = kname; }   // what actually ran
  finish_launch(err, kname);
}

void run_meltw(KernelCtx* k, const void* param, const BatchSpec& b) {
  coalesce_flush();
  const libxsmm_meltw_descriptor& d = k->e;
  MeltwArgs a{};
  scratch_reset();
  a.nbatch = (unsigned int)b.count;
  a.m = (int)d.m; a.n = (int)d.n; a.ldi = (int)d.ldi; a.ldi1 = (int)d.ldi2; a.ldi2 = (int)d.ldi3; a.ldo = (int)d.ldo;
  a.in0_type = d.in0_type; a.in1_type = d.in1_type; a.in2_type = d.in2_type; a.out_type = d.out_type; a.comp_type = d.comp_type;
  a.flags = d.flags; a.type = d.param; a.operation = d.operation;
  if (d.operation == LIBXSMM_MELTW_OPERATION_UNARY) {
    const libxsmm_meltw_unary_param* p = (const libxsmm_meltw_unary_param*)param;
    a.in0 = (const char*)p->in.primary; a.out = (char*)p->out.primary;
    a.aux_in = p->in.secondary; a.aux_out = p->out.secondary;
    a.bs_in0 = b.s[0]; a.bs_out = b.s[1]; a.bs_aux = b.s[2];
    const int t = d.param;
    if ((d.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_COLS) && d.n >= 2048 && b.count == 1) {   // one big matrix reduced over its columns: two passes
      a.ws_bytes = (size_t)128 * 2 * (size_t)d.m * sizeof(float); a.ws = workspace(a.ws_bytes);
    }
    if (d.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) {       // a missing mask pointer would be a device fault: say so instead
      const bool fwd = t == LIBXSMM_MELTW_TYPE_UNARY_RELU || t == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU || t == LIBXSMM_MELTW_TYPE_UNARY_ELU;
      const bool inv = t == LIBXSMM_MELTW_TYPE_UNARY_RELU_INV || t == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV || t == LIBXSMM_MELTW_TYPE_UNARY_ELU_INV;
      if (fwd && !p->out.secondary) { set_error(-2, "unary TPP with BITMASK_2BYTEMULT needs the mask destination in out.secondary"); return; }
      if (inv && !p->in.secondary) { set_error(-2, "unary *_INV TPP with BITMASK_2BYTEMULT needs the mask in in.secondary"); return; }
    }
    // scalars the reference reads through op.primary / out.secondary are read here, on the host
    if (t == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU || t == LIBXSMM_MELTW_TYPE_UNARY_ELU || t == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV || t == LIBXSMM_MELTW_TYPE_UNARY_ELU_INV) {
      if (!p->op.primary) { set_error(-2, "unary TPP needs alpha in op.primary"); return; }
      a.scalar_f32 = *(const float*)p->op.primary;
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_QUANT && (d.out_type == LIBXSMM_DATATYPE_MXFP4X2 || d.out_type == LIBXSMM_DATATYPE_MXBF8 || d.out_type == LIBXSMM_DATATYPE_NVFP4X2)) {
      // block-scaled output: the scales (E8M0, ldo / 32 per column; NVFP4: E4M3, ldo / 16 per column) go to out.secondary [ref: samples/eltwise/eltwise_unary_quantization_to_mxfp4.c:198-201]
      if (!p->out.secondary) { set_error(-2, "QUANT to a microscaling type needs the scale array in out.secondary"); return; }
      a.aux_out = p->out.secondary;
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_QUANT || t == LIBXSMM_MELTW_TYPE_UNARY_DEQUANT) {
      // the scale is a host scalar behind in.secondary [ref: mateltwise ref :2200, :2333]
      a.scalar_f32 = 1.0f;
      if (!(d.flags & LIBXSMM_MELTW_FLAG_UNARY_NO_SCF_QUANT)) {
        if (!p->in.secondary) { set_error(-2, "QUANT/DEQUANT needs the scale in in.secondary"); return; }
        a.scalar_f32 = *(const float*)p->in.secondary;
      }
      a.aux_in = nullptr;
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR) {
      if (!p->op.primary) { set_error(-2, "REPLICATE_COL_VAR needs the column count in op.primary"); return; }
      a.scalar_u64 = *(const unsigned long long*)p->op.primary;
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_DUMP) {
      if (!p->out.secondary) { set_error(-2, "DUMP needs the second destination in out.secondary"); return; }
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_UNZIP) {
      if (!p->out.secondary) { set_error(-2, "UNZIP needs the byte offset in out.secondary"); return; }
      a.scalar_u64 = *(const unsigned long long*)p->out.secondary;
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X2 || t == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X3) {
      // the byte offsets of the second (and third) piece: host scalars behind out.secondary [ref: mateltwise ref :2439, :2467-2469]
      if (!p->out.secondary) { set_error(-2, "DECOMP_FP32_TO_BF16X2/X3 needs the byte offsets of the pieces in out.secondary"); return; }
      a.scalar_u64 = ((const unsigned long long*)p->out.secondary)[0];
      a.scalar_u64b = t == LIBXSMM_MELTW_TYPE_UNARY_DECOMP_FP32_TO_BF16X3 ? ((const unsigned long long*)p->out.secondary)[1] : 0ull;
      a.aux_out = nullptr;
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_DROPOUT || t == LIBXSMM_MELTW_TYPE_UNARY_DROPOUT_INV) {
      // the probability is a host scalar behind op.primary; DROPOUT advances the generator state (64 dwords) behind op.secondary and
      // writes the mask to out.secondary, DROPOUT_INV reads the mask from in.secondary [ref: mateltwise ref :2091, :2361-2424]
      if (!p->op.primary) { set_error(-2, "DROPOUT needs the probability in op.primary"); return; }
      a.scalar_f32 = *(const float*)p->op.primary;
      const bool bitm = (d.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) != 0;
      if (t == LIBXSMM_MELTW_TYPE_UNARY_DROPOUT) {
        if (!p->op.secondary) { set_error(-2, "DROPOUT needs the generator state in op.secondary"); return; }
        if (bitm && !p->out.secondary) { set_error(-2, "DROPOUT with BITMASK_2BYTEMULT needs the mask destination in out.secondary"); return; }
        a.aux_in = staging_allowed(b.count) ? stage(p->op.secondary, 256, true, true) : p->op.secondary;
        if (!a.aux_in) return;
        a.ws_bytes = 256; a.ws = workspace(a.ws_bytes);      // read-only snapshot of the state for the launch (see dropout_kernel)
        if (!a.ws) return;
      } else if (!p->in.secondary) { set_error(-2, "DROPOUT_INV needs the mask in in.secondary"); return; }
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_ADD || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MAX || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_COLS_IDX_OP_MIN) {
      // the number of listed columns is a host scalar behind in.tertiary, the list itself sits behind in.secondary [ref: mateltwise ref :1121-1138, :1076]
      if (!p->in.tertiary || !p->in.secondary) { set_error(-2, "REDUCE_COLS_IDX needs the column list in in.secondary and its length in in.tertiary"); return; }
      a.scalar_u64 = *(const unsigned long long*)p->in.tertiary;
      const size_t isz = (d.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_4BYTES) ? 4 : 8;
      // A synchronous call may hand plain host memory (the reference's driver does, with n = 0 in the shape: the table's width is only
      // known through the list): the list is then host memory too, its largest entry bounds the table.
      hipPointerAttribute_t attr;
      const bool host_table = staging_allowed(b.count) && (hipPointerGetAttributes(&attr, p->in.primary) != hipSuccess || attr.type == hipMemoryTypeUnregistered);
      (void)hipGetLastError();
      if (host_table) {
        unsigned long long top = 0;
        for (unsigned long long jj = 0; jj < a.scalar_u64; ++jj) top = std::max(top, isz == 4 ? (unsigned long long)((const unsigned int*)p->in.secondary)[jj] : ((const unsigned long long*)p->in.secondary)[jj]);
        a.in0 = (const char*)stage(a.in0, ((size_t)top * (size_t)a.ldi + (size_t)a.m) * (size_t)typesize(a.in0_type), true, false);
        a.out = (char*)stage(a.out, (size_t)a.m * (size_t)typesize(a.out_type), true, true);
        if (!a.in0 || !a.out) return;
      }
      a.aux_in = device_visible(p->in.secondary, (size_t)a.scalar_u64 * isz);
      if (a.scalar_u64 && !a.aux_in) return;
    } else if (t == LIBXSMM_MELTW_TYPE_UNARY_GATHER || t == LIBXSMM_MELTW_TYPE_UNARY_SCATTER) {
      const size_t isz = (d.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_8BYTES) ? 8 : 4;
      const size_t cnt = (d.flags & LIBXSMM_MELTW_FLAG_UNARY_GS_COLS) ? d.n : (d.flags & LIBXSMM_MELTW_FLAG_UNARY_GS_ROWS) ? d.m : (size_t)d.m * d.n;
      if (t == LIBXSMM_MELTW_TYPE_UNARY_GATHER) a.aux_in = device_visible(p->in.secondary, cnt * isz);
      else a.aux_out = const_cast<void*>(device_visible(p->out.secondary, cnt * isz));
    }
  } else if (d.operation == LIBXSMM_MELTW_OPERATION_BINARY) {
    const libxsmm_meltw_binary_param* p = (const libxsmm_meltw_binary_param*)param;
    a.in0 = (const char*)p->in0.primary; a.in1 = (const char*)p->in1.primary; a.out = (char*)p->out.primary;
    a.bs_in0 = b.s[0]; a.bs_in1 = b.s[1]; a.bs_out = b.s[2];
  } else {
    const libxsmm_meltw_ternary_param* p = (const libxsmm_meltw_ternary_param*)param;
    a.in0 = (const char*)p->in0.primary; a.in1 = (const char*)p->in1.primary; a.in2 = (const char*)p->in2.primary; a.out = (char*)p->out.primary;
    a.bs_in0 = b.s[0]; a.bs_in1 = b.s[1]; a.bs_in2 = b.s[2]; a.bs_out = b.s[3];
  }
  if (d.operation == LIBXSMM_MELTW_OPERATION_UNARY && (d.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_RECORD_ARGOP)) {
    // the recorded columns go to out.secondary: one index per row [ref: mateltwise ref :1078-1083]
    if (!a.aux_out) { set_error(-2, "REDUCE_RECORD_ARGOP needs the index destination in out.secondary"); return; }
    if (b.count > 1) { set_error(-2, "REDUCE_RECORD_ARGOP is not available in batched launches"); return; }
    const size_t isz = (d.flags & LIBXSMM_MELTW_FLAG_UNARY_IDX_SIZE_4BYTES) ? 4 : 8;
    if (staging_allowed(b.count)) { a.aux_out = stage(a.aux_out, (size_t)a.m * isz, true, true); if (!a.aux_out) return; }
  }
  // Synchronous single calls of the plain element-wise / reduction TPPs accept host memory too (extents follow from the descriptor);
  // TPPs with index arrays, bit masks as inputs or re-laid-out outputs (gather/scatter, transforms, *_INV, SELECT, ZIP/UNZIP, ...)
  // take device-visible operands only.
  if (staging_allowed(b.count)) {
    const int t = d.param;
    const auto bc = [&](int op) -> int {            // 0 none, 1 row (one value per column), 2 column vector, 3 scalar
      const unsigned int f = d.flags;
      if (d.operation == LIBXSMM_MELTW_OPERATION_UNARY) return op ? 0 : (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_ROW) ? 1 : (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_COL) ? 2 : (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_SCALAR) ? 3 : 0;
      if (d.operation == LIBXSMM_MELTW_OPERATION_BINARY) return (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_ROW_IN_0 << op)) ? 1 : (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0 << op)) ? 2 : (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_SCALAR_IN_0 << op)) ? 3 : 0;
      return (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_ROW_IN_0 << op)) ? 1 : (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_COL_IN_0 << op)) ? 2 : (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_SCALAR_IN_0 << op)) ? 3 : 0;
    };
    const auto extent = [&](int kind, long long ld, int type) -> size_t {
      const long long elems = kind == 1 ? ld * (a.n - 1) + 1 : kind == 2 ? a.m : kind == 3 ? 1 : ld * (a.n - 1) + a.m;
      return (size_t)std::max<long long>(elems, 0) * (size_t)typesize(type);
    };
    bool plain = false, reduce = false;
    if (d.operation == LIBXSMM_MELTW_OPERATION_UNARY) {
      switch (t) {
        case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: case LIBXSMM_MELTW_TYPE_UNARY_XOR: case LIBXSMM_MELTW_TYPE_UNARY_X2: case LIBXSMM_MELTW_TYPE_UNARY_SQRT:
        case LIBXSMM_MELTW_TYPE_UNARY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_TANH: case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID: case LIBXSMM_MELTW_TYPE_UNARY_GELU:
        case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: case LIBXSMM_MELTW_TYPE_UNARY_INC: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT:
        case LIBXSMM_MELTW_TYPE_UNARY_EXP: case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU: case LIBXSMM_MELTW_TYPE_UNARY_ELU: case LIBXSMM_MELTW_TYPE_UNARY_DROPOUT: plain = true; break;
        case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD: case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD: case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD:
        case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX: case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN: case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX:
          reduce = (d.flags & (LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS | LIBXSMM_MELTW_FLAG_UNARY_REDUCE_COLS)) != 0; break;
        default: break;
      }
    } else if (d.operation == LIBXSMM_MELTW_OPERATION_BINARY) {
      plain = t == LIBXSMM_MELTW_TYPE_BINARY_ADD || t == LIBXSMM_MELTW_TYPE_BINARY_MUL || t == LIBXSMM_MELTW_TYPE_BINARY_SUB || t == LIBXSMM_MELTW_TYPE_BINARY_DIV ||
              t == LIBXSMM_MELTW_TYPE_BINARY_MULADD || t == LIBXSMM_MELTW_TYPE_BINARY_MAX || t == LIBXSMM_MELTW_TYPE_BINARY_MIN;
    } else plain = t == LIBXSMM_MELTW_TYPE_TERNARY_MULADD || t == LIBXSMM_MELTW_TYPE_TERNARY_NMULADD;
    if (plain || reduce) {
      a.in0 = (const char*)stage(a.in0, extent(bc(0), a.ldi, a.in0_type), true, false);
      if (a.in1) a.in1 = (const char*)stage(a.in1, extent(bc(1), a.ldi1, a.in1_type), true, false);
      if (a.in2) a.in2 = (const char*)stage(a.in2, extent(bc(2), a.ldi2, a.in2_type), true, false);
      size_t out_bytes = extent(0, a.ldo, a.out_type);
      if (reduce) {
        const bool rows = (d.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) != 0;
        const long long len = rows ? a.n : a.m, apart = rows ? a.n : a.ldo;      // the X2 half of X_X2 starts `apart` elements after X
        out_bytes = (size_t)(t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_X2_OP_ADD ? apart + len : len) * (size_t)typesize(a.out_type);
      }
      a.out = (char*)stage(a.out, out_bytes, true, true);
      if (plain && d.operation == LIBXSMM_MELTW_OPERATION_UNARY && (d.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) && a.aux_out)
        a.aux_out = stage(a.aux_out, (size_t)(((a.ldo + 15) / 16) * 16 / 8) * (size_t)a.n, true, true);
      if (!a.in0 || !a.out) return;
    }
  }
  {  // cache policy of the streaming TPP kernels (as stream_nt does for the GEMMs): the caller's hint, else by the bytes one launch moves against the Infinity Cache
    const int hint = tls().stream_hint;
    const unsigned long long elems = (unsigned long long)std::max(a.m, 0) * (unsigned long long)std::max(a.n, 0) * (unsigned long long)std::max<size_t>(b.count, 1);
    const unsigned long long bytes = elems * (unsigned long long)(typesize(a.in0_type) + typesize(a.out_type));
    a.nt = hint == 2 || (hint == 0 && (bytes >= (256ull << 20) || rt_recent_operands_exceed_cache(a.in0, bytes, a.out))) ? 1 : 0;
  }
  const char* kname = nullptr;
  const bool stoch = a.out_type == LIBXSMM_DATATYPE_BF8 &&
    ((d.operation == LIBXSMM_MELTW_OPERATION_UNARY && (d.flags & LIBXSMM_MELTW_FLAG_UNARY_STOCHASTIC_ROUND)) ||
     (d.operation == LIBXSMM_MELTW_OPERATION_BINARY && (d.flags & LIBXSMM_MELTW_FLAG_BINARY_STOCHASTIC_ROUND)) ||
     (d.operation == LIBXSMM_MELTW_OPERATION_TERNARY && (d.flags & LIBXSMM_MELTW_FLAG_TERNARY_STOCHASTIC_ROUND)));
  if (stoch) {
    // Stochastic rounding needs one draw per element IN ELEMENT ORDER from the caller's generator state (op.secondary, every param struct
    // starts with `op`): the TPP runs into a dense f32 workspace, a second kernel draws and rounds [ref: mateltwise ref :2091, :2485].
    const void* state = ((const libxsmm_matrix_op_arg*)param)->secondary;
    if (!state) { set_error(-2, "a TPP with STOCHASTIC_ROUND needs the generator state in op.secondary"); return; }
    MeltwArgs second = a;
    const size_t tile = (size_t)a.m * (size_t)a.n * sizeof(float);
    const size_t img = ((tile * (size_t)a.nbatch) + 255) & ~(size_t)255;
    float* ws = (float*)workspace(img + 256);             // + a read-only snapshot of the generator state (see launch_stochastic_bf8)
    if (!ws) return;
    a.out = (char*)ws; a.out_type = LIBXSMM_DATATYPE_F32; a.ldo = a.m; a.bs_out = (long long)tile;
    int err = launch_meltw(a, tls().stream, &kname);
    second.in0 = (const char*)ws; second.ws = (char*)ws + img; second.ws_bytes = 256;
    second.aux_in = staging_allowed(b.count) ? stage(state, 256, true, true) : state;
    if (err == 0 && !second.aux_in) return;
    if (err == 0) err = launch_stochastic_bf8(second, tls().stream);
    if (kname) { if (b.count > 1) k->kname_batched = kname; else k->kname_single = kname; }
    finish_launch(err, kname);
    return;
  }
  const int err = launch_meltw(a, tls().stream, &kname);
  if (kname) { if (b.count > 1) k->kname_batched = kname; else k->kname_single = kname; }   // the device kernel that actually ran
  finish_launch(err, kname);
}

// geometry of the generic  Y[r][q] (+)= sum val * X[idx][q]  form for a sparse context (shared by run_spmm and the JIT)
static void spmm_geometry(const KernelCtx* k, SpmmArgs& a) {
  const libxsmm_gemm_descriptor& d = k->g;
  const long long P = k->packed_width;
  if (k->kind == K_SPMM_ASPARSE) {
    a.ld_x = (long long)d.ldb * P; a.ld_y = (long long)d.ldc * P; a.ncols = (long long)k->sp_ncols * P;
    a.outer_x = a.outer_y = 0; a.nouter = 1; a.skip_empty = k->sp_skip_empty;
  } else {   // B sparse: one slab per row m of the packed A/C
    a.ld_x = P; a.ld_y = P; a.ncols = P; a.outer_x = (long long)d.lda * P; a.outer_y = (long long)d.ldc * P;
    a.nouter = (int)d.m; a.skip_empty = 0;
  }
  a.dtype = d.a_type; a.beta0 = (d.flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 1 : 0;
  a.rows = k->sp_rows; a.inner = k->sp_inner; a.nnz = k->sp_nnz;
}

void run_spmm(KernelCtx* k, const void* param, const BatchSpec& b) {
  coalesce_flush();
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  SpmmArgs a{};
  // the pattern arrays (and the JIT module) live on the device the kernel was created on
  if (k->device != cur_device()) { set_error(-3, "packed sparse kernel was created on device %d but is called on device %d", k->device, cur_device()); return; }
  spmm_geometry(k, a);
  scratch_reset();
  a.ptr = k->d_ptr; a.idx = k->d_idx; a.vmap = k->d_vmap;
  const bool asp = (k->kind == K_SPMM_ASPARSE);
  const void* vals = asp ? (k->d_vals ? k->d_vals : p->a.primary) : p->b.primary;     // baked (areg / FsSpMDM) or run-time values
  const void* x = asp ? p->b.primary : p->a.primary;
  void* y = p->c.primary;
  if (!vals || !x || !y) { set_error(-2, "sparse kernel called with a NULL operand"); return; }
  {   // synchronous single calls accept plain host memory (e.g. values straight out of an .mtx reader's malloc)
    const size_t es = (a.dtype == LIBXSMM_DATATYPE_F64) ? 8 : 4;
    if (!k->d_vals || !asp) vals = host_input(vals, es * (size_t)a.nnz, b.count);
    if (staging_allowed(b.count)) {
      // the dense operand and C may be PANELS of wider host matrices (PyFR hands column blocks of B and C with ldb = ldc = the full width
      // [ref: samples/xgemm_sparse_Ainregs/pyfr_driver_asp_reg.c:379-393]): only the touched bytes of every row / slab move, never the gaps
      if (asp) {
        x = stage2d(x, es * (size_t)a.ncols, (size_t)a.inner, es * (size_t)a.ld_x, true, false);
        y = stage2d(y, es * (size_t)a.ncols, (size_t)a.rows, es * (size_t)a.ld_y, true, true);
      } else {
        x = stage2d(x, es * (size_t)a.inner * (size_t)a.ld_x, (size_t)a.nouter, es * (size_t)a.outer_x, true, false);
        y = stage2d(y, es * (size_t)a.rows * (size_t)a.ld_y, (size_t)a.nouter, es * (size_t)a.outer_y, true, true);
      }
    }
    if (!vals || !x || !y) return;
  }
  a.vals = vals; a.x = (const char*)x; a.y = (char*)y;
  // batched launch = the caller's loop over elements: the dense operand and C step by byte strides, the sparse operand's values are
  // shared (its stride must be 0) [include/libxsmm_hip.h]
  const long long esz = (a.dtype == LIBXSMM_DATATYPE_F64) ? 8 : 4;
  const long long sx = asp ? b.s[1] : b.s[0], sy = b.s[2], sv = asp ? b.s[0] : b.s[1];
  const long long count = (long long)b.count;
  if (count > 1 && (sv != 0 || sx % esz != 0 || sy % esz != 0)) { set_error(-3, "batched packed kernel: the value stride must be 0 and operand strides multiples of the element size"); return; }
  const char* kname = nullptr;
  int err = 0;
  const long long bslabs = a.nouter;
  if (count > 1 && !k->jit && !k->h_ptr.empty()) {       // specialisation deferred at creation (one call too small to repay hiprtc)
    static std::mutex mu; std::lock_guard<std::mutex> lk(mu);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;     // loading a module is not a capturable operation: wait for an eager launch
    if (tls().stream) (void)hipStreamIsCapturing((hipStream_t)tls().stream, &cs);
    if (cs == hipStreamCaptureStatusNone && !k->jit && !k->h_ptr.empty() && a.ncols * a.nouter * count >= 4096) {
      ::attach_jit(k, k->h_ptr.data(), k->h_idx.data(), k->h_vmap.empty() ? nullptr : k->h_vmap.data(), count);
      k->h_ptr.clear(); k->h_ptr.shrink_to_fit();        // one attempt
    }
  }
  if (jit_spmm_usable(k->jit, a.x, a.y) && (count == 1 || jit_spmm_usable(k->jit, a.x + sx, a.y + sy))) {   // base and stride aligned to the kernel's vector width
    kname = jit_name(k->jit);
    err = jit_spmm_launch_slabs(k->jit, a.vals, a.x, a.y, count, bslabs, a.outer_x, a.outer_y, sx / esz, sy / esz, tls().stream);
  } else if (count == 1) err = launch_spmm(a, tls().stream, &kname);
  else if (asp) {              // precompiled kernels: the slab axis of the A-sparse form is free -> one launch
    a.nouter = (int)count; a.outer_x = sx / esz; a.outer_y = sy / esz;
    err = launch_spmm(a, tls().stream, &kname);
  } else {                     // B-sparse already uses the slab axis for the rows of A: one launch per element
    for (long long e = 0; e < count && err == 0; ++e) {
      SpmmArgs ae = a; ae.x = a.x + e * sx; ae.y = a.y + e * sy;
      err = launch_spmm(ae, tls().stream, &kname);
    }
  }
  if (kname) k->kname_single = k->kname_batched = kname;
  finish_launch(err, kname);
}

static void pgemm_geometry(const KernelCtx* k, PgemmArgs& a) {
  const libxsmm_gemm_descriptor& d = k->g;
  a.M = (int)d.m; a.N = (int)d.n; a.K = (int)d.k; a.lda = (int)d.lda; a.ldb = (int)d.ldb; a.ldc = (int)d.ldc;
  a.dtype = d.a_type; a.beta0 = (d.flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 1 : 0; a.P = k->packed_width;
}
void run_pgemm(KernelCtx* k, const void* param) {
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  PgemmArgs a{};
  pgemm_geometry(k, a);
  a.a = (const char*)p->a.primary; a.b = (const char*)p->b.primary; a.c = (char*)p->c.primary;
  if (!a.a || !a.b || !a.c) { set_error(-2, "packed GEMM called with a NULL operand"); return; }
  const char* kname = nullptr;
  int err;
  if (jit_pgemm_usable(k->jit, a.a, a.b, a.c)) { kname = jit_name(k->jit); err = jit_spmm_launch(k->jit, a.a, a.b, a.c, tls().stream); }
  else err = launch_pgemm(a, tls().stream, &kname);
  if (kname) k->kname_single = k->kname_batched = kname;
  finish_launch(err, kname);
}

void run_bcsc(KernelCtx* k, const void* param) {
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  const libxsmm_gemm_descriptor& d = k->g;
  BcscArgs a{};
  scratch_reset();
  if (!p->b.quaternary) { set_error(-2, "BCSC kernel needs the block-column count in b.quaternary"); return; }
  const unsigned long long nblk_n = *(const unsigned long long*)p->b.quaternary;   // [ref: spmm_kernel.c:451-456]
  a.M = k->packed_width; a.N = (int)d.ldc; a.K = (int)d.k; a.m_blocks = (int)d.m; a.bk = k->bk; a.bn = k->bn; a.nblk_n = (int)nblk_n;
  a.a_type = d.a_type; a.b_type = d.b_type; a.c_type = d.c_type; a.vnni_a = (d.flags & LIBXSMM_GEMM_FLAG_VNNI_A) ? 1 : 0; a.beta0 = (d.flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 1 : 0;
  a.stream_hint = tls().stream_hint;
  a.a = (const char*)p->a.primary; a.bvals = (const char*)p->b.primary; a.c = (char*)p->c.primary;
  if (!p->b.secondary || !p->b.tertiary) { set_error(-2, "BCSC kernel needs colptr in b.secondary and rowidx in b.tertiary"); return; }
  if (k->device != cur_device() && !k->bcsc_cache.empty()) { set_error(-3, "BCSC kernel holds a cached pattern on device %d but is called on device %d", k->device, cur_device()); return; }
  const int nkb = (a.bk > 0 && a.K % a.bk == 0) ? a.K / a.bk : 0;
  hipPointerAttribute_t attr;
  const auto on_host = [&](const void* q) {
    const hipError_t e = hipPointerGetAttributes(&attr, q);
    const bool host = (e != hipSuccess) || attr.type == hipMemoryTypeUnregistered || attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return host;
  };
  if (nkb > 0 && on_host(p->b.secondary) && on_host(p->b.tertiary)) {
    // The pattern arrived in HOST memory (the reference's calling convention: plain malloc'd colptr / rowidx [ref: spmm_kernel.c:306-347]).
    // It is readable here, so it is inverted on the host -- table[n-block][k-block] = block id -- and kept on the device with its key:
    // a pattern rarely changes between calls, and a call with a known pattern then costs no staging copy and no inversion kernel.
    const unsigned int* hc = (const unsigned int*)p->b.secondary; const unsigned int* hr = (const unsigned int*)p->b.tertiary;
    const unsigned int nnzb = hc[nblk_n];
    const size_t n_ptr = (size_t)nblk_n + 1, n_idx = std::max<size_t>(1, nnzb), n_tab = std::max<size_t>(1, (size_t)nblk_n * nkb);
    // key = [nblk_n, nkb, colptr..., rowidx...]; compared in place against the caller's arrays: the hit path neither allocates nor locks
    const auto matches = [&](const KernelCtx::BcscCached* e) {
      return e && e->pattern.size() == 2 + n_ptr + nnzb && e->pattern[0] == (unsigned int)nblk_n && e->pattern[1] == (unsigned int)nkb &&
             std::memcmp(e->pattern.data() + 2, hc, n_ptr * sizeof(unsigned int)) == 0 &&
             (nnzb == 0 || std::memcmp(e->pattern.data() + 2 + n_ptr, hr, (size_t)nnzb * sizeof(unsigned int)) == 0);
    };
    const KernelCtx::BcscCached* hit = k->bcsc_last.load(std::memory_order_acquire);
    if (!matches(hit)) {
      static std::mutex cache_lock;
      std::lock_guard<std::mutex> guard(cache_lock);
      hit = nullptr;
      for (const auto* e : k->bcsc_cache) if (matches(e)) { hit = e; break; }
      if (!hit) {
        std::vector<unsigned int> img(n_ptr + n_idx + n_tab, 0xffffffffu);
        std::copy(hc, hc + n_ptr, img.begin()); std::copy(hr, hr + nnzb, img.begin() + n_ptr);
        for (unsigned long long nb = 0; nb < nblk_n; ++nb) {
          if (hc[nb] > hc[nb + 1] || hc[nb + 1] > nnzb) { set_error(-2, "BCSC colptr is not monotone"); return; }
          for (unsigned int b = hc[nb]; b < hc[nb + 1]; ++b) {
            if (hr[b] >= (unsigned int)nkb) { set_error(-2, "BCSC rowidx[%u] = %u is outside the %d k-blocks", b, hr[b], nkb); return; }
            img[n_ptr + n_idx + nb * nkb + hr[b]] = b;
          }
        }
        unsigned int* blockp = nullptr;
        if (!hip_ok(hipMalloc((void**)&blockp, img.size() * sizeof(unsigned int)), "hipMalloc(BCSC pattern)")) return;
        if (!hip_ok(hipMemcpy(blockp, img.data(), img.size() * sizeof(unsigned int), hipMemcpyHostToDevice), "hipMemcpy(BCSC pattern)")) return;
        KernelCtx::BcscCached* fresh = new KernelCtx::BcscCached();
        fresh->pattern.reserve(2 + n_ptr + nnzb);
        fresh->pattern.push_back((unsigned int)nblk_n); fresh->pattern.push_back((unsigned int)nkb);
        fresh->pattern.insert(fresh->pattern.end(), hc, hc + n_ptr); fresh->pattern.insert(fresh->pattern.end(), hr, hr + nnzb);
        fresh->d_block = blockp;
        if (nkb <= 64 && a.bn > 0)        // the k-blocks the first 64 columns use (see BcscArgs::kmask0)
          for (unsigned long long nb = 0; nb < nblk_n && nb * (unsigned long long)a.bn < 64ull; ++nb)
            for (unsigned int b = hc[nb]; b < hc[nb + 1]; ++b) fresh->kmask0 |= 1ull << hr[b];
        // an evicted entry stays alive until the kernel is released: another thread may be past its lock-free hit, a launch may still read its table
        if (k->bcsc_cache.size() >= 4) { k->bcsc_old.push_back(k->bcsc_cache.front()); k->bcsc_cache.erase(k->bcsc_cache.begin()); }
        // ... but a caller that cycles through many patterns must not grow device memory without bound (advisor, round 3): beyond 64 retired entries that still own a device table the
        // tables of the oldest 32 are freed after the device has drained (no launch can still read them; a thread would have to sit between its lock-free hit and
        // its launch across 32 pattern insertions for the pointer to matter).  The small host parts stay until the kernel is released.
        size_t live_old = 0;
        for (auto* e : k->bcsc_old) live_old += e->d_block ? 1 : 0;
        if (live_old > 64) {
          (void)hipDeviceSynchronize();
          size_t freed = 0;
          for (auto* e : k->bcsc_old) { if (freed == 32) break; if (e->d_block) { (void)hipFree(e->d_block); e->d_block = nullptr; ++freed; } }
        }
        k->bcsc_cache.push_back(fresh);
        k->device = cur_device();
        hit = fresh;
      }
      k->bcsc_last.store(hit, std::memory_order_release);
    }
    unsigned int* blockp = hit->d_block;
    a.colptr = blockp; a.rowidx = blockp + n_ptr; a.table = blockp + n_ptr + n_idx; a.table_ready = 1; a.nnzb = (int)nnzb; a.kmask0 = hit->kmask0;
  } else {
  a.colptr = (const unsigned int*)device_visible(p->b.secondary, (size_t)(nblk_n + 1) * sizeof(unsigned int));
  if (!a.colptr) { set_error(-2, "BCSC kernel needs colptr in b.secondary"); return; }
  // rowidx: a device array is used in place (no size needed, no host round trip -> capturable in a hipGraph); a host
  // array is staged, its length colptr[nblk_n] being host-readable in that case
  {
    const bool idx_on_device = (hipPointerGetAttributes(&attr, p->b.tertiary) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged));
    (void)hipGetLastError();
    if (idx_on_device) a.rowidx = (const unsigned int*)p->b.tertiary;
    else {
      const bool ptr_on_device = (hipPointerGetAttributes(&attr, p->b.secondary) == hipSuccess && attr.type == hipMemoryTypeDevice);
      (void)hipGetLastError();
      unsigned int nnzb = 0;
      if (!ptr_on_device) nnzb = ((const unsigned int*)p->b.secondary)[nblk_n];
      else { (void)hipMemcpyAsync(&nnzb, (const unsigned int*)p->b.secondary + nblk_n, sizeof(nnzb), hipMemcpyDeviceToHost, cur_stream()); (void)hipStreamSynchronize(cur_stream()); }
      a.rowidx = (const unsigned int*)device_visible(p->b.tertiary, (size_t)std::max(1u, nnzb) * sizeof(unsigned int));
    }
  }
  const KernelCtx::BcscBound& bd = k->bcsc_bound;
  if (nkb > 0 && bd.d_table && bd.colptr == p->b.secondary && bd.rowidx == p->b.tertiary && bd.nblk_n == nblk_n && bd.nkb == nkb) {
    a.table = bd.d_table; a.table_ready = 1;        // libxsmm_hip_bcsc_bind_pattern: inverted once, no inversion kernel per call
    a.nnzb = bd.nnzb; a.kmask0 = bd.kmask0;         // ... and read once: the streaming kernels that keep B in LDS apply as for a host-resident pattern
  } else if (nkb > 0) a.table = workspace((size_t)std::max(1, a.nblk_n) * (size_t)nkb * sizeof(unsigned int));
  }
  if (!a.a || !a.bvals || !a.c || !a.rowidx) { set_error(-2, "BCSC kernel called with a NULL operand"); return; }
  const char* kname = nullptr;
  const int err = launch_bcsc(a, tls().stream, &kname);
  if (kname) k->kname_single = k->kname_batched = kname;
  finish_launch(err, kname);
}

void run_csparse(KernelCtx* k, const void* param) {
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  const libxsmm_gemm_descriptor& d = k->g;
  if (k->device != cur_device()) { set_error(-3, "packed sparse kernel was created on device %d but is called on device %d", k->device, cur_device()); return; }
  scratch_reset();
  CsparseArgs a{};
  a.rows = k->d_idx; a.cols = k->d_vmap; a.nnz = k->sp_nnz; a.K = (int)d.k; a.lda = (int)d.lda; a.ldb = (int)d.ldb; a.P = k->packed_width;
  a.beta0 = (d.flags & LIBXSMM_GEMM_FLAG_BETA_0) ? 1 : 0;
  // a synchronous call may be handed host memory like the other packed kernels: extents follow from the descriptor
  a.a = (const char*)host_input(p->a.primary, (size_t)d.k * d.lda * (size_t)a.P * 4, 1);
  a.b = (const char*)host_input(p->b.primary, (size_t)d.k * d.ldb * (size_t)a.P * 4, 1);
  a.c = (char*)host_inout(p->c.primary, (size_t)std::max(1u, a.nnz) * 4, 1);
  if (!a.a || !a.b || !a.c) { set_error(-2, "packed C-sparse kernel called with a NULL operand"); return; }
  const char* kname = nullptr;
  const int err = launch_csparse(a, tls().stream, &kname);
  finish_launch(err, kname);
}

// ---- coalescing launch mode (libxsmm_hip_set_async(2) / LIBXSMM_HIP_COALESCE=1) ---------------------------------------------------------------
// The reference executes ONE small GEMM per call and leaves the batch loop to the caller [ref: documentation/libxsmm_mm.md:95-107]; on a GPU that
// is one launch per 65 kflop.  In this mode consecutive calls through ONE plain (BR)GEMM handle are only QUEUED -- three pointers per call -- and
// leave as ONE pointer-list batch launch when something else happens: a call through another handle or of another kind, a different batch-reduce
// count, libxsmm_hip_sync / _set_stream / _set_async / finalize, a full queue, or a call that touches what a queued call writes (or writes what a
// queued call reads): the queue keeps the byte ranges of its operands, so a caller's dependent sequence (C of call i read by call i + 1, two
// calls accumulating into one C) keeps its order -- the batched launch only ever holds mutually independent calls.  Results are valid after
// libxsmm_hip_sync(), as in every stream-ordered mode.  Queued: NONE / STRIDE batch-reduce, no fused operator, no per-call scale operands.
struct CoalesceQueue {
  KernelCtx* k = nullptr;
  unsigned long long br_count = 0;
  std::vector<const void*> a, b; std::vector<void*> c;
  size_t ea = 0, eb = 0, ec = 0;                        // bytes a call reads through a / b and writes through c
  uintptr_t cmin = 0, cmax = 0, rmin = 0, rmax = 0;     // hulls of the queued C ranges and of the queued A / B ranges
  bool c_monotonic = true;                              // every queued C started at or behind the end of the hull so far: no two overlap
  bool strided = true; long long sa = 0, sb = 0, sc = 0;  // the calls so far step by constant byte strides (the usual loop): they leave as a STRIDED batch
  uintptr_t low_bits = 0;                               // OR of all queued pointers: alignment of the lists
  std::unordered_map<uintptr_t, uintptr_t> index;       // queued C blocks by address / ec (built lazily, see coalesce_try)
  bool indexed = false;
  unsigned int gen = 0;                                 // registry generation the queued handle belongs to (libxsmm_finalize on ANOTHER thread frees it)
  // a thread that ends with calls still queued never launched them: results are defined after libxsmm_hip_sync() only (include/libxsmm_hip.h) -- say so instead of
  // dropping them silently (no launch from a thread_local destructor: the thread's stream state may be gone already)
  ~CoalesceQueue() { if (!a.empty() && libxsmm_verbosity != 0) std::fprintf(stderr, "LIBXSMM-AMD: a thread ended with %zu coalesced calls still queued (no libxsmm_hip_sync()): they were never launched\n", a.size()); }
};
thread_local CoalesceQueue t_queue;
static const size_t kCoalesceCap = 65536;

void coalesce_flush() {
  CoalesceQueue& q = t_queue;
  if (q.a.empty()) return;
  std::vector<const void*> la, lb; std::vector<void*> lc;
  la.swap(q.a); lb.swap(q.b); lc.swap(q.c);             // the queue is empty before anything is launched: run_gemm's own flush hook finds nothing
  KernelCtx* k = q.k; q.k = nullptr;
  if (q.gen != g_generation.load(std::memory_order_acquire)) {        // the registry (and this handle) was freed by libxsmm_finalize on another thread: nothing to launch through
    la.clear(); lb.clear(); lc.clear(); q.a.swap(la); q.b.swap(lb); q.c.swap(lc);
    set_error(-3, "coalesced calls dropped: libxsmm_finalize() ran on another thread while they were queued (libxsmm_hip_sync() first)");
    return;
  }
  libxsmm_gemm_param p; std::memset(&p, 0, sizeof(p));
  unsigned long long brc = q.br_count;
  p.op.tertiary = &brc; p.a.primary = const_cast<void*>(la[0]); p.b.primary = const_cast<void*>(lb[0]); p.c.primary = lc[0];
  BatchSpec bs; bs.count = la.size();
  if (q.strided && la.size() >= 2) { bs.s[0] = q.sa; bs.s[1] = q.sb; bs.s[2] = q.sc; }       // exactly libxsmm_hip_gemm_batch_strided
  else if (la.size() >= 2) { bs.la = la.data(); bs.lb = lb.data(); bs.lc = lc.data(); bs.lists_on_host = true; bs.lists_aligned16 = (q.low_bits & 15u) == 0; }
  run_gemm(k, &p, bs);
  la.clear(); lb.clear(); lc.clear();
  q.a.swap(la); q.b.swap(lb); q.c.swap(lc);             // keep the capacity
}
static bool ranges_overlap(uintptr_t a0, size_t an, uintptr_t b0, size_t bn) { return a0 < b0 + bn && b0 < a0 + an; }
// true: the call has been queued
bool coalesce_try(KernelCtx* k, const void* param) {
  if (k->kind != K_GEMM || t_nest > 0 || tls().pipe_lanes > 1 || !param) return false;
  {  // a stream that is being CAPTURED records addresses, not contents: a queued call's pointer list would be re-read from whatever the staging slot holds at
     // replay time.  While capturing, calls launch one by one (mode 1 semantics); what was queued before the capture began has left already (flushed below).
    // Never asked of the NULL stream: it cannot be captured, and the query would invalidate another stream's global-mode capture [advisor, round 5].
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (cur_stream() != nullptr) {
      if (hipStreamIsCapturing(cur_stream(), &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    }
    if (cs != hipStreamCaptureStatusNone) return false;
  }
  const libxsmm_gemm_descriptor& d = k->g;
  const unsigned int never = LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI | LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS | LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET |
    LIBXSMM_GEMM_FLAG_INTLV_A_FORMAT | LIBXSMM_GEMM_FLAG_USE_COL_VEC_SCF | LIBXSMM_GEMM_FLAG_USE_COL_VEC_ZPT | LIBXSMM_GEMM_FLAG_USE_MxK_ZPT | LIBXSMM_GEMM_FLAG_USE_MxK_SCF;
  if (d.flags & never) return false;
  const auto plain = [](int t) { return t == LIBXSMM_DATATYPE_F32 || t == LIBXSMM_DATATYPE_F64 || t == LIBXSMM_DATATYPE_BF16 || t == LIBXSMM_DATATYPE_F16; };
  if (!plain(d.a_type) || !plain(d.b_type) || !plain(d.c_type)) return false;          // (8-bit and MX types carry per-call scale operands)
  const libxsmm_gemm_param* p = (const libxsmm_gemm_param*)param;
  if (!p->a.primary || !p->b.primary || !p->c.primary) return false;
  unsigned long long brc = 1;
  const bool strided = (d.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE) != 0;
  if (strided) { if (!p->op.tertiary) return false; brc = *(const unsigned long long*)p->op.tertiary; if (brc == 0 || d.br_stride_a < 0 || d.br_stride_b < 0) return false; }
  CoalesceQueue& q = t_queue;
  if (!q.a.empty() && (q.k != k || q.br_count != brc || q.a.size() >= kCoalesceCap)) coalesce_flush();
  const bool ta = (d.flags & LIBXSMM_GEMM_FLAG_TRANS_A) != 0, tb = (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B) != 0;
  const size_t ea = (size_t)(brc - 1) * (size_t)(strided ? d.br_stride_a : 0) + (size_t)d.lda * (size_t)(ta ? d.m : d.k) * (size_t)typesize(d.a_type);
  const size_t eb = (size_t)(brc - 1) * (size_t)(strided ? d.br_stride_b : 0) + (size_t)d.ldb * (size_t)(tb ? d.k : d.n) * (size_t)typesize(d.b_type);
  const size_t ec = (size_t)d.ldc * (size_t)(d.n + ((d.flags & LIBXSMM_GEMM_FLAG_VNNI_C) ? (d.n & 1) : 0)) * (size_t)typesize(d.c_type);
  const uintptr_t pa = (uintptr_t)p->a.primary, pb = (uintptr_t)p->b.primary, pc = (uintptr_t)p->c.primary;
  if (!q.a.empty()) {
    // read-after-write: this call's A / B against the queued C ranges; write-after-write / write-after-read: its C against the queued C and A / B ranges
    bool hazard = false;
    const bool a_near = ranges_overlap(pa, ea, q.cmin, q.cmax - q.cmin), b_near = ranges_overlap(pb, eb, q.cmin, q.cmax - q.cmin);
    const bool c_near_c = !(q.c_monotonic && pc >= q.cmax) && ranges_overlap(pc, ec, q.cmin, q.cmax - q.cmin);
    const bool c_near_r = ranges_overlap(pc, ec, q.rmin, q.rmax - q.rmin);
    // Exact checks without an O(n) scan per call [advisor, round 4: interleaved layouts ({A_i, B_i, C_i} structs, operands carved alternately from one arena)
    // are "near" on every call].  Queued C blocks never overlap each other (a write-after-write flushes first) and all have the size ec, so a table keyed by
    // address / ec holds at most one block start per key: a range query probes (length / ec + 2) keys.  The table is built on the first near call only --
    // the usual loop over disjoint arrays never pays for it.
    if (a_near || b_near || c_near_c) {
      if (!q.indexed) { q.index.clear(); q.index.reserve(q.c.size() * 2 + 64); for (void* c : q.c) q.index.emplace((uintptr_t)c / q.ec, (uintptr_t)c); q.indexed = true; }
      const auto hits = [&q](uintptr_t x, size_t len) {
        const uintptr_t k0 = x / q.ec, k1 = (x + len - 1) / q.ec;
        if (k1 - k0 > 64) return true;                      // a very long read range (a deep batch-reduce chain): not worth probing, order it behind the queue
        for (uintptr_t key = (k0 > 0 ? k0 - 1 : 0); key <= k1; ++key) {
          const auto it = q.index.find(key);
          if (it != q.index.end() && ranges_overlap(x, len, it->second, q.ec)) return true;
        }
        return false;
      };
      hazard = (a_near && hits(pa, ea)) || (b_near && hits(pb, eb)) || (c_near_c && hits(pc, ec));
    }
    // write-after-read (this call's C against what the queued calls READ: two sizes, no common grid): the exact scan is bounded to a few hundred entries,
    // beyond that a near call simply flushes -- a batch of 256+ problems already amortises its launch
    if (c_near_r && !hazard) {
      if (q.a.size() > 256) hazard = true;
      for (size_t i = 0; i < q.a.size() && !hazard; ++i)
        hazard = ranges_overlap(pc, ec, (uintptr_t)q.a[i], q.ea) || ranges_overlap(pc, ec, (uintptr_t)q.b[i], q.eb);
    }
    if (hazard) coalesce_flush();
  }
  if (q.a.empty()) {
    q.k = k; q.gen = g_generation.load(std::memory_order_acquire); q.br_count = brc; q.ea = ea; q.eb = eb; q.ec = ec;
    q.cmin = pc; q.cmax = pc + ec; q.rmin = std::min(pa, pb); q.rmax = std::max(pa + ea, pb + eb); q.c_monotonic = true;
    q.strided = true; q.sa = q.sb = q.sc = 0; q.low_bits = 0; q.indexed = false;
  } else {
    const size_t n = q.a.size();
    if (n == 1) { q.sa = (long long)(pa - (uintptr_t)q.a[0]); q.sb = (long long)(pb - (uintptr_t)q.b[0]); q.sc = (long long)(pc - (uintptr_t)q.c[0]); }
    else if (q.strided) q.strided = (long long)(pa - (uintptr_t)q.a[n - 1]) == q.sa && (long long)(pb - (uintptr_t)q.b[n - 1]) == q.sb && (long long)(pc - (uintptr_t)q.c[n - 1]) == q.sc;
    if (pc < q.cmax) q.c_monotonic = false;
    q.cmin = std::min(q.cmin, pc); q.cmax = std::max(q.cmax, pc + ec);
    q.rmin = std::min(q.rmin, std::min(pa, pb)); q.rmax = std::max(q.rmax, std::max(pa + ea, pb + eb));
  }
  q.low_bits |= pa | pb | pc;
  if (q.indexed) q.index.emplace(pc / q.ec, pc);
  q.a.push_back(p->a.primary); q.b.push_back(p->b.primary); q.c.push_back(p->c.primary);
  return true;
}

void run_any(KernelCtx* k, const void* param, const BatchSpec& b) {
  coalesce_flush();                   // whatever launches next is ordered behind the queued calls
  if (!param && k->kind != K_TILECFG) { set_error(-2, "kernel called with a NULL parameter struct"); return; }
  if (g_device_count <= 0 && k->kind != K_TILECFG) { set_error(-4, "no HIP device: kernel not launched (this backend has no CPU path)"); return; }
  switch (k->kind) {
    case K_GEMM: run_gemm(k, param, b); break;
    case K_MELTW: run_meltw(k, param, b); break;
    case K_SPMM_ASPARSE: case K_SPMM_BSPARSE: run_spmm(k, param, b); break;
    case K_BCSC: run_bcsc(k, param); break;
    case K_SPMM_CSPARSE: run_csparse(k, param); break;
    case K_PGEMM: run_pgemm(k, param); break;
    case K_MEQN: scratch_reset(); run_meqn(k->eqn, param); break;
    case K_TILECFG: break;   // AMX tile configuration has no meaning here [ref: gemm ref :2821-2826]
  }
}

}  // namespace

namespace xamd {
const void* rt_new_meqn_handle(EqnPlan* plan) {
  std::lock_guard<std::mutex> guard(g_lock);
  KernelCtx* c = new_ctx_locked(K_MEQN);
  if (!c) return nullptr;
  c->registered = true; c->eqn = plan; c->kname_single = c->kname_batched = meqn_plan_name(plan);
  g_meqn_ctxs.push_back(c);            // registry-owned like the reference's equation kernels: released by libxsmm_finalize
  return handle_for_slot(c->slot);
}
void rt_finish_launch(int err, const char* kernel_name) { finish_launch(err, kernel_name); }
void* rt_workspace(size_t nbytes) { return workspace(nbytes); }
void rt_workspace_reserve(size_t nbytes) { t_ws_reserved = (nbytes + 255) & ~(size_t)255; }
bool rt_ready() { return runtime_ready(); }
const void* rt_small_host_input(const void* p, size_t nbytes) { return device_visible(p, nbytes); }
void* rt_small_host_output(void* p, size_t nbytes) { return host_inout(p, nbytes, 1); }
void rt_scratch_reset() { scratch_reset(); }
void rt_nest(int delta) { t_nest += delta; }
void rt_note(const char* what, int a, int b, int c) { vlog(1, "%s (%d, %d, %d)", what, a, b, c); }
void* rt_stream() { return tls().stream; }
int rt_jit_mode() { return jit_mode(); }
bool rt_dryrun() { return g_dryrun; }

void invoke(int slot, const void* param) {
  KernelCtx* k = g_slots[slot];
  if (!k) { set_error(-3, "call through a released kernel handle"); return; }
  if (tls().async == 2 && g_device_count > 0 && coalesce_try(k, param)) return;
  run_any(k, param, BatchSpec{});
}
}  // namespace xamd

// =====================================================================================================
// C API
// =====================================================================================================
extern "C" {

LIBXSMM_API void libxsmm_init(void) {
  std::lock_guard<std::mutex> guard(g_lock);
  if (libxsmm_ninit >= 2) return;
  libxsmm_ninit = 1;
  const char* v = std::getenv("LIBXSMM_VERBOSE");
  if (v) libxsmm_verbosity = std::atoi(v);
  thunk_pool_init_locked();
  if (const char* cap = std::getenv("LIBXSMM_HIP_MAX_HANDLES")) { const int n = std::atoi(cap); if (n > 0) { g_slot_limit = std::min(n, kSlots); g_registered_limit = std::min(g_registered_limit, g_slot_limit); } }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { n = 0; (void)hipGetLastError(); }
  g_device_count = n;
  { const char* dry = std::getenv("LIBXSMM_HIP_DRYRUN"); g_dryrun = (n <= 0 && dry && dry[0] == '1'); }
  libxsmm_target_archid = LIBXSMM_X86_GENERIC;   // what the reference reports for LIBXSMM_TARGET=generic: below SPR (callers must not hoist AMX tile
                                                 // config) and the id its drivers test for when they pick plain-C gold code
  vlog(1, "initialised: %d HIP device(s), target gfx950, %s handles", n, g_thunk_pool ? "thunk-pool" : "static-trampoline");
  libxsmm_ninit = 2;
}

// Releases everything the registry owns (dispatched kernels, equation kernels, retired staging blocks) and returns the library to its
// un-initialised state: a later dispatch re-initialises, as the reference's init/finalize cycles do [ref: src/libxsmm_main.c:1503-1640].
// Caller-owned kernels (create_*) stay valid until libxsmm_release_kernel.  Every thread's dispatch cache is invalidated by the generation.
LIBXSMM_API void libxsmm_finalize(void) {
  coalesce_flush();                        // the calling thread's queued calls (other threads drain theirs at their own libxsmm_hip_sync)
  std::lock_guard<std::mutex> guard(g_lock);
  if (libxsmm_ninit < 2) return;
  if (g_device_count > 0) (void)hipDeviceSynchronize();
  if (libxsmm_verbosity != 0) std::fprintf(stderr, "LIBXSMM-AMD: registry holds %zu kernels at exit\n", g_registry.size() + g_meqn_ctxs.size());
  for (auto& kv : g_registry) free_ctx_locked(kv.second);
  g_registry.clear();
  for (KernelCtx* c : g_meqn_ctxs) free_ctx_locked(c);
  g_meqn_ctxs.clear();
  free_meqn_equations();
  { Retired& r = retired(); std::lock_guard<std::mutex> g2(r.lock); for (void* p : r.blocks) (void)hipFree(p); r.blocks.clear(); }
  g_generation.fetch_add(1, std::memory_order_release);
  libxsmm_ninit = 0;
}

LIBXSMM_API int libxsmm_get_target_archid(void) { return libxsmm_target_archid; }
// There is one code path (gfx950); the id callers can observe stays at or below X86_GENERIC so that they never hoist AMX tile
// configuration [ref: samples/xgemm/gemm_kernel.c:3813-3824].  Requests are clamped like the reference clamps to the CPUID level
// [ref: src/libxsmm_main.c:1770-1800] and a request for something else is said out loud at LIBXSMM_VERBOSE >= 1.
LIBXSMM_API void libxsmm_set_target_archid(int id) {
  if (id > LIBXSMM_X86_GENERIC) { vlog(1, "libxsmm_set_target_archid(%d): this backend only runs gfx950 kernels; id stays %d (generic)", id, (int)LIBXSMM_X86_GENERIC); id = LIBXSMM_X86_GENERIC; }
  if (id < LIBXSMM_TARGET_ARCH_GENERIC) id = LIBXSMM_TARGET_ARCH_GENERIC;
  libxsmm_target_archid = id;
}
LIBXSMM_API const char* libxsmm_get_target_arch(void) { return "gfx950"; }
LIBXSMM_API void libxsmm_set_target_arch(const char* arch) {
  if (!arch) return;
  if (std::strcmp(arch, "gfx950") != 0 && std::strcmp(arch, "generic") != 0 && std::strcmp(arch, "0") != 0)
    vlog(1, "libxsmm_set_target_arch(\"%s\"): ignored, kernels are built for gfx950 only", arch);
  if (std::strcmp(arch, "0") == 0) libxsmm_target_archid = LIBXSMM_TARGET_ARCH_GENERIC; else libxsmm_target_archid = LIBXSMM_X86_GENERIC;
}
LIBXSMM_API const char* libxsmm_get_typename(libxsmm_datatype t) { return ((int)t >= 0 && t < LIBXSMM_DATATYPE_COUNT_) ? kTypeNames[t] : "void"; }
LIBXSMM_API unsigned char libxsmm_typesize(libxsmm_datatype t) { return (unsigned char)typesize((int)t); }
LIBXSMM_API int libxsmm_get_verbosity(void) { return libxsmm_verbosity; }
LIBXSMM_API void libxsmm_set_verbosity(int level) { libxsmm_verbosity = level; }
LIBXSMM_API int libxsmm_cpuid(void* info) { (void)info; return LIBXSMM_TARGET_ARCH_GENERIC; }
/* drivers pre-pack bf16 A with this factor; it stays the x86 value [ref: src/libxsmm_cpuid_x86.c:775] */
LIBXSMM_API int libxsmm_cpuid_dot_pack_factor(libxsmm_datatype t) {
  switch (t) {      // [ref: src/libxsmm_cpuid_x86.c:775-797]
    case LIBXSMM_DATATYPE_BF16: case LIBXSMM_DATATYPE_F16: case LIBXSMM_DATATYPE_I16: case LIBXSMM_DATATYPE_U16: return 2;
    case LIBXSMM_DATATYPE_I8: case LIBXSMM_DATATYPE_U8: case LIBXSMM_DATATYPE_BF8: case LIBXSMM_DATATYPE_HF8:
    case LIBXSMM_DATATYPE_MXBF8: case LIBXSMM_DATATYPE_MXHF8: case LIBXSMM_DATATYPE_MXBF6: case LIBXSMM_DATATYPE_MXHF6: return 4;
    case LIBXSMM_DATATYPE_MXFP4X2: return 8;
    default: return 1;
  }
}
LIBXSMM_API int libxsmm_cpuid_vlen(int id) { (void)id; return 64; }
// 32-bit lanes per "vector": the rows DROPOUT draws for at a time (the reference's gold loops ask this too) [ref: src/libxsmm_cpuid_x86.c:670]
LIBXSMM_API int libxsmm_cpuid_vlen32(int id) { (void)id; return 16; }

// ---- shapes / configs -----------------------------------------------------------------------------------
LIBXSMM_API libxsmm_gemm_shape libxsmm_create_gemm_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, libxsmm_datatype a, libxsmm_datatype b, libxsmm_datatype out, libxsmm_datatype comp) {
  libxsmm_gemm_shape s; std::memset(&s, 0, sizeof(s));
  s.m = m; s.n = n; s.k = k; s.lda = lda; s.ldb = ldb; s.ldc = ldc; s.a_in_type = a; s.b_in_type = b; s.out_type = out; s.comp_type = comp;
  return s;
}
LIBXSMM_API libxsmm_gemm_batch_reduce_config libxsmm_create_gemm_batch_reduce_config(libxsmm_gemm_batch_reduce_type t,
  libxsmm_blasint sa, libxsmm_blasint sb, unsigned char unroll) {
  libxsmm_gemm_batch_reduce_config c; std::memset(&c, 0, sizeof(c));
  c.br_type = t; c.br_stride_a_hint = sa; c.br_stride_b_hint = sb; c.br_unroll_hint = unroll; return c;
}
LIBXSMM_API libxsmm_gemm_ext_unary_argops libxsmm_create_gemm_ext_unary_argops(
  libxsmm_blasint ldap, libxsmm_meltw_unary_type apt, libxsmm_bitfield apf, libxsmm_blasint sap,
  libxsmm_blasint ldbp, libxsmm_meltw_unary_type bpt, libxsmm_bitfield bpf, libxsmm_blasint sbp,
  libxsmm_blasint ldcp, libxsmm_meltw_unary_type cpt, libxsmm_bitfield cpf, libxsmm_blasint scp) {
  libxsmm_gemm_ext_unary_argops r; std::memset(&r, 0, sizeof(r));
  r.ldap = ldap; r.ap_unary_type = apt; r.ap_unary_flags = apf; r.store_ap = sap;
  r.ldbp = ldbp; r.bp_unary_type = bpt; r.bp_unary_flags = bpf; r.store_bp = sbp;
  r.ldcp = ldcp; r.cp_unary_type = cpt; r.cp_unary_flags = cpf; r.store_cp = scp; return r;
}
LIBXSMM_API libxsmm_gemm_ext_binary_postops libxsmm_create_gemm_ext_binary_postops(libxsmm_blasint ldd, libxsmm_datatype dt,
  libxsmm_meltw_binary_type bt, libxsmm_bitfield bf) {
  libxsmm_gemm_ext_binary_postops r; std::memset(&r, 0, sizeof(r));
  r.ldd = ldd; r.d_in_type = dt; r.d_binary_type = bt; r.d_binary_flags = bf; return r;
}
LIBXSMM_API libxsmm_meltw_unary_shape libxsmm_create_meltw_unary_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ldi, libxsmm_blasint ldo,
  libxsmm_datatype in0, libxsmm_datatype out, libxsmm_datatype comp) {
  libxsmm_meltw_unary_shape s; std::memset(&s, 0, sizeof(s));
  s.m = m; s.n = n; s.ldi = ldi; s.ldo = ldo; s.in0_type = in0; s.out_type = out; s.comp_type = comp; return s;
}
LIBXSMM_API libxsmm_meltw_binary_shape libxsmm_create_meltw_binary_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ldi, libxsmm_blasint ldi2, libxsmm_blasint ldo,
  libxsmm_datatype in0, libxsmm_datatype in1, libxsmm_datatype out, libxsmm_datatype comp) {
  libxsmm_meltw_binary_shape s; std::memset(&s, 0, sizeof(s));
  s.m = m; s.n = n; s.ldi = ldi; s.ldi2 = ldi2; s.ldo = ldo; s.in0_type = in0; s.in1_type = in1; s.out_type = out; s.comp_type = comp; return s;
}
LIBXSMM_API libxsmm_meltw_ternary_shape libxsmm_create_meltw_ternary_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ldi, libxsmm_blasint ldi2, libxsmm_blasint ldi3, libxsmm_blasint ldo,
  libxsmm_datatype in0, libxsmm_datatype in1, libxsmm_datatype in2, libxsmm_datatype out, libxsmm_datatype comp) {
  libxsmm_meltw_ternary_shape s; std::memset(&s, 0, sizeof(s));
  s.m = m; s.n = n; s.ldi = ldi; s.ldi2 = ldi2; s.ldi3 = ldi3; s.ldo = ldo; s.in0_type = in0; s.in1_type = in1; s.in2_type = in2; s.out_type = out; s.comp_type = comp; return s;
}

// ---- descriptors ---------------------------------------------------------------------------------------------
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_datatype a_type, libxsmm_datatype b_type, libxsmm_datatype comp_type, libxsmm_datatype c_type,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, int flags, int prefetch) {
  if (!blob || m < 0 || n < 0 || k < 0 || lda < 0 || ldb < 0 || ldc < 0) return nullptr;
  std::memset(blob, 0, sizeof(*blob));
  libxsmm_gemm_descriptor* d = reinterpret_cast<libxsmm_gemm_descriptor*>(blob);
  d->m = (uint32_t)m; d->n = (uint32_t)n; d->k = (uint32_t)k; d->lda = (uint32_t)lda; d->ldb = (uint32_t)ldb; d->ldc = (uint32_t)ldc;
  d->flags = (uint32_t)flags; d->prefetch = (uint8_t)prefetch;
  d->a_type = (uint8_t)a_type; d->b_type = (uint8_t)b_type; d->c_type = (uint8_t)c_type; d->comp_type = (uint8_t)comp_type;
  return d;
}

static libxsmm_gemm_descriptor* init_br(libxsmm_descriptor_blob* blob, const libxsmm_gemm_shape& s, libxsmm_bitfield flags,
  libxsmm_bitfield prefetch, const libxsmm_gemm_batch_reduce_config* br, unsigned int abi) {
  unsigned int f = (unsigned int)flags;
  if (tilecfg_halfset(f)) return nullptr;                               // [ref: generator.c:154-157,187-190]
  f |= abi;
  if (br) {
    if (br->br_type == LIBXSMM_GEMM_BATCH_REDUCE_ADDRESS) f |= LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS;
    else if (br->br_type == LIBXSMM_GEMM_BATCH_REDUCE_OFFSET) f |= LIBXSMM_GEMM_FLAG_BATCH_REDUCE_OFFSET;
    else if (br->br_type == LIBXSMM_GEMM_BATCH_REDUCE_STRIDE) f |= LIBXSMM_GEMM_FLAG_BATCH_REDUCE_STRIDE;
  }
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init(blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type,
    s.m, s.n, s.k, s.lda, s.ldb, s.ldc, (int)f, (int)prefetch);
  if (d && br && br->br_type != LIBXSMM_GEMM_BATCH_REDUCE_NONE) {
    if (br->br_type == LIBXSMM_GEMM_BATCH_REDUCE_STRIDE) { d->br_stride_a = br->br_stride_a_hint; d->br_stride_b = br->br_stride_b_hint; }   // bytes [ref: generator.c:215-217]
    d->br_unroll = (br->br_unroll_hint > 0 && br->br_unroll_hint < 255) ? br->br_unroll_hint : 0;
  }
  return d;
}
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init_gemm(libxsmm_descriptor_blob* blob, libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch) {
  return init_br(blob, s, flags, prefetch, nullptr, LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI);
}
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init_brgemm(libxsmm_descriptor_blob* blob, libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_gemm_batch_reduce_config br) {
  return init_br(blob, s, flags, prefetch, &br, LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI);
}
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init_brgemm_ext(libxsmm_descriptor_blob* blob, libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_gemm_batch_reduce_config br, libxsmm_gemm_ext_unary_argops u, libxsmm_gemm_ext_binary_postops bp) {
  libxsmm_gemm_descriptor* d = init_br(blob, s, flags, prefetch, &br, LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI);
  if (!d) return nullptr;
  d->d_type = (uint8_t)bp.d_in_type; d->bin_type = (uint16_t)bp.d_binary_type; d->bin_flags = (uint16_t)bp.d_binary_flags; d->ldd = (uint32_t)bp.ldd;
  d->ap_type = (uint16_t)u.ap_unary_type; d->ap_flags = (uint16_t)u.ap_unary_flags; d->ldap = (uint32_t)u.ldap;
  d->bp_type = (uint16_t)u.bp_unary_type; d->bp_flags = (uint16_t)u.bp_unary_flags; d->ldbp = (uint32_t)u.ldbp;
  d->cp_type = (uint16_t)u.cp_unary_type; d->cp_flags = (uint16_t)u.cp_unary_flags; d->ldcp = (uint32_t)u.ldcp;
  d->store_mask = (uint8_t)((u.store_ap ? 1 : 0) | (u.store_bp ? 2 : 0) | (u.store_cp ? 4 : 0));
  return d;
}
LIBXSMM_API libxsmm_meltw_descriptor* libxsmm_meltw_descriptor_init2(libxsmm_descriptor_blob* blob,
  libxsmm_datatype in0, libxsmm_datatype in1, libxsmm_datatype in2, libxsmm_datatype comp, libxsmm_datatype out,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ldi, libxsmm_blasint ldo, libxsmm_blasint ldi2, libxsmm_blasint ldi3,
  unsigned short flags, unsigned short param, unsigned char operation) {
  if (!blob) return nullptr;
  std::memset(blob, 0, sizeof(*blob));
  libxsmm_meltw_descriptor* d = reinterpret_cast<libxsmm_meltw_descriptor*>(blob);
  d->m = (uint32_t)m; d->n = (uint32_t)n; d->ldi = (uint32_t)ldi; d->ldo = (uint32_t)ldo; d->ldi2 = (uint32_t)ldi2; d->ldi3 = (uint32_t)ldi3;
  d->in0_type = (uint8_t)in0; d->in1_type = (uint8_t)in1; d->in2_type = (uint8_t)in2; d->comp_type = (uint8_t)comp; d->out_type = (uint8_t)out;
  d->operation = operation; d->flags = flags; d->param = param;
  return d;
}
LIBXSMM_API libxsmm_meltw_descriptor* libxsmm_meltw_descriptor_init(libxsmm_descriptor_blob* blob, libxsmm_datatype in_type, libxsmm_datatype out_type,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ldi, libxsmm_blasint ldo, unsigned short flags, unsigned short param, unsigned char operation) {
  return libxsmm_meltw_descriptor_init2(blob, in_type, LIBXSMM_DATATYPE_IMPLICIT, LIBXSMM_DATATYPE_IMPLICIT, LIBXSMM_DATATYPE_IMPLICIT, out_type, m, n, ldi, ldo, 0, 0, flags, param, operation);
}

// ---- dispatch -------------------------------------------------------------------------------------------------
}  // extern "C" (the dispatch cache below is C++)

// Thread-local dispatch cache in front of the locked registry [ref: libxsmm_main.c:2739-2763, 16 entries like LIBXSMM_CACHE_MAXSIZE].
// The hit path touches no heap and takes no lock: build the fixed-size key, hash it, compare one direct-mapped entry.
struct DispatchCacheEntry { RegKey key; const void* fn; unsigned int generation; };
static inline void make_key(RegKey& key, Kind kind, const void* desc, size_t size) {
  std::memset(key.w, 0, sizeof(key.w));
  key.w[0] = (unsigned long long)kind;
  std::memcpy(&key.w[1], desc, std::min(size, (size_t)LIBXSMM_DESCRIPTOR_MAXSIZE));
}
static inline const void* cache_lookup(const RegKey& key, unsigned long long h, DispatchCacheEntry*& entry) {
  thread_local DispatchCacheEntry cache[16];
  entry = &cache[h & 15u];
  if (entry->fn && entry->generation == g_generation.load(std::memory_order_acquire) && entry->key == key) return entry->fn;
  return nullptr;
}
static const char* meltw_label(const libxsmm_meltw_descriptor& e);
// `supported`: evaluated on a miss only -- a cached descriptor has passed it before
template <typename Supported>
static const void* find_or_build(Kind kind, const void* desc, size_t size, Supported&& supported) {
  RegKey key; make_key(key, kind, desc, size);
  const unsigned long long h = hash_key(key);
  DispatchCacheEntry* entry = nullptr;
  if (const void* hit = cache_lookup(key, h, entry)) return hit;
  if (!supported()) return nullptr;
  std::lock_guard<std::mutex> guard(g_lock);
  auto it = g_registry.find(key);
  KernelCtx* c = nullptr;
  if (it != g_registry.end()) c = it->second;
  else {
    if ((int)g_registry.size() >= g_registered_limit) { vlog(1, "registry is full (%d kernels)", g_registered_limit); return nullptr; }
    c = new_ctx_locked(kind);
    if (!c) return nullptr;
    c->registered = true;
    if (kind == K_GEMM || kind == K_TILECFG) {
      std::memcpy(&c->g, desc, sizeof(c->g));
      c->nflops = (unsigned int)(2ull * c->g.m * c->g.n * c->g.k);
      { libxsmm_gemm_descriptor e = c->g; e.flags = effective_gemm_flags(e); c->kname_single = gemm_kernel_name(e, false); c->kname_batched = gemm_kernel_name(e, true); }
    } else {
      std::memcpy(&c->e, desc, sizeof(c->e));
      c->nflops = c->e.m * c->e.n;
      c->kname_single = c->kname_batched = meltw_label(c->e);     // replaced by the device kernel's name at the first launch
    }
    g_registry.emplace(key, c);
    vlog(2, "built kernel #%d kind=%d", c->slot, (int)kind);
  }
  const void* fn = handle_for_slot(c->slot);
  entry->key = key; entry->fn = fn; entry->generation = g_generation.load(std::memory_order_acquire);
  return fn;
}
// TPP handles are told apart before their first launch by operation and type: "meltw_unary#RELU" is not available without the
// enum's names, so the label carries the numeric type as the reference's own kernel names do [ref: src/libxsmm_main.c:2420-2440].
static const char* meltw_label(const libxsmm_meltw_descriptor& e) {
  static std::mutex lock; static std::unordered_map<unsigned int, std::string> names;
  const unsigned int id = ((unsigned int)e.operation << 16) | e.param;
  std::lock_guard<std::mutex> guard(lock);
  auto it = names.find(id);
  if (it == names.end()) {
    const char* op = e.operation == LIBXSMM_MELTW_OPERATION_UNARY ? "unary" : e.operation == LIBXSMM_MELTW_OPERATION_BINARY ? "binary" : "ternary";
    it = names.emplace(id, std::string("meltw_") + op + "_type" + std::to_string((int)e.param)).first;
  }
  return it->second.c_str();    // node-based map: the string never moves
}

extern "C" {

LIBXSMM_API libxsmm_xmmfunction libxsmm_xmmdispatch(const libxsmm_gemm_descriptor* d) {
  libxsmm_xmmfunction r; r.ptr_const = nullptr;
  if (!d || !runtime_ready()) return r;
  const bool a = (d->flags & LIBXSMM_GEMM_FLAG_NO_RESET_TILECONFIG) != 0, b = (d->flags & LIBXSMM_GEMM_FLAG_NO_SETUP_TILECONFIG) != 0;
  if (a != b) { r.ptr_const = find_or_build(K_TILECFG, d, sizeof(*d), []() { return true; }); return r; }
  r.ptr_const = find_or_build(K_GEMM, d, sizeof(*d), [d]() {
    libxsmm_gemm_descriptor e = *d; e.flags = effective_gemm_flags(e);
    if (gemm_supported(e)) return true;
    vlog(1, "unsupported GEMM descriptor (types %s/%s/%s, flags 0x%x)", kTypeNames[d->a_type], kTypeNames[d->b_type], kTypeNames[d->c_type], d->flags);
    return false; });
  return r;
}
LIBXSMM_API libxsmm_gemmfunction libxsmm_dispatch_gemm(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch) {
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init_gemm(&blob, s, flags, prefetch);
  return d ? libxsmm_xmmdispatch(d).gemm : nullptr;
}
LIBXSMM_API libxsmm_gemmfunction libxsmm_dispatch_brgemm(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_gemm_batch_reduce_config br) {
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init_brgemm(&blob, s, flags, prefetch, br);
  return d ? libxsmm_xmmdispatch(d).gemm : nullptr;
}
LIBXSMM_API libxsmm_gemmfunction_ext libxsmm_dispatch_brgemm_ext(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_gemm_batch_reduce_config br,
  libxsmm_gemm_ext_unary_argops u, libxsmm_gemm_ext_binary_postops bp) {
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init_brgemm_ext(&blob, s, flags, prefetch, br, u, bp);
  if (!d) return nullptr;
  // fusions this backend implements: column-bias add, ReLU (+bitmask), sigmoid on C; anything else is refused
  const bool bin_ok = d->bin_type == LIBXSMM_MELTW_TYPE_BINARY_NONE ||
    (d->bin_type == LIBXSMM_MELTW_TYPE_BINARY_ADD && (d->bin_flags & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0 | LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_1)));
  const bool cp_ok = d->cp_type == LIBXSMM_MELTW_TYPE_UNARY_NONE || d->cp_type == LIBXSMM_MELTW_TYPE_UNARY_RELU || d->cp_type == LIBXSMM_MELTW_TYPE_UNARY_SIGMOID;
  if (!bin_ok || !cp_ok || d->ap_type != 0 || d->bp_type != 0) { vlog(1, "unsupported fused op in BRGEMM_ext"); return nullptr; }
  return libxsmm_xmmdispatch(d).gemm_ext;
}
LIBXSMM_API libxsmm_tilecfgfunction libxsmm_dispatch_tilecfg_gemm(libxsmm_gemm_shape s, libxsmm_bitfield flags) {
  // only meaningful with exactly one of the two tile-config flags [ref: libxsmm_main.c:3355-3387]
  if (!tilecfg_halfset((unsigned int)flags) || !runtime_ready()) return nullptr;
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init(&blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
    (int)(flags | LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI), 0);
  if (!d) return nullptr;
  libxsmm_xmmfunction r; r.ptr_const = find_or_build(K_TILECFG, d, sizeof(*d), []() { return true; });
  return r.tilecfg;
}

LIBXSMM_API libxsmm_xmeltwfunction libxsmm_dispatch_meltw(const libxsmm_meltw_descriptor* d) {
  libxsmm_xmeltwfunction r; r.xmeltw = nullptr;
  if (!d || !runtime_ready()) return r;
  r.xmeltw = (void (*)(const void*))find_or_build(K_MELTW, d, sizeof(*d), [d]() {
    if (meltw_supported(*d)) return true;
    vlog(1, "unsupported TPP (operation %d, type %d, in %s, out %s)", d->operation, d->param, kTypeNames[d->in0_type], kTypeNames[d->out_type]);
    return false; });
  return r;
}
LIBXSMM_API libxsmm_meltwfunction_unary libxsmm_dispatch_meltw_unary(libxsmm_meltw_unary_type t, libxsmm_meltw_unary_shape s, libxsmm_bitfield f) {
  libxsmm_descriptor_blob blob;   // [ref: libxsmm_main.c:3472-3483]
  return libxsmm_dispatch_meltw(libxsmm_meltw_descriptor_init2(&blob, s.in0_type, LIBXSMM_DATATYPE_UNSUPPORTED, LIBXSMM_DATATYPE_UNSUPPORTED, s.comp_type, s.out_type,
    s.m, s.n, s.ldi, s.ldo, 0, 0, (unsigned short)f, (unsigned short)t, LIBXSMM_MELTW_OPERATION_UNARY)).meltw_unary;
}
LIBXSMM_API libxsmm_meltwfunction_binary libxsmm_dispatch_meltw_binary(libxsmm_meltw_binary_type t, libxsmm_meltw_binary_shape s, libxsmm_bitfield f) {
  libxsmm_descriptor_blob blob;
  return libxsmm_dispatch_meltw(libxsmm_meltw_descriptor_init2(&blob, s.in0_type, s.in1_type, LIBXSMM_DATATYPE_UNSUPPORTED, s.comp_type, s.out_type,
    s.m, s.n, s.ldi, s.ldo, s.ldi2, 0, (unsigned short)f, (unsigned short)t, LIBXSMM_MELTW_OPERATION_BINARY)).meltw_binary;
}
LIBXSMM_API libxsmm_meltwfunction_ternary libxsmm_dispatch_meltw_ternary(libxsmm_meltw_ternary_type t, libxsmm_meltw_ternary_shape s, libxsmm_bitfield f) {
  libxsmm_descriptor_blob blob;
  return libxsmm_dispatch_meltw(libxsmm_meltw_descriptor_init2(&blob, s.in0_type, s.in1_type, s.in2_type, s.comp_type, s.out_type,
    s.m, s.n, s.ldi, s.ldo, s.ldi2, s.ldi3, (unsigned short)f, (unsigned short)t, LIBXSMM_MELTW_OPERATION_TERNARY)).meltw_ternary;
}

// ---- packed / sparse creators (caller owned) ---------------------------------------------------------------------
static KernelCtx* new_unregistered(Kind kind, const libxsmm_gemm_descriptor* d) {
  std::lock_guard<std::mutex> guard(g_lock);
  KernelCtx* c = new_ctx_locked(kind);
  if (c) { c->g = *d; c->registered = false; }
  return c;
}
static void drop_unregistered(KernelCtx* c) { std::lock_guard<std::mutex> guard(g_lock); free_ctx_locked(c); }

// build the device copy of a "rows -> list of (inner index, value position)" pattern
static bool upload_pattern(KernelCtx* c, int rows, int inner, const unsigned int* ptr, const unsigned int* idx, const unsigned int* vmap) {
  const unsigned int nnz = ptr[rows];
  for (unsigned int z = 0; z < nnz; ++z) if ((int)idx[z] >= inner) return false;
  c->sp_rows = rows; c->sp_inner = inner; c->sp_nnz = nnz;
  c->kname_single = c->kname_batched = "spmm_stream_kernel";
  c->d_ptr = to_device(ptr, (size_t)rows + 1); c->d_idx = to_device(idx, nnz);
  if (vmap) c->d_vmap = to_device(vmap, nnz);
  return c->d_ptr && c->d_idx && (!vmap || c->d_vmap);
}

// specialise the kernel for this pattern.  Mode (libxsmm_hip_set_jit / LIBXSMM_HIP_JIT): 0 never, 1 auto (default: only when
// one call covers enough columns to repay ~0.1 s of hiprtc), 2 always.  Failure is not an error: precompiled kernels serve.
static int g_jit_mode = -1;
static int jit_mode() {
  if (g_jit_mode < 0) { const char* e = getenv("LIBXSMM_HIP_JIT"); g_jit_mode = (e && e[0] >= '0' && e[0] <= '2') ? (e[0] - '0') : 1; }
  return g_jit_mode;
}
static void attach_jit(KernelCtx* c, const unsigned int* ptr, const unsigned int* idx, const unsigned int* vmap, long long batch) {
  const int mode = jit_mode();
  if (mode == 0) return;
  SpmmArgs a{}; spmm_geometry(c, a);
  if (mode == 1 && a.ncols * a.nouter * batch < 4096) {  // ~0.1 s of hiprtc is not repaid by launch-bound toy sizes ...
    if (batch == 1 && c->h_ptr.empty()) {                // ... unless a batched launch later covers enough columns: keep the pattern
      c->h_ptr.assign(ptr, ptr + a.rows + 1); c->h_idx.assign(idx, idx + ptr[a.rows]);
      if (vmap) c->h_vmap.assign(vmap, vmap + ptr[a.rows]);
    }
    return;
  }
  SpmmJitSpec s{};
  s.dtype = a.dtype; s.rows = a.rows; s.inner = a.inner; s.nouter = (int)std::min<long long>(a.nouter * batch, 1 << 30); s.beta0 = a.beta0; s.skip_empty = a.skip_empty;
  s.ptr = ptr; s.idx = idx; s.vmap = vmap; s.ld_x = a.ld_x; s.ld_y = a.ld_y; s.outer_x = a.outer_x; s.outer_y = a.outer_y; s.ncols = a.ncols;
  std::string why;
  c->jit = jit_spmm_create(s, &why);
  if (c->jit) { c->kname_single = c->kname_batched = jit_name(c->jit); vlog(2, "JIT %s (%zu bytes of code)", jit_name(c->jit), jit_code_size(c->jit)); }
  else vlog(2, "sparse kernel not specialised: %s", why.c_str());
}

LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_spgemm_csr(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_blasint packed_width, const unsigned int* row_ptr, const unsigned int* column_idx, const void* values) {
  if (!runtime_ready()) return nullptr;
  if (s.a_in_type != s.b_in_type || !row_ptr || !column_idx || !values) return nullptr;             // [ref: libxsmm_main.c:3567-3572]
  if ((s.a_in_type != LIBXSMM_DATATYPE_F32 && s.a_in_type != LIBXSMM_DATATYPE_F64) || s.out_type != s.a_in_type) return nullptr;   // [ref: :2353-2354]
  if (packed_width <= 0 || tilecfg_halfset((unsigned int)flags) || (flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B))) return nullptr;
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init(&blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
    (int)(flags | LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI), (int)prefetch);
  if (!d) return nullptr;
  KernelCtx* c = nullptr; bool ok = false;
  if (s.lda == 0 && s.ldb > 0 && s.ldc > 0) {          // A sparse [ref: generator_packed_spgemm.c:27]
    if (s.ldb < s.n || s.ldc < s.n) return nullptr;
    c = new_unregistered(K_SPMM_ASPARSE, d); if (!c) return nullptr;
    c->packed_width = packed_width; c->sp_ncols = s.n; c->sp_skip_empty = 1;
    ok = upload_pattern(c, s.m, s.k, row_ptr, column_idx, nullptr);
    if (ok) attach_jit(c, row_ptr, column_idx, nullptr, 1);
    c->nflops = (unsigned int)(2ull * row_ptr[s.m] * s.n * packed_width);   // [ref: libxsmm_main.c:2356-2359]
  } else if (s.ldb == 0 && s.lda > 0 && s.ldc > 0) {   // B sparse, CSR over rows k -> regroup by output column n
    if (s.lda < s.k || s.ldc < s.n) return nullptr;
    const unsigned int nnz = row_ptr[s.k];
    std::vector<unsigned int> cptr((size_t)s.n + 1, 0), ridx(nnz), vmap(nnz);
    for (unsigned int z = 0; z < nnz; ++z) { if ((int)column_idx[z] >= s.n) return nullptr; ++cptr[column_idx[z] + 1]; }
    for (int n = 0; n < s.n; ++n) cptr[n + 1] += cptr[n];
    std::vector<unsigned int> fill(cptr.begin(), cptr.end() - 1);
    for (int k = 0; k < s.k; ++k) for (unsigned int z = row_ptr[k]; z < row_ptr[k + 1]; ++z) { const unsigned int q = fill[column_idx[z]]++; ridx[q] = (unsigned int)k; vmap[q] = z; }
    c = new_unregistered(K_SPMM_BSPARSE, d); if (!c) return nullptr;
    c->packed_width = packed_width;
    ok = upload_pattern(c, s.n, s.k, cptr.data(), ridx.data(), vmap.data());
    if (ok) attach_jit(c, cptr.data(), ridx.data(), vmap.data(), 1);
    c->nflops = (unsigned int)(2ull * nnz * s.m * packed_width);
  } else return nullptr;                                 // C sparse: not on the hot path
  if (!ok) { drop_unregistered(c); return nullptr; }
  return (libxsmm_gemmfunction)handle_for_slot(c->slot);
}

LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_spgemm_csc(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_blasint packed_width, const unsigned int* column_ptr, const unsigned int* row_idx, const void* values) {
  if (!runtime_ready()) return nullptr;
  if (s.a_in_type != s.b_in_type || !column_ptr || !row_idx || !values) return nullptr;               // [ref: libxsmm_main.c:3611-3616]
  if ((s.a_in_type != LIBXSMM_DATATYPE_F32 && s.a_in_type != LIBXSMM_DATATYPE_F64) || s.out_type != s.a_in_type) return nullptr;
  if (packed_width <= 0 || tilecfg_halfset((unsigned int)flags) || (flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B))) return nullptr;
  if (s.lda > 0 && s.ldb > 0 && s.ldc == 0) {
    // C sparse [ref: src/generator_packed_spgemm.c:81-94]: f32 only [ref: generator_packed_spgemm_csc_csparse_avx_avx2_avx512.c:614-630], ldb >= n;
    // the pattern (column_ptr over n, row_idx) is C's, the values array is not read at creation
    if (s.a_in_type != LIBXSMM_DATATYPE_F32 || s.ldb < s.n || s.lda < s.m) return nullptr;
    libxsmm_descriptor_blob blob;
    libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init(&blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
      (int)(flags | LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI), (int)prefetch);
    if (!d) return nullptr;
    const unsigned int nnz = column_ptr[s.n];
    std::vector<unsigned int> cols(std::max(1u, nnz));
    for (libxsmm_blasint n = 0; n < s.n; ++n) {
      if (column_ptr[n] > column_ptr[n + 1] || column_ptr[n + 1] > nnz) return nullptr;
      for (unsigned int z = column_ptr[n]; z < column_ptr[n + 1]; ++z) { if ((libxsmm_blasint)row_idx[z] >= s.m) return nullptr; cols[z] = (unsigned int)n; }
    }
    KernelCtx* c = new_unregistered(K_SPMM_CSPARSE, d); if (!c) return nullptr;
    c->packed_width = packed_width; c->sp_nnz = nnz;
    c->d_idx = to_device(row_idx, nnz); c->d_vmap = to_device(cols.data(), nnz);
    if (!c->d_idx || !c->d_vmap) { drop_unregistered(c); return nullptr; }
    c->nflops = (unsigned int)(2ull * nnz * s.k * packed_width);
    c->kname_single = c->kname_batched = "csparse_kernel";
    return (libxsmm_gemmfunction)handle_for_slot(c->slot);
  }
  if (!(s.ldb == 0 && s.lda >= s.k && s.ldc >= s.n)) return nullptr;                                  // otherwise: B sparse
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init(&blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
    (int)(flags | LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI), (int)prefetch);
  if (!d) return nullptr;
  KernelCtx* c = new_unregistered(K_SPMM_BSPARSE, d); if (!c) return nullptr;
  c->packed_width = packed_width;
  if (!upload_pattern(c, s.n, s.k, column_ptr, row_idx, nullptr)) { drop_unregistered(c); return nullptr; }
  attach_jit(c, column_ptr, row_idx, nullptr, 1);
  c->nflops = (unsigned int)(2ull * column_ptr[s.n] * s.m * packed_width);
  return (libxsmm_gemmfunction)handle_for_slot(c->slot);
}

LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_spgemm_bcsc(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_spgemm_config cfg) {
  if (!runtime_ready()) return nullptr;
  if (tilecfg_halfset((unsigned int)flags)) return nullptr;                                            // [ref: libxsmm_main.c:3664-3667]
  const bool f32 = s.a_in_type == LIBXSMM_DATATYPE_F32 && s.b_in_type == LIBXSMM_DATATYPE_F32 && s.out_type == LIBXSMM_DATATYPE_F32;
  const bool bf16 = s.a_in_type == LIBXSMM_DATATYPE_BF16 && s.b_in_type == LIBXSMM_DATATYPE_BF16 && (s.out_type == LIBXSMM_DATATYPE_BF16 || s.out_type == LIBXSMM_DATATYPE_F32);
  // 8-bit integers: unsigned A x signed B or signed A x unsigned B -> int32, A in VNNI-4 [ref: samples/xgemm_sparse/spmm_kernel.c:851-856, :254-262]
  const bool i8 = ((s.a_in_type == LIBXSMM_DATATYPE_U8 && s.b_in_type == LIBXSMM_DATATYPE_I8) || (s.a_in_type == LIBXSMM_DATATYPE_I8 && s.b_in_type == LIBXSMM_DATATYPE_U8)) &&
    s.out_type == LIBXSMM_DATATYPE_I32;
  if (!(f32 || bf16 || i8)) return nullptr;
  if (flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_VNNI_B | LIBXSMM_GEMM_FLAG_VNNI_C)) return nullptr;
  if (f32 && (flags & LIBXSMM_GEMM_FLAG_VNNI_A)) return nullptr;
  if (i8 && (!(flags & LIBXSMM_GEMM_FLAG_VNNI_A) || (s.k & 3))) return nullptr;
  if (cfg.packed_width <= 0 || cfg.bk <= 0 || cfg.bn <= 0 || s.k % cfg.bk != 0 || s.ldc % cfg.bn != 0 || s.ldb != 0) return nullptr;
  if (bf16 && (flags & LIBXSMM_GEMM_FLAG_VNNI_A) && (s.k & 1)) return nullptr;
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init(&blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
    (int)(flags | LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI), (int)prefetch);
  if (!d) return nullptr;
  KernelCtx* c = new_unregistered(K_BCSC, d); if (!c) return nullptr;
  c->packed_width = cfg.packed_width; c->bk = cfg.bk; c->bn = cfg.bn;
  c->nflops = 0; c->kname_single = c->kname_batched = "bcsc_kernel";
  return (libxsmm_gemmfunction)handle_for_slot(c->slot);
}

LIBXSMM_API libxsmm_tilecfgfunction libxsmm_create_tilecfg_packed_spgemm_bcsc(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_spgemm_config cfg) {
  (void)cfg;
  return libxsmm_dispatch_tilecfg_gemm(s, flags);
}

// ---- dense packed GEMMs [ref: libxsmm_main.c:3733-3840] -----------------------------------------------------------
static libxsmm_gemm_descriptor* packed_descriptor(libxsmm_descriptor_blob* blob, const libxsmm_gemm_shape& s, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_blasint packed_width) {
  if (!runtime_ready() || packed_width <= 0) return nullptr;
  if (s.a_in_type != s.b_in_type || (s.a_in_type != LIBXSMM_DATATYPE_F32 && s.a_in_type != LIBXSMM_DATATYPE_F64) || s.out_type != s.a_in_type) return nullptr;
  if (tilecfg_halfset((unsigned int)flags) || (flags & (LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B))) return nullptr;
  if (s.m <= 0 || s.n <= 0 || s.k <= 0) return nullptr;
  return libxsmm_gemm_descriptor_init(blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
    (int)(flags | LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI), (int)prefetch);
}
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_gemm(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_blasint packed_width) {
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = packed_descriptor(&blob, s, flags, prefetch, packed_width);
  if (!d || s.lda < s.m || s.ldb < s.k || s.ldc < s.m) return nullptr;            // column-major, every element P wide
  KernelCtx* c = new_unregistered(K_PGEMM, d); if (!c) return nullptr;
  c->packed_width = packed_width;
  c->nflops = (unsigned int)(2ull * s.m * s.n * s.k * packed_width);
  c->kname_single = c->kname_batched = "pgemm_generic_kernel";
  if (jit_mode() != 0) {
    PgemmArgs g{}; pgemm_geometry(c, g);
    std::string why;
    c->jit = jit_pgemm_create(g, &why);
    if (c->jit) { c->kname_single = c->kname_batched = jit_name(c->jit); vlog(2, "JIT %s (%zu bytes of code)", jit_name(c->jit), jit_code_size(c->jit)); }
    else vlog(2, "packed GEMM not specialised: %s", why.c_str());
  }
  return (libxsmm_gemmfunction)handle_for_slot(c->slot);
}
// A and C packed, B a plain row-major K x N matrix: the fixed-pattern kernel with a dense pattern whose values are
// read from B at run time (value of (n, k) at B[k*ldb + n])
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_gemm_ac_rm(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_blasint packed_width) {
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = packed_descriptor(&blob, s, flags, prefetch, packed_width);
  if (!d || s.lda < s.k || s.ldb < s.n || s.ldc < s.n) return nullptr;
  if ((long long)s.n * s.k >= (1ll << 28)) return nullptr;
  std::vector<unsigned int> ptr((size_t)s.n + 1), idx((size_t)s.n * s.k), vmap((size_t)s.n * s.k);
  for (int n = 0; n <= s.n; ++n) ptr[n] = (unsigned int)((long long)n * s.k);
  for (int n = 0; n < s.n; ++n) for (int k = 0; k < s.k; ++k) { idx[(size_t)n * s.k + k] = (unsigned int)k; vmap[(size_t)n * s.k + k] = (unsigned int)((long long)k * s.ldb + n); }
  KernelCtx* c = new_unregistered(K_SPMM_BSPARSE, d); if (!c) return nullptr;
  c->packed_width = packed_width;
  if (!upload_pattern(c, s.n, s.k, ptr.data(), idx.data(), vmap.data())) { drop_unregistered(c); return nullptr; }
  attach_jit(c, ptr.data(), idx.data(), vmap.data());
  c->nflops = (unsigned int)(2ull * s.m * s.n * s.k * packed_width);
  return (libxsmm_gemmfunction)handle_for_slot(c->slot);
}
// B and C packed, A a plain row-major M x K matrix (value of (m, k) at A[m*lda + k])
LIBXSMM_API libxsmm_gemmfunction libxsmm_create_packed_gemm_bc_rm(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_blasint packed_width) {
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = packed_descriptor(&blob, s, flags, prefetch, packed_width);
  if (!d || s.lda < s.k || s.ldb < s.n || s.ldc < s.n) return nullptr;
  if ((long long)s.m * s.k >= (1ll << 28)) return nullptr;
  std::vector<unsigned int> ptr((size_t)s.m + 1), idx((size_t)s.m * s.k), vmap((size_t)s.m * s.k);
  for (int m = 0; m <= s.m; ++m) ptr[m] = (unsigned int)((long long)m * s.k);
  for (int m = 0; m < s.m; ++m) for (int k = 0; k < s.k; ++k) { idx[(size_t)m * s.k + k] = (unsigned int)k; vmap[(size_t)m * s.k + k] = (unsigned int)((long long)m * s.lda + k); }
  KernelCtx* c = new_unregistered(K_SPMM_ASPARSE, d); if (!c) return nullptr;
  c->packed_width = packed_width; c->sp_ncols = s.n; c->sp_skip_empty = 0;
  if (!upload_pattern(c, s.m, s.k, ptr.data(), idx.data(), vmap.data())) { drop_unregistered(c); return nullptr; }
  attach_jit(c, ptr.data(), idx.data(), vmap.data());
  c->nflops = (unsigned int)(2ull * s.m * s.n * s.k * packed_width);
  return (libxsmm_gemmfunction)handle_for_slot(c->slot);
}

LIBXSMM_API libxsmm_gemmfunction libxsmm_create_spgemm_csr_areg(libxsmm_gemm_shape s, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_blasint max_N, const unsigned int* row_ptr, const unsigned int* column_idx, const double* values) {
  // row-major C[M x max_N] (+)= A_csr * B with A's values fixed at creation (always passed as double)
  // [ref: libxsmm_main.c:3842-3883; SURVEY Appendix B.6].  No limit on the number of unique values here.
  if (!runtime_ready()) return nullptr;
  if (!row_ptr || !column_idx || !values || s.a_in_type != s.b_in_type) return nullptr;
  if ((s.a_in_type != LIBXSMM_DATATYPE_F32 && s.a_in_type != LIBXSMM_DATATYPE_F64) || s.out_type != s.a_in_type) return nullptr;
  if (s.lda != 0 || max_N <= 0 || s.ldb < max_N || s.ldc < max_N) return nullptr;
  libxsmm_descriptor_blob blob;
  libxsmm_gemm_descriptor* d = libxsmm_gemm_descriptor_init(&blob, s.a_in_type, s.b_in_type, s.comp_type, s.out_type, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
    (int)(flags | LIBXSMM_GEMM_FLAG_USE_XGEMM_ABI), (int)prefetch);
  if (!d) return nullptr;
  KernelCtx* c = new_unregistered(K_SPMM_ASPARSE, d); if (!c) return nullptr;
  c->packed_width = 1; c->sp_ncols = max_N; c->sp_skip_empty = 0;
  bool ok = upload_pattern(c, s.m, s.k, row_ptr, column_idx, nullptr);
  if (ok) {   // values arrive as double whatever the kernel's type: convert once, here
    if (s.a_in_type == LIBXSMM_DATATYPE_F32) { std::vector<float> v32(values, values + row_ptr[s.m]); c->d_vals = to_device(v32.data(), v32.size()); }
    else c->d_vals = to_device(values, row_ptr[s.m]);
    ok = c->d_vals != nullptr;
  }
  if (!ok) { drop_unregistered(c); return nullptr; }
  attach_jit(c, row_ptr, column_idx, nullptr);
  c->nflops = (unsigned int)(2ull * row_ptr[s.m] * max_N);
  return (libxsmm_gemmfunction)handle_for_slot(c->slot);
}

// ---- lifetime / introspection --------------------------------------------------------------------------------------
LIBXSMM_API void libxsmm_release_kernel(const void* kernel) {
  KernelCtx* c = ctx_from_handle(kernel);
  if (!c) return;
  if (c->registered) { vlog(1, "libxsmm_release_kernel: registered kernels are owned by the registry (no-op)"); return; }   // [ref: libxsmm_main.c:3900-3922]
  coalesce_flush();
  (void)hipDeviceSynchronize();
  drop_unregistered(c);
}
LIBXSMM_API int libxsmm_get_kernel_info(const void* kernel, libxsmm_kernel_info* info) {
  KernelCtx* c = ctx_from_handle(kernel);
  if (!c || !info) return EXIT_FAILURE;
  std::memset(info, 0, sizeof(*info));
  info->kind = !c->registered ? LIBXSMM_KERNEL_UNREGISTERED : (c->kind == K_MELTW ? LIBXSMM_KERNEL_KIND_MELTW : LIBXSMM_KERNEL_KIND_MATMUL);
  info->nflops = c->nflops; info->code_size = 0; info->is_reference_kernel = 0;
  return EXIT_SUCCESS;
}
LIBXSMM_API int libxsmm_get_mmkernel_info(libxsmm_xmmfunction kernel, libxsmm_mmkernel_info* info) {
  KernelCtx* c = ctx_from_handle(kernel.ptr_const);
  if (!c || !info || c->kind == K_MELTW) return EXIT_FAILURE;
  // the COMMON precision of A and B with the signedness dropped, UNSUPPORTED when they differ beyond that [ref: src/libxsmm_main.c:3057, LIBXSMM_GEMM_GETENUM_AB_COMMON_PREC] --
  // round 5: found by the differential test (was: A's type as given)
  const auto sgn = [](int t) { return t == LIBXSMM_DATATYPE_U64 ? LIBXSMM_DATATYPE_I64 : t == LIBXSMM_DATATYPE_U32 ? LIBXSMM_DATATYPE_I32 : t == LIBXSMM_DATATYPE_U16 ? LIBXSMM_DATATYPE_I16 :
                                  t == LIBXSMM_DATATYPE_U8 ? LIBXSMM_DATATYPE_I8 : t == LIBXSMM_DATATYPE_U4X2 ? LIBXSMM_DATATYPE_I4X2 : t; };
  info->iprecision = (libxsmm_datatype)(sgn(c->g.a_type) == sgn(c->g.b_type) ? sgn(c->g.a_type) : (int)LIBXSMM_DATATYPE_UNSUPPORTED); info->oprecision = (libxsmm_datatype)c->g.c_type;
  info->prefetch = (libxsmm_gemm_prefetch_type)c->g.prefetch; info->flags = (int)c->g.flags;
  info->lda = c->g.lda; info->ldb = c->g.ldb; info->ldc = c->g.ldc; info->m = c->g.m; info->n = c->g.n; info->k = c->g.k;
  return EXIT_SUCCESS;
}
LIBXSMM_API int libxsmm_get_meltwkernel_info(libxsmm_xmeltwfunction kernel, libxsmm_meltwkernel_info* info) {
  KernelCtx* c = ctx_from_handle((const void*)kernel.xmeltw);
  if (!c || !info || c->kind != K_MELTW) return EXIT_FAILURE;
  info->ldi = c->e.ldi; info->ldo = c->e.ldo; info->m = c->e.m; info->n = c->e.n; info->datatype = c->e.in0_type; info->flags = c->e.flags;
  info->operation = c->e.operation;          // the KIND (unary 1 / binary 2 / ternary 3), as the reference reports it [ref: src/libxsmm_main.c:3106] -- round 5: found by the differential test (was: the TPP type)
  return EXIT_SUCCESS;
}
LIBXSMM_API int libxsmm_get_registry_info(libxsmm_registry_info* info) {
  if (!info) return EXIT_FAILURE;
  std::lock_guard<std::mutex> guard(g_lock);
  std::memset(info, 0, sizeof(*info));
  info->capacity = (size_t)g_registered_limit; info->size = g_registry.size(); info->nbytes = g_registry.size() * sizeof(KernelCtx);
  return EXIT_SUCCESS;
}

// ---- libxsmm_hip.h ---------------------------------------------------------------------------------------------------
LIBXSMM_API int libxsmm_hip_device_count(void) { if (libxsmm_ninit < 2) libxsmm_init(); return g_device_count > 0 ? g_device_count : 0; }
LIBXSMM_API void libxsmm_hip_set_jit(int mode) { g_jit_mode = (mode < 0 || mode > 2) ? 1 : mode; }
LIBXSMM_API int libxsmm_hip_get_jit(void) { return jit_mode(); }
LIBXSMM_API int libxsmm_hip_available(void) { return libxsmm_hip_device_count() > 0 ? 1 : 0; }
LIBXSMM_API int libxsmm_hip_set_device(int device) {
  coalesce_flush();                     // queued calls belong to the device they were issued for
  if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return -1;
  tls().device = device; return 0;
}
LIBXSMM_API int libxsmm_hip_get_device(void) { int d = 0; if (hipGetDevice(&d) != hipSuccess) return -1; return d; }
LIBXSMM_API void libxsmm_hip_set_stream(void* s) { coalesce_flush(); if (tls().pipe_lanes > 1) libxsmm_hip_pipeline_end(); tls().stream = s; if (tls().async != 2) tls().async = 1; }
LIBXSMM_API void* libxsmm_hip_get_stream(void) { return tls().stream; }
LIBXSMM_API void libxsmm_hip_set_async(int enable) { coalesce_flush(); tls().async = enable == 2 ? 2 : (enable ? 1 : 0); }
LIBXSMM_API void libxsmm_hip_set_streaming_hint(int mode) { tls().stream_hint = (mode >= 0 && mode <= 2) ? mode : 0; }
LIBXSMM_API int libxsmm_hip_get_streaming_hint(void) { return tls().stream_hint; }
LIBXSMM_API int libxsmm_hip_streaming_window_verdict(void) { return rt_window_verdict(); }
LIBXSMM_API int libxsmm_hip_get_async(void) { return tls().async; }
LIBXSMM_API void libxsmm_hip_sync(void) {
  coalesce_flush();
  if (tls().pipe_lanes > 1) libxsmm_hip_pipeline_end();          // a synchronisation closes an open pipeline section
  (void)hip_ok(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
}
// Pipeline sections: see include/libxsmm_hip.h.  Fork = an event on the caller's stream that every lane waits for; join = one event per lane that
// the caller's stream waits for.  All of it is stream-ordered (no host synchronisation), so a section can be captured into a hipGraph: the lanes
// become parallel branches of the graph.
LIBXSMM_API int libxsmm_hip_bcsc_bind_pattern(libxsmm_gemmfunction kernel, const unsigned int* colptr, const unsigned int* rowidx, unsigned long long n_block_columns) {
  KernelCtx* c = ctx_from_handle((const void*)kernel);
  if (!c || c->kind != K_BCSC) { set_error(-3, "libxsmm_hip_bcsc_bind_pattern: not a packed BCSC kernel"); return EXIT_FAILURE; }
  std::lock_guard<std::mutex> guard(g_lock);
  if (c->bcsc_bound.d_table) { retire_block(c->bcsc_bound.d_table); c->bcsc_bound = KernelCtx::BcscBound(); }       // a launch in flight may still read the old table
  if (!colptr || !rowidx) return EXIT_SUCCESS;                                                                       // unbind
  const int nkb = (c->bk > 0 && (int)c->g.k % c->bk == 0) ? (int)c->g.k / c->bk : 0;
  if (nkb <= 0 || n_block_columns == 0 || n_block_columns >= (1ull << 31)) { set_error(-3, "libxsmm_hip_bcsc_bind_pattern: this kernel's block shape has no inverted table"); return EXIT_FAILURE; }
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, colptr) != hipSuccess || attr.type != hipMemoryTypeDevice || hipPointerGetAttributes(&attr, rowidx) != hipSuccess || attr.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    set_error(-3, "libxsmm_hip_bcsc_bind_pattern takes DEVICE arrays (host-resident patterns are recognised and cached by the call itself)");
    return EXIT_FAILURE;
  }
  unsigned int* table = nullptr;
  if (!hip_ok(hipMalloc((void**)&table, (size_t)n_block_columns * (size_t)nkb * sizeof(unsigned int)), "hipMalloc(BCSC table)")) return EXIT_FAILURE;
  const int err = launch_bcsc_invert(colptr, rowidx, table, (int)n_block_columns, nkb, tls().stream);       // stream-ordered before the calls that follow on this stream
  if (err != 0) { (void)hipFree(table); set_error(err, "launch of bcsc_invert_kernel failed: %s", hipGetErrorString((hipError_t)err)); return EXIT_FAILURE; }
  if (!tls().async) (void)hipStreamSynchronize(cur_stream());
  // The caller promises not to change the arrays: they are read once here (the number of blocks and the k-blocks the first 64 columns use, what a host-resident
  // pattern tells the launcher), unless the bind is being captured into a graph.  This waits for the stream, like the first call with a new host pattern does.
  int nnzb_read = 0; unsigned long long kmask_read = 0ull;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (tls().stream && hipStreamIsCapturing(cur_stream(), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }      // (the NULL stream cannot be captured and is not asked)
  if (cap == hipStreamCaptureStatusNone && nkb <= 64 && c->bn > 0 && n_block_columns <= 4096) {
    std::vector<unsigned int> hc((size_t)n_block_columns + 1);
    if (hipMemcpyAsync(hc.data(), colptr, hc.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, cur_stream()) == hipSuccess && hipStreamSynchronize(cur_stream()) == hipSuccess &&
        hc[n_block_columns] > 0 && hc[n_block_columns] <= (1u << 20)) {
      std::vector<unsigned int> hr(hc[n_block_columns]);
      if (hipMemcpyAsync(hr.data(), rowidx, hr.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, cur_stream()) == hipSuccess && hipStreamSynchronize(cur_stream()) == hipSuccess) {
        bool sane = true;
        for (unsigned long long nb = 0; nb < n_block_columns && sane; ++nb) sane = hc[nb] <= hc[nb + 1] && hc[nb + 1] <= hc[n_block_columns];
        for (unsigned int b = 0; b < hc[n_block_columns] && sane; ++b) sane = hr[b] < (unsigned int)nkb;
        if (sane) {
          nnzb_read = (int)hc[n_block_columns];
          for (unsigned long long nb = 0; nb < n_block_columns && nb * (unsigned long long)c->bn < 64ull; ++nb)
            for (unsigned int b = hc[nb]; b < hc[nb + 1]; ++b) kmask_read |= 1ull << hr[b];
        }
      }
    }
    (void)hipGetLastError();
  }
  c->bcsc_bound.nnzb = nnzb_read; c->bcsc_bound.kmask0 = kmask_read;
  c->bcsc_bound.colptr = colptr; c->bcsc_bound.rowidx = rowidx; c->bcsc_bound.nblk_n = n_block_columns; c->bcsc_bound.d_table = table; c->bcsc_bound.nkb = nkb;
  c->device = cur_device();
  return EXIT_SUCCESS;
}
LIBXSMM_API int libxsmm_hip_pipeline_begin(int lanes) {
  coalesce_flush();                     // calls queued before the section leave BEFORE the fork event: every lane is ordered behind them
  ThreadState& t = tls();
  if (!runtime_ready() || g_dryrun) return EXIT_FAILURE;
  if (t.pipe_lanes > 1) { set_error(-3, "libxsmm_hip_pipeline_begin: a pipeline section is already open on this thread"); return EXIT_FAILURE; }
  if (!t.async) { set_error(-3, "libxsmm_hip_pipeline_begin needs stream-ordered launches (libxsmm_hip_set_stream / libxsmm_hip_set_async)"); return EXIT_FAILURE; }
  lanes = std::max(1, std::min(lanes, 8));
  if (lanes == 1) return EXIT_SUCCESS;
  if (t.pipe_device != cur_device()) {           // (re)create the lanes of this thread on the current device
    for (int i = 0; i < 8; ++i) if (t.pipe_stream[i]) { (void)hipStreamDestroy((hipStream_t)t.pipe_stream[i]); t.pipe_stream[i] = nullptr; }
    for (int i = 0; i < 9; ++i) if (t.pipe_event[i]) { (void)hipEventDestroy((hipEvent_t)t.pipe_event[i]); t.pipe_event[i] = nullptr; }
    t.pipe_device = cur_device();
  }
  for (int i = 0; i < lanes; ++i) {
    if (!t.pipe_stream[i] && !hip_ok(hipStreamCreateWithFlags((hipStream_t*)&t.pipe_stream[i], hipStreamNonBlocking), "hipStreamCreate(pipeline lane)")) return EXIT_FAILURE;
    if (!t.pipe_event[i] && !hip_ok(hipEventCreateWithFlags((hipEvent_t*)&t.pipe_event[i], hipEventDisableTiming), "hipEventCreate(pipeline lane)")) return EXIT_FAILURE;
  }
  if (!t.pipe_event[8] && !hip_ok(hipEventCreateWithFlags((hipEvent_t*)&t.pipe_event[8], hipEventDisableTiming), "hipEventCreate(pipeline fork)")) return EXIT_FAILURE;
  t.pipe_user = t.stream;
  if (!hip_ok(hipEventRecord((hipEvent_t)t.pipe_event[8], (hipStream_t)t.pipe_user), "hipEventRecord(pipeline fork)")) return EXIT_FAILURE;
  for (int i = 0; i < lanes; ++i)
    if (!hip_ok(hipStreamWaitEvent((hipStream_t)t.pipe_stream[i], (hipEvent_t)t.pipe_event[8], 0), "hipStreamWaitEvent(pipeline fork)")) return EXIT_FAILURE;
  // every lane gets a partial-result workspace as large as the thread's own BEFORE the section opens: a first hipMalloc inside the section would be
  // illegal while a graph is being captured (warm-up launches outside the section size lane 0)
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)t.pipe_user, &cap_status) != hipSuccess) { (void)hipGetLastError(); cap_status = hipStreamCaptureStatusNone; }
  for (int i = 1; i < lanes && cap_status == hipStreamCaptureStatusNone; ++i) {        // (an allocation under capture would invalidate the capture: open a section once outside it to size the lanes)
    Workspace& w = t_workspace[i];
    if (w.base && w.device != cur_device()) { retire_block(w.base); w.base = nullptr; w.cap = 0; }
    if (w.cap < t_workspace[0].cap) {
      void* nb = nullptr;
      if (hipMalloc(&nb, t_workspace[0].cap) == hipSuccess) { if (w.base) retire_block(w.base); w.base = nb; w.cap = t_workspace[0].cap; w.device = cur_device(); }
      else (void)hipGetLastError();          // (e.g. under capture: the section still works for kernels that need no workspace)
    }
  }
  t.pipe_lanes = lanes; t.pipe_cur = 0; t.stream = t.pipe_stream[0];
  return EXIT_SUCCESS;
}
LIBXSMM_API int libxsmm_hip_pipeline_end(void) {
  ThreadState& t = tls();
  if (t.pipe_lanes <= 1) return EXIT_SUCCESS;
  const int lanes = t.pipe_lanes;
  t.pipe_lanes = 0; t.pipe_cur = 0; t.stream = t.pipe_user;
  bool ok = true;
  for (int i = 0; i < lanes; ++i) {
    ok = hip_ok(hipEventRecord((hipEvent_t)t.pipe_event[i], (hipStream_t)t.pipe_stream[i]), "hipEventRecord(pipeline join)") && ok;
    ok = hip_ok(hipStreamWaitEvent((hipStream_t)t.pipe_user, (hipEvent_t)t.pipe_event[i], 0), "hipStreamWaitEvent(pipeline join)") && ok;
  }
  return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
LIBXSMM_API int libxsmm_hip_get_last_error(void) { return tls().last_error; }
LIBXSMM_API const char* libxsmm_hip_get_last_error_string(void) { return tls().last_error_msg.c_str(); }
LIBXSMM_API void libxsmm_hip_clear_last_error(void) { tls().last_error = 0; tls().last_error_msg.clear(); }
LIBXSMM_API void* libxsmm_hip_malloc(size_t n) { void* p = nullptr; if (!hip_ok(hipMalloc(&p, n ? n : 1), "hipMalloc")) return nullptr; return p; }
LIBXSMM_API void libxsmm_hip_free(void* p) { coalesce_flush(); if (p) (void)hipFree(p); }
// Blocking copies, ORDERED ON THE CALLING THREAD'S STREAM: queued (coalescing mode) and stream-ordered launches issued before the copy have run when it
// reads or overwrites their operands -- a copy on the legacy default stream would not wait for a non-blocking user stream.
LIBXSMM_API int libxsmm_hip_memcpy_h2d(void* d, const void* s, size_t n) {
  coalesce_flush();
  if (!hip_ok(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, cur_stream()), "hipMemcpy(H2D)")) return -1;
  return hip_ok(hipStreamSynchronize(cur_stream()), "hipMemcpy(H2D)") ? 0 : -1;
}
LIBXSMM_API int libxsmm_hip_memcpy_d2h(void* d, const void* s, size_t n) {
  coalesce_flush();
  if (!hip_ok(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, cur_stream()), "hipMemcpy(D2H)")) return -1;
  return hip_ok(hipStreamSynchronize(cur_stream()), "hipMemcpy(D2H)") ? 0 : -1;
}
LIBXSMM_API int libxsmm_hip_memset(void* d, int v, size_t n) {
  coalesce_flush();
  if (!hip_ok(hipMemsetAsync(d, v, n, cur_stream()), "hipMemset")) return -1;
  return hip_ok(hipStreamSynchronize(cur_stream()), "hipMemset") ? 0 : -1;
}
LIBXSMM_API int libxsmm_hip_probe_mfma(libxsmm_datatype datatype, const void* operands, int iterations, double* flop) {
  if (!runtime_ready() || g_dryrun || !operands || iterations <= 0 || (datatype != LIBXSMM_DATATYPE_BF16 && datatype != LIBXSMM_DATATYPE_F32)) return EXIT_FAILURE;
  const int err = launch_mfma_probe(datatype == LIBXSMM_DATATYPE_BF16 ? 1 : 0, operands, iterations, tls().stream, flop);
  if (err != 0) { set_error(err, "launch of mfma_probe_kernel failed: %s", hipGetErrorString((hipError_t)err)); return EXIT_FAILURE; }
  return EXIT_SUCCESS;
}
LIBXSMM_API unsigned long long libxsmm_hip_launch_count(int reset) { const unsigned long long n = tls().launches; if (reset) tls().launches = 0; return n; }
LIBXSMM_API const char* libxsmm_hip_kernel_name(const void* kernel, int batched) {
  KernelCtx* c = ctx_from_handle(kernel);
  return c ? (batched ? c->kname_batched : c->kname_single) : "";
}

static KernelCtx* batch_ctx(const void* fn, Kind want) {
  KernelCtx* c = ctx_from_handle(fn);
  if (!c) { set_error(-3, "batched launch through an unknown kernel handle"); return nullptr; }
  if (c->kind != want) { set_error(-3, "batched launch: handle is not of the expected kind"); return nullptr; }
  return c;
}
LIBXSMM_API void libxsmm_hip_gemm_batch_strided(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* param, size_t count, long long sa, long long sb, long long sc) {
  KernelCtx* c = ctx_from_handle((const void*)kernel);
  if (!c) { set_error(-3, "batched launch through an unknown kernel handle"); return; }
  if (!param || count == 0) return;
  BatchSpec b; b.count = count; b.s[0] = sa; b.s[1] = sb; b.s[2] = sc;
  if (c->kind == K_SPMM_ASPARSE || c->kind == K_SPMM_BSPARSE) { run_spmm(c, param, b); return; }     // packed sparse kernels: the element loop
  if (c->kind != K_GEMM) { set_error(-3, "batched launch: handle is not a (BR)GEMM or packed sparse kernel"); return; }
  if (c->g.flags & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI) { set_error(-3, "use libxsmm_hip_gemm_ext_batch_strided for ext kernels"); return; }
  run_gemm(c, param, b);
}
LIBXSMM_API void libxsmm_hip_gemm_ext_batch_strided(libxsmm_gemmfunction_ext kernel, const libxsmm_gemm_ext_param* param, size_t count,
  long long sa, long long sb, long long sc, long long sd, long long smask) {
  KernelCtx* c = batch_ctx((const void*)kernel, K_GEMM); if (!c || !param || count == 0) return;
  if (!(c->g.flags & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI)) { set_error(-3, "handle was not dispatched with libxsmm_dispatch_brgemm_ext"); return; }
  BatchSpec b; b.count = count; b.s[0] = sa; b.s[1] = sb; b.s[2] = sc; b.s[3] = sd; b.s[4] = smask;
  run_gemm(c, param, b);
}
LIBXSMM_API void libxsmm_hip_gemm_batch_strided_2d(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* param, size_t count_i, size_t count_j,
  long long stride_a_i, long long stride_b_j, long long stride_c_i, long long stride_c_j) {
  KernelCtx* c = batch_ctx((const void*)kernel, K_GEMM); if (!c || !param || count_i == 0 || count_j == 0) return;
  if (c->g.flags & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI) { set_error(-3, "use libxsmm_hip_gemm_ext_batch_strided_2d for ext kernels"); return; }
  if (c->g.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS) { set_error(-3, "2-D batches take STRIDE / OFFSET / plain kernels (a pointer list per element has no 2-D form)"); return; }
  if (count_i * count_j >= (1ull << 31)) { set_error(-3, "2-D batch too large"); return; }
  BatchSpec b; b.count = count_i * count_j; b.inner = count_i; b.s[0] = stride_a_i; b.s[1] = stride_b_j; b.s[2] = stride_c_i; b.c2 = stride_c_j;
  run_gemm(c, param, b);
}
LIBXSMM_API void libxsmm_hip_gemm_ext_batch_strided_2d(libxsmm_gemmfunction_ext kernel, const libxsmm_gemm_ext_param* param, size_t count_i, size_t count_j,
  long long stride_a_i, long long stride_b_j, long long stride_c_i, long long stride_c_j, long long stride_d_i, long long stride_mask_i, long long stride_mask_j) {
  KernelCtx* c = batch_ctx((const void*)kernel, K_GEMM); if (!c || !param || count_i == 0 || count_j == 0) return;
  if (!(c->g.flags & LIBXSMM_GEMM_FLAG_USE_XGEMM_EXT_ABI)) { set_error(-3, "handle was not dispatched with libxsmm_dispatch_brgemm_ext"); return; }
  if (c->g.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE_ADDRESS) { set_error(-3, "2-D batches take STRIDE / OFFSET / plain kernels (a pointer list per element has no 2-D form)"); return; }
  if (count_i * count_j >= (1ull << 31)) { set_error(-3, "2-D batch too large"); return; }
  BatchSpec b; b.count = count_i * count_j; b.inner = count_i; b.s[0] = stride_a_i; b.s[1] = stride_b_j; b.s[2] = stride_c_i; b.c2 = stride_c_j;
  b.s[3] = stride_d_i; b.s[4] = stride_mask_i; b.mask2 = stride_mask_j;
  run_gemm(c, param, b);
}
LIBXSMM_API void libxsmm_hip_gemm_batch_pointers(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* param, size_t count,
  const void* const* a_list, const void* const* b_list, void* const* c_list) {
  KernelCtx* c = batch_ctx((const void*)kernel, K_GEMM); if (!c || !param || count == 0) return;
  if (!a_list || !b_list || !c_list) { set_error(-2, "pointer-list batch with a NULL list"); return; }
  BatchSpec b; b.count = count; b.la = a_list; b.lb = b_list; b.lc = c_list;
  run_gemm(c, param, b);
}
LIBXSMM_API void libxsmm_hip_meltw_unary_batch_strided(libxsmm_meltwfunction_unary kernel, const libxsmm_meltw_unary_param* param, size_t count,
  long long s_in, long long s_out, long long s_aux) {
  KernelCtx* c = batch_ctx((const void*)kernel, K_MELTW); if (!c || !param || count == 0) return;
  BatchSpec b; b.count = count; b.s[0] = s_in; b.s[1] = s_out; b.s[2] = s_aux; run_meltw(c, param, b);
}
LIBXSMM_API void libxsmm_hip_meltw_binary_batch_strided(libxsmm_meltwfunction_binary kernel, const libxsmm_meltw_binary_param* param, size_t count,
  long long s0, long long s1, long long so) {
  KernelCtx* c = batch_ctx((const void*)kernel, K_MELTW); if (!c || !param || count == 0) return;
  BatchSpec b; b.count = count; b.s[0] = s0; b.s[1] = s1; b.s[2] = so; run_meltw(c, param, b);
}
LIBXSMM_API void libxsmm_hip_meltw_ternary_batch_strided(libxsmm_meltwfunction_ternary kernel, const libxsmm_meltw_ternary_param* param, size_t count,
  long long s0, long long s1, long long s2, long long so) {
  KernelCtx* c = batch_ctx((const void*)kernel, K_MELTW); if (!c || !param || count == 0) return;
  BatchSpec b; b.count = count; b.s[0] = s0; b.s[1] = s1; b.s[2] = s2; b.s[3] = so; run_meltw(c, param, b);
}
// ---- multi-device launch from ONE host thread (the reference's scale-out axis is the caller's loop,
// samples/xgemm/gemm_kernel.c:4063-4066) --------------------------------------------------------------------------------------------------------
// Every shard has a context of its own on the calling thread -- device, a non-blocking stream there, staging scratch, partial-result workspaces -- that is
// swapped into the thread's state around the shard's launch, so the existing launch paths run unchanged and shards on one device (virtual shards: the
// one-GPU test) overlap without sharing a byte of scratch.
namespace {
struct ShardCtx {
  int device = -1; hipStream_t stream = nullptr; hipEvent_t done = nullptr;
  Scratch scratch; Workspace ws[8];
  ~ShardCtx() { if (stream) (void)hipStreamDestroy(stream); if (done) (void)hipEventDestroy(done); }
};
struct ShardSet { std::vector<ShardCtx*> ctx; hipEvent_t fork = nullptr; int fork_device = -1;
  ~ShardSet() { for (ShardCtx* c : ctx) delete c; if (fork) (void)hipEventDestroy(fork); } };
thread_local ShardSet t_shards;
// field-wise: a std::swap of the structs would run a destructor on the temporary and retire a live block
void swap_scratch(Scratch& a, Scratch& b) { std::swap(a.base, b.base); std::swap(a.cap, b.cap); std::swap(a.used, b.used); std::swap(a.device, b.device); }
void swap_workspace(Workspace& a, Workspace& b) { std::swap(a.base, b.base); std::swap(a.cap, b.cap); std::swap(a.device, b.device); }
void shard_swap(ShardCtx& sc) { swap_scratch(t_scratch, sc.scratch); for (int i = 0; i < 8; ++i) swap_workspace(t_workspace[i], sc.ws[i]); }
ShardCtx* shard_ctx(int index, int device) {
  ShardSet& set = t_shards;
  if ((size_t)index >= set.ctx.size()) set.ctx.resize((size_t)index + 1, nullptr);
  ShardCtx*& c = set.ctx[(size_t)index];
  if (c && c->device != device) { delete c; c = nullptr; }          // the slot moved to another device: its stream belongs to the old one
  if (!c) {
    c = new ShardCtx(); c->device = device;
    if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate(shard)") ||
        !hip_ok(hipEventCreateWithFlags(&c->done, hipEventDisableTiming), "hipEventCreate(shard)")) { delete c; c = nullptr; }
  }
  return c;
}
std::mutex g_peer_lock; std::vector<char> g_peer_enabled;     // [from * ndev + to]
void enable_peer(int from, int to) {       // `from` is the current device
  if (from == to || g_device_count <= 1) return;
  std::lock_guard<std::mutex> guard(g_peer_lock);
  if (g_peer_enabled.empty()) g_peer_enabled.assign((size_t)g_device_count * (size_t)g_device_count, 0);
  char& done = g_peer_enabled[(size_t)from * (size_t)g_device_count + (size_t)to];
  if (done) return;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can) { const hipError_t e = hipDeviceEnablePeerAccess(to, 0); if (e != hipSuccess) (void)hipGetLastError(); }
  else (void)hipGetLastError();          // no direct path: hipMemcpyPeerAsync stages through the host by itself
  done = 1;
}
}  // namespace
LIBXSMM_API int libxsmm_hip_launch_shards(const libxsmm_hip_shard* shards, int nshards, int gather_device, void* gather_dst) {
  coalesce_flush();
  if (!runtime_ready() || g_dryrun || g_device_count <= 0) { set_error(-4, "libxsmm_hip_launch_shards: no HIP device"); return EXIT_FAILURE; }
  if (!shards || nshards <= 0 || nshards > 64) { set_error(-2, "libxsmm_hip_launch_shards: 1..64 shards"); return EXIT_FAILURE; }
  ThreadState& t = tls();
  if (t.pipe_lanes > 1 || t_nest > 0) { set_error(-3, "libxsmm_hip_launch_shards: not inside a pipeline section or a kernel invocation"); return EXIT_FAILURE; }
  for (int i = 0; i < nshards; ++i) {
    const libxsmm_hip_shard& sh = shards[i];
    if (sh.device < 0 || sh.device >= g_device_count) { set_error(-2, "libxsmm_hip_launch_shards: shard %d names device %d, %d visible", i, sh.device, g_device_count); return EXIT_FAILURE; }
    if (!sh.kernel || !sh.param) { set_error(-2, "libxsmm_hip_launch_shards: shard %d has no kernel / param", i); return EXIT_FAILURE; }
    if (!ctx_from_handle(sh.kernel)) { set_error(-3, "libxsmm_hip_launch_shards: shard %d: unknown kernel handle", i); return EXIT_FAILURE; }
    if (sh.gather_bytes && (!gather_dst || !sh.gather_src || gather_device < 0 || gather_device >= g_device_count)) {
      set_error(-2, "libxsmm_hip_launch_shards: shard %d asks for a gather without source / destination / root device", i); return EXIT_FAILURE; }
    if (sh.gather_bytes && sh.gather_rows > 1 && (sh.gather_src_pitch < sh.gather_bytes || sh.gather_dst_pitch < sh.gather_bytes)) {
      set_error(-2, "libxsmm_hip_launch_shards: shard %d: a pitched gather needs pitches of at least one row (gather_bytes)", i); return EXIT_FAILURE; }
  }
  const int home_device = t.device, home_async = t.async; void* const home_stream = t.stream;
  int hip_home = 0; (void)hipGetDevice(&hip_home);
  const int err0 = t.last_error; const std::string msg0 = t.last_error_msg;      // a stale error of the thread must not hide a shard's failure of the same code: cleared for the loop, restored if no shard failed
  t.last_error = 0;
  bool ok = true;
  // fork: work already issued to the thread's stream (stream-ordered mode) comes first on every shard
  ShardSet& set = t_shards;
  const bool fork = home_async != 0;
  if (fork) {
    if (set.fork && set.fork_device != hip_home) { (void)hipEventDestroy(set.fork); set.fork = nullptr; }
    if (!set.fork) { ok = hip_ok(hipEventCreateWithFlags(&set.fork, hipEventDisableTiming), "hipEventCreate(shard fork)"); set.fork_device = hip_home; }
    ok = ok && hip_ok(hipEventRecord(set.fork, (hipStream_t)home_stream), "hipEventRecord(shard fork)");
  }
  int issued = 0;
  for (int i = 0; i < nshards && ok; ++i) {
    const libxsmm_hip_shard& sh = shards[i];
    if (!hip_ok(hipSetDevice(sh.device), "hipSetDevice(shard)")) { ok = false; break; }
    ShardCtx* sc = shard_ctx(i, sh.device);
    if (!sc) { ok = false; break; }
    if (fork && !hip_ok(hipStreamWaitEvent(sc->stream, set.fork, 0), "hipStreamWaitEvent(shard fork)")) { ok = false; break; }
    t.device = sh.device; t.stream = sc->stream; t.async = 1;       // the shard's launches are stream-ordered on its own stream
    shard_swap(*sc);
    KernelCtx* k = ctx_from_handle(sh.kernel);
    if (sh.count == 0) run_any(k, sh.param, BatchSpec{});
    else {
      BatchSpec b; b.count = sh.count; for (int j = 0; j < 5; ++j) b.s[j] = sh.stride[j];
      if (k->kind == K_GEMM || k->kind == K_MELTW || k->kind == K_SPMM_ASPARSE || k->kind == K_SPMM_BSPARSE) run_any(k, sh.param, b);
      else set_error(-3, "libxsmm_hip_launch_shards: shard %d: this kind of kernel has no strided batch (count must be 0)", i);
    }
    shard_swap(*sc);
    if (t.last_error != 0) ok = false;
    if (ok && sh.gather_bytes) {
      char* dst = (char*)gather_dst + sh.gather_dst_offset;
      if (sh.gather_rows > 1) {        // pitched (round 6): the shard's column block of a row-major result, in place
        enable_peer(sh.device, gather_device);
        ok = hip_ok(hipMemcpy2DAsync(dst, sh.gather_dst_pitch, sh.gather_src, sh.gather_src_pitch, sh.gather_bytes, sh.gather_rows, hipMemcpyDeviceToDevice, sc->stream), "hipMemcpy2DAsync(shard gather)");
      }
      else if (sh.device == gather_device) ok = hip_ok(hipMemcpyAsync(dst, sh.gather_src, sh.gather_bytes, hipMemcpyDeviceToDevice, sc->stream), "hipMemcpyAsync(shard gather)");
      else { enable_peer(sh.device, gather_device);      // each source pushes over its own xGMI link into the root: no ring, up to nshards - 1 links at once
             ok = hip_ok(hipMemcpyPeerAsync(dst, gather_device, sh.gather_src, sh.device, sh.gather_bytes, sc->stream), "hipMemcpyPeerAsync(shard gather)"); }
    }
    ok = hip_ok(hipEventRecord(sc->done, sc->stream), "hipEventRecord(shard)") && ok;
    issued = i + 1;
  }
  (void)hipSetDevice(hip_home);
  t.device = home_device; t.stream = home_stream; t.async = home_async;
  // join: stream-ordered callers get the shards' completion as a dependency of their stream, blocking callers get results
  for (int i = 0; i < issued; ++i) {
    ShardCtx* sc = set.ctx[(size_t)i];
    if (fork) ok = hip_ok(hipStreamWaitEvent((hipStream_t)home_stream, sc->done, 0), "hipStreamWaitEvent(shard join)") && ok;
    else { const hipError_t e = hipEventSynchronize(sc->done); if (e != hipSuccess) { set_error((int)e, "shard %d faulted: %s", i, hipGetErrorString(e)); ok = false; } }
  }
  if (ok && t.last_error == 0) { t.last_error = err0; t.last_error_msg = msg0; }
  return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
static int batch_sharded(const void* kernel, const char* params, size_t param_size, size_t count, const long long* strides, int nstrides, int c_slot_offset,
                         int nshards, const int* devices, int gather_device, void* gather_dst) {
  if (nshards <= 0 || nshards > 64 || !params) { set_error(-2, "sharded batch: 1..64 shards and one param per shard"); return EXIT_FAILURE; }
  const int ndev = libxsmm_hip_device_count();
  if (ndev <= 0) { set_error(-4, "sharded batch: no HIP device"); return EXIT_FAILURE; }
  libxsmm_hip_shard sh[64];
  int n = 0;
  for (int s = 0; s < nshards; ++s) {
    size_t b = 0, e = 0;
    libxsmm_hip_shard_range(count, 1, nshards, s, &b, &e);
    if (e == b) continue;                                           // more shards than problems
    libxsmm_hip_shard& x = sh[n++];
    std::memset(&x, 0, sizeof(x));
    x.device = devices ? devices[s] : s % ndev;
    x.kernel = kernel; x.param = params + param_size * (size_t)s; x.count = e - b;
    for (int j = 0; j < nstrides; ++j) x.stride[j] = strides[j];
    if (gather_dst) {
      const long long sc = strides[2];
      if (sc <= 0) { set_error(-2, "sharded batch: a gather needs a positive C stride"); return EXIT_FAILURE; }
      x.gather_src = *(void* const*)((const char*)x.param + c_slot_offset); x.gather_bytes = (e - b) * (size_t)sc; x.gather_dst_offset = b * (size_t)sc;
    }
  }
  return n == 0 ? EXIT_SUCCESS : libxsmm_hip_launch_shards(sh, n, gather_device, gather_dst);
}
LIBXSMM_API int libxsmm_hip_gemm_batch_strided_sharded(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* shard_params, size_t count,
  long long stride_a, long long stride_b, long long stride_c, int nshards, const int* devices, int gather_device, void* gather_dst) {
  const long long st[3] = {stride_a, stride_b, stride_c};
  return batch_sharded((const void*)kernel, (const char*)shard_params, sizeof(libxsmm_gemm_param), count, st, 3, (int)offsetof(libxsmm_gemm_param, c), nshards, devices, gather_device, gather_dst);
}
LIBXSMM_API int libxsmm_hip_gemm_ext_batch_strided_sharded(libxsmm_gemmfunction_ext kernel, const libxsmm_gemm_ext_param* shard_params, size_t count,
  long long stride_a, long long stride_b, long long stride_c, long long stride_d, long long stride_mask, int nshards, const int* devices, int gather_device, void* gather_dst) {
  const long long st[5] = {stride_a, stride_b, stride_c, stride_d, stride_mask};
  return batch_sharded((const void*)kernel, (const char*)shard_params, sizeof(libxsmm_gemm_ext_param), count, st, 5, (int)offsetof(libxsmm_gemm_ext_param, c), nshards, devices, gather_device, gather_dst);
}
// ---- result gather without a collective library: the root pulls every shard over its own xGMI link ----------------------------
namespace { struct IpcBlob { hipIpcMemHandle_t h; unsigned long long offset, size; }; }
LIBXSMM_API int libxsmm_hip_ipc_export(const void* device_ptr, void* handle) {
  static_assert(sizeof(IpcBlob) == LIBXSMM_HIP_IPC_HANDLE_BYTES, "IPC blob size differs from the C ABI's LIBXSMM_HIP_IPC_HANDLE_BYTES");
  if (!device_ptr || !handle) return -1;
  IpcBlob b; std::memset(&b, 0, sizeof(b));
  // an IPC handle names a whole ALLOCATION; device_ptr may sit inside one (sub-allocating memory pools): carry the offset along
  hipDeviceptr_t base = nullptr; size_t size = 0;
  if (!hip_ok(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)const_cast<void*>(device_ptr)), "hipMemGetAddressRange")) return -1;
  if (!hip_ok(hipIpcGetMemHandle(&b.h, (void*)base), "hipIpcGetMemHandle")) return -1;
  b.offset = (unsigned long long)((const char*)device_ptr - (const char*)base); b.size = (unsigned long long)size;
  std::memcpy(handle, &b, sizeof(b));
  return 0;
}
LIBXSMM_API int libxsmm_hip_gather_shards(void* dst, int world, int self_rank, const void* handles, const void* self_src,
  const size_t* src_offsets, const size_t* dst_offsets, const size_t* nbytes) {
  if (!dst || world <= 0 || !dst_offsets || !nbytes) return -1;
  std::vector<void*> opened((size_t)world, nullptr);
  std::vector<hipStream_t> streams((size_t)world, nullptr);
  int rc = 0;
  for (int r = 0; r < world && rc == 0; ++r) {
    if (nbytes[r] == 0) continue;
    const char* src = nullptr;
    if (r == self_rank) src = (const char*)self_src;
    else {
      if (!handles) { set_error(-2, "libxsmm_hip_gather_shards: no IPC handle for rank %d", r); rc = -1; break; }
      IpcBlob b; std::memcpy(&b, (const char*)handles + sizeof(IpcBlob) * (size_t)r, sizeof(b));
      if (!hip_ok(hipIpcOpenMemHandle(&opened[r], b.h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) { rc = -1; break; }
      if (b.offset + (src_offsets ? src_offsets[r] : 0) + nbytes[r] > b.size) { set_error(-2, "libxsmm_hip_gather_shards: shard of rank %d exceeds the exported allocation", r); rc = -1; break; }
      src = (const char*)opened[r] + b.offset;
    }
    if (!src) { set_error(-2, "libxsmm_hip_gather_shards: NULL source for rank %d", r); rc = -1; break; }
    if (!hip_ok(hipStreamCreateWithFlags(&streams[r], hipStreamNonBlocking), "hipStreamCreate")) { rc = -1; break; }
    // one copy per source, each on its own stream: the sources sit behind different xGMI links of the root, so the copies run
    // concurrently at up to (world - 1) x one link instead of a ring's one link per hop
    if (!hip_ok(hipMemcpyAsync((char*)dst + dst_offsets[r], src + (src_offsets ? src_offsets[r] : 0), nbytes[r], hipMemcpyDeviceToDevice, streams[r]), "hipMemcpyAsync(peer shard)")) rc = -1;
  }
  for (int r = 0; r < world; ++r) {
    if (streams[r]) { if (hipStreamSynchronize(streams[r]) != hipSuccess) rc = -1; (void)hipStreamDestroy(streams[r]); }
    if (opened[r]) (void)hipIpcCloseMemHandle(opened[r]);
  }
  return rc;
}
LIBXSMM_API void libxsmm_hip_shard_range(size_t count, size_t granule, int world, int rank, size_t* begin, size_t* end) {
  if (granule == 0) granule = 1;
  if (world <= 0) world = 1;
  if (rank < 0) rank = 0;
  const size_t units = (count + granule - 1) / granule;           // shard whole granules
  const size_t base = units / (size_t)world, extra = units % (size_t)world;
  const size_t r = (size_t)rank;
  const size_t ub = r * base + std::min(r, extra), ue = ub + base + (r < extra ? 1 : 0);
  if (begin) *begin = std::min(ub * granule, count);
  if (end) *end = std::min(ue * granule, count);
}

}  // extern "C"
