/* guard_alloc.c -- test helper (not product): device buffers whose first or last byte touches UNMAPPED address space, through HIP's virtual-memory API.
 * A kernel that loads or stores one element outside such an operand takes a GPU page fault (the process aborts) instead of silently reading the
 * allocator's slack -- which is all that torch / hipMalloc memory can show (round-3 advisor note, round-4 review item 7).
 * Layout of one reservation: [ one granule unmapped | mapped, rounded up to granules | one granule unmapped ].
 *   guard_alloc(nbytes, 0): the returned block ENDS at the end of the mapped part (an over-read faults); its start is kept 16-byte aligned (what the
 *     library's kernels test operands for, so the guarded run selects the same kernels as the plain one): up to 15 bytes of slack when nbytes % 16 != 0;
 *   guard_alloc(nbytes, 1): it STARTS at the start of the mapped part (an under-read faults).
 * Built by the test with:  gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include guard_alloc.c -L/opt/rocm/lib -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

typedef struct guard_rec { void* user; char* base; size_t reserve, mapped; hipMemGenericAllocationHandle_t handle; struct guard_rec* next; } guard_rec;
static guard_rec* g_head = NULL;

size_t guard_granularity(void) {
  hipMemAllocationProp prop; size_t gran = 0; int dev = 0;
  memset(&prop, 0, sizeof(prop));
  (void)hipGetDevice(&dev);
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return 0;
  return gran;
}

void* guard_alloc(size_t nbytes, int front) {
  hipMemAllocationProp prop; hipMemAccessDesc acc; guard_rec* r; size_t gran; int dev = 0; void* base = NULL;
  if (nbytes == 0) nbytes = 1;
  gran = guard_granularity();
  if (gran == 0) return NULL;
  r = (guard_rec*)calloc(1, sizeof(*r));
  if (!r) return NULL;
  (void)hipGetDevice(&dev);
  memset(&prop, 0, sizeof(prop)); memset(&acc, 0, sizeof(acc));
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  r->mapped = (nbytes + gran - 1) / gran * gran; r->reserve = r->mapped + 2 * gran;
  if (hipMemAddressReserve(&base, r->reserve, gran, NULL, 0) != hipSuccess) { free(r); return NULL; }
  r->base = (char*)base;
  if (hipMemCreate(&r->handle, r->mapped, &prop, 0) != hipSuccess) { (void)hipMemAddressFree(base, r->reserve); free(r); return NULL; }
  if (hipMemMap(r->base + gran, r->mapped, 0, r->handle, 0) != hipSuccess || hipMemSetAccess(r->base + gran, r->mapped, &acc, 1) != hipSuccess) {
    (void)hipMemRelease(r->handle); (void)hipMemAddressFree(base, r->reserve); free(r); return NULL; }
  r->user = front ? (void*)(r->base + gran) : (void*)(r->base + gran + r->mapped - ((nbytes + 15) & ~(size_t)15));
  r->next = g_head; g_head = r;
  return r->user;
}

/* guard_free QUARANTINES by default (guard_set_reuse(1) really unmaps): in the first guarded run of round 5, where freed blocks were unmapped and released at once, the
 * SECOND and later tests of a process read wrong C values (beta = 1 cases: what the kernel saw of a freshly uploaded C was not what libxsmm_hip_memcpy_h2d had
 * written) although nothing faulted; with freed blocks kept mapped until the process exits every guarded parity test passes.  tools/guard_probe.py (copy kernel, 20
 * alloc / upload / free rounds with reuse on) did NOT reproduce it -- profiles/r05_guard_probe.json -- so the cause stays open (candidates: a stale translation or
 * cache line of a virtual range that is unmapped and re-mapped to other physical pages within microseconds); what matters for the parity tests is that the memory
 * holds what was uploaded, and quarantined blocks do (a 4 KiB granule each on this stack). */
static int g_reuse = 0;
void guard_set_reuse(int on) { g_reuse = on; }
void guard_free(void* user) {
  guard_rec** pp = &g_head;
  if (!g_reuse) return;
  for (; *pp; pp = &(*pp)->next) if ((*pp)->user == user) {
    guard_rec* r = *pp; size_t gran = (r->reserve - r->mapped) / 2;
    *pp = r->next;
    (void)hipMemUnmap(r->base + gran, r->mapped); (void)hipMemRelease(r->handle); (void)hipMemAddressFree(r->base, r->reserve);
    free(r); return;
  }
}
