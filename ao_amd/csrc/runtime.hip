// Error plumbing and version of the C-ABI library (include/ao_mi355.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

#include <algorithm>
#include "splitk.h"

#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace ao {
namespace {
thread_local char g_err[512] = "";
// profiling state (single-threaded use from bench/tools)
std::vector<hipEvent_t> g_prof_events;  // 2 per record
int g_prof_capacity = 0;                // records
int g_prof_count = 0;
bool g_prof_on = false;
}

bool prof_next_events(hipEvent_t* start, hipEvent_t* stop) {
  if (!g_prof_on || g_prof_count >= g_prof_capacity) return false;
  *start = g_prof_events[2 * g_prof_count];
  *stop = g_prof_events[2 * g_prof_count + 1];
  ++g_prof_count;
  return true;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_failed(hipError_t e, const char* what) {
  set_error("%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
  return AO_ERR_HIP;
}

// ---- dynamic-LDS opt-in per (kernel, device) -------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember which (kernel, device) pairs
// have it, under a mutex (the launchers used to keep one unsynchronised process-wide flag per template).
namespace {
std::mutex g_attr_mu;
struct AttrDone { const void* kernel; int dev; size_t bytes; };
std::vector<AttrDone> g_attr_done;
}  // namespace

int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what) {
  if (bytes <= 48 * 1024) return AO_OK;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_failed(e, "hipGetDevice");
  std::lock_guard<std::mutex> lock(g_attr_mu);
  AttrDone* slot = nullptr;
  for (auto& kd : g_attr_done)
    if (kd.kernel == kernel && kd.dev == dev) slot = &kd;
  if (slot != nullptr && slot->bytes >= bytes) return AO_OK;  // the attribute is a maximum: only ever raised
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return hip_failed(e, what);
  if (slot != nullptr) slot->bytes = bytes; else g_attr_done.push_back(AttrDone{kernel, dev, bytes});
  return AO_OK;
}

// ---- split-K workspace (splitk.h): the library's only device-side state ----------------------------------------------
// One workspace (fp32 / int32 parts + tickets) per (device, stream), allocated on the first split-K launch on that stream and
// sized to what the launches on it have asked for so far (power-of-two steps from 8 MiB up to the kSplitSlotFloats cap; growing
// re-allocates after a device synchronise).  Launches on ONE stream are ordered, and the last arriver of a launch leaves every
// ticket at zero, so consecutive launches on a stream share their workspace safely; launches on different streams never share
// one.  At most kSplitSlots streams per device hold a workspace at a time: a further stream EVICTS the least recently used slot
// (device synchronise first -- nothing of the evicted stream's launches may still be using the memory -- then the slot, memory
// and all, changes owner).  Allocation, growth and eviction cannot happen inside stream capture (hipMalloc / synchronise are
// illegal there): run the op once on the stream, outside capture, with the largest shape first.  A graph captured on a stream
// whose slot is later evicted must be re-captured (its launches would share the buffer with the new owner).
namespace {
struct SplitWs {
  hipStream_t stream = nullptr;
  bool used = false;
  char* base = nullptr;       // [floats fp32][kSplitMaxTickets unsigned]
  size_t floats = 0;          // capacity of the parts area
  unsigned long long last_use = 0;
};
std::mutex g_split_mu;
SplitWs g_split_ws[64][kSplitSlots];
unsigned long long g_split_clock = 0;
std::vector<char*> g_split_retired;  // outgrown buffers, kept alive for graphs captured before the growth
constexpr size_t kSplitMinFloats = (size_t)2 << 20;  // 8 MiB

int split_alloc(SplitWs* w, size_t floats, int dev) {
  char* p = nullptr;
  hipError_t e = hipMalloc(&p, floats * sizeof(float) + kSplitMaxTickets * sizeof(unsigned));
  if (e != hipSuccess)
    return hip_failed(e, "hipMalloc(split-K workspace); call the op with this shape once on this stream outside stream capture "
                         "before capturing it into a graph");
  e = hipMemset(p + floats * sizeof(float), 0, kSplitMaxTickets * sizeof(unsigned));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(p); return hip_failed(e, "hipMemset(split-K tickets)"); }
  w->base = p;
  w->floats = floats;
  return AO_OK;
}
}  // namespace

int splitk_workspace(hipStream_t stream, float** part, unsigned** tickets, size_t need_floats, int parts) {
  AO_REQUIRE(parts >= 1 && parts <= kSplitMaxParts, "split-K: %d K parts (the ticket encoding and the two-level meeting take 1 .. %d)", parts, kSplitMaxParts);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_failed(e, "hipGetDevice");
  AO_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  AO_REQUIRE(need_floats <= kSplitSlotFloats, "split-K workspace request of %zu floats exceeds the %zu-float cap", need_floats, kSplitSlotFloats);
  std::lock_guard<std::mutex> lock(g_split_mu);
  SplitWs* w = nullptr;
  for (int i = 0; i < kSplitSlots && w == nullptr; ++i)
    if (g_split_ws[dev][i].used && g_split_ws[dev][i].stream == stream) w = &g_split_ws[dev][i];
  for (int i = 0; i < kSplitSlots && w == nullptr; ++i)
    if (!g_split_ws[dev][i].used) w = &g_split_ws[dev][i];
  if (w == nullptr) {
    // every slot has an owner: the least recently used one changes hands once the device is idle
    e = hipDeviceSynchronize();
    if (e != hipSuccess)
      return hip_failed(e, "hipDeviceSynchronize (evicting a split-K workspace: more than kSplitSlots streams of this device run split-K "
                           "kernels; not possible inside stream capture)");
    w = &g_split_ws[dev][0];
    for (int i = 1; i < kSplitSlots; ++i)
      if (g_split_ws[dev][i].last_use < w->last_use) w = &g_split_ws[dev][i];
  }
  size_t want = kSplitMinFloats;
  while (want < need_floats) want <<= 1;
  if (want > kSplitSlotFloats) want = kSplitSlotFloats;
  if (w->base == nullptr) {
    if (int rc = split_alloc(w, want, dev)) return rc;
  } else if (w->floats < need_floats) {
    e = hipDeviceSynchronize();  // earlier launches on this stream may still be meeting in the old buffer
    if (e != hipSuccess) return hip_failed(e, "hipDeviceSynchronize (growing the split-K workspace; run the largest shape once outside stream capture)");
    // the outgrown buffer is RETIRED, not freed: a hipGraph captured on this stream earlier keeps launching into it (each launch
    // is self-contained -- it leaves its tickets at zero), so freeing it would pull memory from under such a graph.  At most
    // log2(cap / 8 MiB) retirements per slot, < the cap in total.
    g_split_retired.push_back(w->base);
    w->base = nullptr;
    w->floats = 0;
    if (int rc = split_alloc(w, want, dev)) return rc;
  }
  w->used = true;
  w->stream = stream;
  w->last_use = ++g_split_clock;
  *part = reinterpret_cast<float*>(w->base);
  *tickets = reinterpret_cast<unsigned*>(w->base + w->floats * sizeof(float));
  return AO_OK;
}

}  // namespace ao

// Pre-size the calling stream's split-K workspace (outside stream capture), so that every later launch on that stream -- whatever its
// shape -- is capture-safe without a warm-up call.  bytes <= 0: the cap (128 MiB).
extern "C" int ao_splitk_reserve(void* stream, int64_t bytes) {
  using namespace ao;
  size_t floats = bytes <= 0 ? kSplitSlotFloats : std::min<size_t>(kSplitSlotFloats, ((size_t)bytes + 3) / 4);
  float* part = nullptr;
  unsigned* tickets = nullptr;
  return splitk_workspace((hipStream_t)stream, &part, &tickets, floats, 1);
}

namespace ao {
// ---- collectives: wait bound shared by the one-shot all-reduce and the all-to-all-v (peer_sync.h) --------------------------------
namespace {
std::atomic<int> g_collective_timeout_ms{5000};
}
unsigned long long collective_timeout_ticks() { return (unsigned long long)g_collective_timeout_ms.load() * 100000ull; }  // 100 MHz

}  // namespace ao

extern "C" int ao_collective_set_timeout_ms(int ms) {
  using namespace ao;
  AO_REQUIRE(ms >= 1 && ms <= 600000, "ao_collective_set_timeout_ms: %d ms outside [1, 600000]", ms);
  g_collective_timeout_ms.store(ms);
  return AO_OK;
}
extern "C" int ao_collective_timeout_ms(void) { return ao::g_collective_timeout_ms.load(); }

// ---- peer-visible device memory (flags / staging of the hand-written collectives) -------------------------------------------------
extern "C" int ao_peer_alloc(void** ptr_out_host, int64_t bytes, int kind) {
  using namespace ao;
  AO_REQUIRE_PTR(ptr_out_host);
  AO_REQUIRE(bytes > 0 && bytes < (1ll << 40), "ao_peer_alloc: bad size %lld", (long long)bytes);
  AO_REQUIRE(kind == 0 || kind == 1, "ao_peer_alloc: kind must be 0 (uncached: flags) or 1 (fine-grained: staging), got %d", kind);
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, kind == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
  if (e != hipSuccess) return hip_failed(e, kind == 0 ? "hipExtMallocWithFlags(hipDeviceMallocUncached)" : "hipExtMallocWithFlags(hipDeviceMallocFinegrained)");
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(p); return hip_failed(e, "hipMemset(peer buffer)"); }
  *ptr_out_host = p;
  return AO_OK;
}
extern "C" int ao_peer_free(void* ptr) {
  using namespace ao;
  if (ptr == nullptr) return AO_OK;
  hipError_t e = hipFree(ptr);
  if (e != hipSuccess) return hip_failed(e, "hipFree(peer buffer)");
  return AO_OK;
}
extern "C" int ao_peer_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }
extern "C" int ao_peer_export(void* ptr, void* handle_out_host) {
  using namespace ao;
  AO_REQUIRE_PTR(ptr);
  AO_REQUIRE_PTR(handle_out_host);
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) return hip_failed(e, "hipIpcGetMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on dmabuf-only hosts)");
  memcpy(handle_out_host, &h, sizeof(h));
  return AO_OK;
}
extern "C" int ao_peer_import(const void* handle_host, void** ptr_out_host) {
  using namespace ao;
  AO_REQUIRE_PTR(handle_host);
  AO_REQUIRE_PTR(ptr_out_host);
  hipIpcMemHandle_t h;
  memcpy(&h, handle_host, sizeof(h));
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return hip_failed(e, "hipIpcOpenMemHandle");
  *ptr_out_host = p;
  return AO_OK;
}
extern "C" int ao_peer_close(void* imported_ptr) {
  using namespace ao;
  if (imported_ptr == nullptr) return AO_OK;
  hipError_t e = hipIpcCloseMemHandle(imported_ptr);
  if (e != hipSuccess) return hip_failed(e, "hipIpcCloseMemHandle");
  return AO_OK;
}

extern "C" int ao_abi_version(void) { return AO_MI355_ABI_VERSION; }
extern "C" const char* ao_last_error(void) { return ao::g_err; }

extern "C" int ao_prof_enable(int max_records) {
  using namespace ao;
  AO_REQUIRE(max_records > 0 && max_records <= (1 << 20), "ao_prof_enable: bad max_records=%d", max_records);
  while ((int)g_prof_events.size() < 2 * max_records) {
    hipEvent_t e;
    hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) return hip_failed(err, "hipEventCreate");
    g_prof_events.push_back(e);
  }
  g_prof_capacity = max_records;
  g_prof_count = 0;
  g_prof_on = true;
  return AO_OK;
}

extern "C" int ao_prof_collect(float* ms_out_host, int capacity, int* n_out_host) {
  using namespace ao;
  AO_REQUIRE_PTR(ms_out_host);
  AO_REQUIRE_PTR(n_out_host);
  g_prof_on = false;
  const int n = g_prof_count < capacity ? g_prof_count : capacity;
  for (int i = 0; i < n; ++i) {
    hipError_t err = hipEventSynchronize(g_prof_events[2 * i + 1]);
    if (err != hipSuccess) return hip_failed(err, "hipEventSynchronize");
    err = hipEventElapsedTime(&ms_out_host[i], g_prof_events[2 * i], g_prof_events[2 * i + 1]);
    if (err != hipSuccess) return hip_failed(err, "hipEventElapsedTime");
  }
  *n_out_host = n;
  g_prof_count = 0;
  return AO_OK;
}
