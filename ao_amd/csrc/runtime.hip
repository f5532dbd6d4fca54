// Error plumbing and version of the C-ABI library (include/ao_mi355.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "splitk.h"

#include <mutex>
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
// One workspace (fp32 / int32 parts + tickets) per (device, stream), allocated on the first split-K launch on that stream.
// Launches on ONE stream are ordered, and the last arriver of a launch leaves every ticket at zero, so consecutive launches
// on a stream share their workspace safely; launches on different streams never share one.  (Round 1 rotated 4
// process-wide slots with no tie to streams: more than 4 split-K kernels in flight across streams aliased a slot.)
namespace {
struct SplitWs {
  hipStream_t stream = nullptr;
  bool used = false;
  float* part = nullptr;
  unsigned* tickets = nullptr;
};
std::mutex g_split_mu;
SplitWs g_split_ws[64][kSplitSlots];
}  // namespace

int splitk_workspace(hipStream_t stream, float** part, unsigned** tickets) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_failed(e, "hipGetDevice");
  AO_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_split_mu);
  SplitWs* w = nullptr;
  for (int i = 0; i < kSplitSlots && w == nullptr; ++i)
    if (g_split_ws[dev][i].used && g_split_ws[dev][i].stream == stream) w = &g_split_ws[dev][i];
  for (int i = 0; i < kSplitSlots && w == nullptr; ++i)
    if (!g_split_ws[dev][i].used) w = &g_split_ws[dev][i];
  AO_REQUIRE(w != nullptr, "split-K kernels were launched on more than %d streams of device %d: each stream owns a %zu MB workspace; "
             "reuse streams (a captured graph keeps using the workspace of the stream it was captured on)", kSplitSlots, dev,
             (kSplitSlotFloats * sizeof(float)) >> 20);
  if (w->part == nullptr) {
    const size_t bytes = kSplitSlotFloats * sizeof(float) + kSplitMaxTickets * sizeof(unsigned);
    char* p = nullptr;
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
      return hip_failed(e, "hipMalloc(split-K workspace); call the op with this shape once on this stream outside stream capture "
                           "before capturing it into a graph");
    unsigned* t = reinterpret_cast<unsigned*>(p + kSplitSlotFloats * sizeof(float));
    e = hipMemset(t, 0, kSplitMaxTickets * sizeof(unsigned));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return hip_failed(e, "hipMemset(split-K tickets)"); }
    w->part = reinterpret_cast<float*>(p);
    w->tickets = t;
  }
  w->used = true;
  w->stream = stream;
  *part = w->part;
  *tickets = w->tickets;
  return AO_OK;
}
}  // namespace ao

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
