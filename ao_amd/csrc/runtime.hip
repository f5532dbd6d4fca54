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

// ---- split-K workspace (splitk.h): the library's only device-side state ----------------------------
namespace {
struct SplitWs {
  float* part = nullptr;
  unsigned* tickets = nullptr;
  unsigned next_slot = 0;
};
std::mutex g_split_mu;
SplitWs g_split_ws[64];
}  // namespace

int splitk_workspace(float** part, unsigned** tickets) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_failed(e, "hipGetDevice");
  AO_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_split_mu);
  SplitWs& w = g_split_ws[dev];
  if (w.part == nullptr) {
    const size_t bytes = kSplitSlots * (kSplitSlotFloats * sizeof(float) + kSplitMaxTiles * sizeof(unsigned));
    char* p = nullptr;
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
      return hip_failed(e, "hipMalloc(split-K workspace); call the op with this shape once outside stream capture before "
                           "capturing it into a graph");
    unsigned* t = reinterpret_cast<unsigned*>(p + kSplitSlots * kSplitSlotFloats * sizeof(float));
    e = hipMemset(t, 0, kSplitSlots * kSplitMaxTiles * sizeof(unsigned));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return hip_failed(e, "hipMemset(split-K tickets)"); }
    w.part = reinterpret_cast<float*>(p);
    w.tickets = t;
  }
  const unsigned slot = w.next_slot++ % kSplitSlots;
  *part = w.part + slot * kSplitSlotFloats;
  *tickets = w.tickets + slot * kSplitMaxTiles;
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
