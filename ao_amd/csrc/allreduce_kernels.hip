// One-shot SUM all-reduce over peer-mapped buffers for decode-size partials (SURVEY.md 5 last row, 7 step 6, 8(e)).
//
// torchao has no collective of its own: the row-parallel linears of the TP benchmark (ao_amd/parallel.py; reference harness
// torchao/testing/utils.py:370-467) end in an all-reduce the CALLER issues.  On an 8 x MI355X node every GPU reaches every
// other over its own xGMI link (7 x ~153 GB/s, point to point), so for the 16 - 64 KiB partials of a decode step a ring (14 hops)
// or any multi-kernel collective is pure latency.  One launch per rank instead:
//   1. copy the rank's vector into ITS staging buffer (device memory every peer has mapped through IPC), release it at system scope;
//   2. raise flag[rank] = epoch in EVERY peer's flag block (one 4-byte store over each link);
//   3. wait until every peer has raised its flag in the local block (bounded spin: a rank that never shows up turns into a status
//      code, not a hang);
//   4. read all `world` staged vectors -- own from HBM, peers' over xGMI, with system-scope loads that bypass the non-coherent
//      caches -- and add them IN RANK ORDER in fp32 (int32: exact), so every rank computes bit-identical sums.
// Staging and flags are double-buffered by epoch parity: a rank can only enter epoch e + 1's wait after finishing epoch e's reads,
// so passing that wait proves every peer is done with the parity-e buffers before epoch e + 2 overwrites them.
#include <algorithm>

#include "common.h"
#include "peer_sync.h"

namespace ao {
namespace {

constexpr int kMaxWorld = 8;
constexpr int kArBlocks = 16;            // EVERY call launches this many blocks (slices and epochs line up across calls of any size)

struct ArArgs {
  char* data[kMaxWorld];      // staging of every rank (index = rank), 2 x slot_bytes each (parity-major)
  unsigned* flags[kMaxWorld];  // flag block of every rank: [2 parities][kArBlocks][kMaxWorld]
  const void* in;
  void* out;
  unsigned* state;            // local, never shared: [0] status (1 = a wait timed out), [1 + b] epoch of block b's last call
  long long count;            // elements
  long long slot_bytes;
  unsigned long long timeout_ticks;  // 100 MHz ticks a wait may take
  int rank, world;
};

// DT: 0 fp32, 1 bf16, 2 int32;  MAX: elementwise maximum instead of the sum (the amax exchange of the exact row-parallel protocol)
template <int DT, bool MAX>
__global__ __launch_bounds__(256) void allreduce_oneshot_kernel(ArArgs a) {
  __shared__ int s_late;
  const int tid = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  // the epoch lives on the device (one counter per block, bumped by every call; every call launches all kArBlocks blocks, so the
  // counters move in lockstep = a call counter): a launch captured into a hipGraph replays with fresh epochs and alternating parity.
  // Every rank makes the same sequence of calls, so the counters agree across ranks.
  const unsigned epoch = a.state[1 + b] + 1u;
  const int par = epoch & 1;
  constexpr int ES = (DT == 1) ? 2 : 4;
  const long long nvec = (a.count * ES + 15) / 16;                 // 16-byte units (the buffers are padded to 16 bytes)
  const long long v0 = nvec * b / nb, v1 = nvec * (b + 1) / nb;    // this block's slice
  char* mine = a.data[a.rank] + (long long)par * a.slot_bytes;
  // 1. stage
  for (long long i = v0 + tid; i < v1; i += 256)
    *reinterpret_cast<u32x4*>(mine + i * 16) = *reinterpret_cast<const u32x4*>(static_cast<const char*>(a.in) + i * 16);
  __threadfence_system();  // the slice is in memory, visible to every agent, before any flag says so
  __syncthreads();
  // 2. signal every rank (own block included), 3. wait for every rank
  if (tid == 0) s_late = 0;
  __syncthreads();
  if (tid < a.world) {
    unsigned* f = a.flags[tid] + ((size_t)par * kArBlocks + b) * kMaxWorld + a.rank;
    __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* w = a.flags[a.rank] + ((size_t)par * kArBlocks + b) * kMaxWorld + tid;
    if (!wait_flag(w, epoch, a.timeout_ticks)) { atomicExch(a.state, 1u); s_late = 1; }
  }
  __syncthreads();
  if (tid == 0) a.state[1 + b] = epoch;  // (every thread read the old value before the barriers above)
  if (s_late) {
    // a peer did not arrive in time: its staging holds an older call's bytes.  Poison this slice (NaN / INT_MIN) instead of returning a
    // plausible wrong sum; state[0] = 1 says why (OneShotAllReduce.check() raises on it).
    const u32x4 bad = (DT == 0) ? u32x4{0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u}
                                : (DT == 1) ? u32x4{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u} : u32x4{0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    for (long long i = v0 + tid; i < v1; i += 256) *reinterpret_cast<u32x4*>(static_cast<char*>(a.out) + i * 16) = bad;
    return;
  }
  // 4. reduce in rank order
  auto comb = [](f32x4 x, f32x4 y) {
    if constexpr (MAX) return f32x4{fmaxf(x.x, y.x), fmaxf(x.y, y.y), fmaxf(x.z, y.z), fmaxf(x.w, y.w)};
    else return x + y;
  };
  for (long long i = v0 + tid; i < v1; i += 256) {
    f32x4 s0, s1;
    i32x4 si;
    for (int r = 0; r < a.world; ++r) {
      const u32x4 v = ld_sys16(a.data[r] + (long long)par * a.slot_bytes + i * 16);
      if constexpr (DT == 0) {
        const f32x4 t = __builtin_bit_cast(f32x4, v);
        s0 = (r == 0) ? t : comb(s0, t);
      } else if constexpr (DT == 1) {
        const f32x4 t0 = {bf16_lo_to_f32(v.x), bf16_hi_to_f32(v.x), bf16_lo_to_f32(v.y), bf16_hi_to_f32(v.y)};
        const f32x4 t1 = {bf16_lo_to_f32(v.z), bf16_hi_to_f32(v.z), bf16_lo_to_f32(v.w), bf16_hi_to_f32(v.w)};
        s0 = (r == 0) ? t0 : comb(s0, t0);
        s1 = (r == 0) ? t1 : comb(s1, t1);
      } else {
        const i32x4 t = __builtin_bit_cast(i32x4, v);
        if constexpr (MAX) si = (r == 0) ? t : i32x4{max(si.x, t.x), max(si.y, t.y), max(si.z, t.z), max(si.w, t.w)};
        else si = (r == 0) ? t : si + t;
      }
    }
    u32x4 o;
    if constexpr (DT == 0) o = __builtin_bit_cast(u32x4, s0);
    else if constexpr (DT == 1) o = u32x4{pack_bf16x2(s0.x, s0.y), pack_bf16x2(s0.z, s0.w), pack_bf16x2(s1.x, s1.y), pack_bf16x2(s1.z, s1.w)};
    else o = __builtin_bit_cast(u32x4, si);
    *reinterpret_cast<u32x4*>(static_cast<char*>(a.out) + i * 16) = o;
  }
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int64_t ao_allreduce_flag_bytes(void) { return (int64_t)2 * kArBlocks * kMaxWorld * sizeof(unsigned); }
extern "C" int64_t ao_allreduce_state_bytes(void) { return (int64_t)(1 + kArBlocks) * sizeof(unsigned); }

extern "C" int ao_allreduce_oneshot_op(void* const* peer_data_host, void* const* peer_flags_host, const void* input, void* output, void* local_state,
                                       int64_t count, int dtype, int op, int64_t slot_bytes, int rank, int world, void* stream) {
  AO_REQUIRE_PTR(peer_data_host);
  AO_REQUIRE_PTR(peer_flags_host);
  AO_REQUIRE_PTR(local_state);
  AO_REQUIRE(world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world, "ao_allreduce_oneshot: bad rank %d / world %d (at most %d ranks)", rank, world,
             kMaxWorld);
  AO_REQUIRE(dtype >= 0 && dtype <= 2, "ao_allreduce_oneshot: dtype must be 0 (fp32), 1 (bf16) or 2 (int32), got %d", dtype);
  AO_REQUIRE(op == 0 || op == 1, "ao_allreduce_oneshot: op must be 0 (SUM) or 1 (MAX), got %d", op);
  const int64_t bytes = count * (dtype == 1 ? 2 : 4);
  AO_REQUIRE(count >= 0 && ((bytes + 15) / 16) * 16 <= slot_bytes, "ao_allreduce_oneshot: %lld bytes do not fit the %lld-byte staging slot", (long long)bytes,
             (long long)slot_bytes);
  AO_REQUIRE(bytes % 16 == 0, "ao_allreduce_oneshot: the vector must be a multiple of 16 bytes, got %lld", (long long)bytes);
  if (count == 0) return AO_OK;
  AO_REQUIRE_PTR(input);
  AO_REQUIRE_PTR(output);
  AO_REQUIRE(((uintptr_t)input % 16 == 0) && ((uintptr_t)output % 16 == 0) && slot_bytes % 16 == 0, "ao_allreduce_oneshot: 16-byte aligned buffers expected");
  ArArgs a{};
  for (int r = 0; r < world; ++r) {
    AO_REQUIRE(peer_data_host[r] != nullptr && peer_flags_host[r] != nullptr, "ao_allreduce_oneshot: rank %d's buffers are not mapped", r);
    a.data[r] = static_cast<char*>(peer_data_host[r]);
    a.flags[r] = static_cast<unsigned*>(peer_flags_host[r]);
  }
  a.in = input; a.out = output; a.state = static_cast<unsigned*>(local_state);
  a.count = count; a.slot_bytes = slot_bytes; a.rank = rank; a.world = world;
  a.timeout_ticks = collective_timeout_ticks();
  const dim3 grid(kArBlocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype * 2 + op) {
    case 0: ao::launch(allreduce_oneshot_kernel<0, false>, grid, block, 0, st, a); break;
    case 1: ao::launch(allreduce_oneshot_kernel<0, true>, grid, block, 0, st, a); break;
    case 2: ao::launch(allreduce_oneshot_kernel<1, false>, grid, block, 0, st, a); break;
    case 3: ao::launch(allreduce_oneshot_kernel<1, true>, grid, block, 0, st, a); break;
    case 4: ao::launch(allreduce_oneshot_kernel<2, false>, grid, block, 0, st, a); break;
    default: ao::launch(allreduce_oneshot_kernel<2, true>, grid, block, 0, st, a); break;
  }
  AO_LAUNCH_CHECK("allreduce_oneshot_kernel launch");
  return AO_OK;
}

extern "C" int ao_allreduce_oneshot(void* const* peer_data_host, void* const* peer_flags_host, const void* input, void* output, void* local_state,
                                    int64_t count, int dtype, int64_t slot_bytes, int rank, int world, void* stream) {
  return ao_allreduce_oneshot_op(peer_data_host, peer_flags_host, input, output, local_state, count, dtype, 0, slot_bytes, rank, world, stream);
}
