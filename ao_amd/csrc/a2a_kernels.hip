// On-device all-to-all-v of MXFP8 token rows over peer-mapped buffers (reference
// torchao/prototype/moe_training/kernels/mxfp8/comms.py:25-400, _mxfp8_all_to_all_v_kernel :318-402, _exchange_row_offsets :405-460).
//
// An expert-parallel dispatch through `all_to_all_single` needs the split sizes on the HOST (one device-to-host sync per layer).
// The reference avoids it with a Triton kernel over symmetric memory: every rank keeps its e4m3 rows, its E8M0 scale rows and its
// int64 split vector in buffers all peers have mapped, and each rank PULLS what is addressed to it -- offsets computed on the device
// from the peers' split vectors.  Same algorithm here, for one xGMI-connected node (every GPU reaches every other over its own link):
//   barrier (everyone's inputs are staged) | block (remote, part): read remote's splits, in_off = sum splits[remote][< rank],
//   out_off = sum_{q < remote} splits[q][rank], copy rows [in_off, in_off + n) of remote's data / scale buffers to local rows
//   [out_off, ..) with system-scope 16-byte loads | barrier (everyone is done reading: the staging may be overwritten).
// The barriers are the one-shot all-reduce's (allreduce_kernels.hip): one flag word per (phase, block, rank) in every peer, epochs
// counted on the device (so a captured launch replays), bounded spin -> a status bit instead of a hang.
#include <algorithm>

#include "common.h"
#include "peer_sync.h"

namespace ao {
namespace {

constexpr int kA2AMaxWorld = 8;
constexpr int kA2ABlocksPerRank = 16;                         // reference: BLOCKS_PER_REMOTE_RANK = 32 of 16 Ki elements
constexpr int kA2AMaxBlocks = kA2AMaxWorld * kA2ABlocksPerRank;

struct A2AArgs {
  const char* data[kA2AMaxWorld];        // every rank's staged rows      [max_rows][row_bytes]
  const char* scales[kA2AMaxWorld];      // every rank's staged scale rows [max_rows][scale_row_bytes]
  const long long* splits[kA2AMaxWorld];  // every rank's split vector      [world] (rows it sends to rank r)
  unsigned* flags[kA2AMaxWorld];         // every rank's flag block        [2 phases][kA2AMaxBlocks][kA2AMaxWorld]
  char* out_data;
  char* out_scales;
  long long* out_splits;                 // [world]: rows received from rank r
  unsigned* state;                       // local: [0] status bits (1 a wait timed out, 2 more rows than the output holds, 4 a peer's splits
                                         //        reach past its staged rows), [1 + b] epoch
  long long row_bytes, scale_row_bytes, max_out_rows, max_in_rows;
  unsigned long long timeout_ticks;
  int rank, world;
};

__global__ __launch_bounds__(256) void moe_a2a_v_kernel(A2AArgs a) {
  __shared__ long long s_remote[kA2AMaxWorld], s_to_me[kA2AMaxWorld];
  __shared__ int s_late;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int remote = b / kA2ABlocksPerRank, part = b % kA2ABlocksPerRank;
  const unsigned epoch = a.state[1 + b] + 1u;
  auto barrier = [&](int phase) {  // every rank's block b has arrived (all blocks of a rank take part: a rank's staging is read by all)
    __threadfence_system();
    __syncthreads();
    if (tid < a.world) {
      unsigned* f = a.flags[tid] + ((size_t)phase * kA2AMaxBlocks + b) * kA2AMaxWorld + a.rank;
      __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned* w = a.flags[a.rank] + ((size_t)phase * kA2AMaxBlocks + b) * kA2AMaxWorld + tid;
      if (!wait_flag(w, epoch, a.timeout_ticks)) { atomicOr(a.state, 1u); s_late = 1; }
    }
    __syncthreads();
  };
  if (tid == 0) s_late = 0;
  barrier(0);
  if (s_late) {
    // a peer never staged its inputs (status bit 0): nothing of its memory is read; this block reports zero rows from `remote`
    // and still takes part in the closing barrier so that the epochs stay in step
    if (part == 0 && tid == 0) a.out_splits[remote] = 0;
    barrier(1);
    if (tid == 0) a.state[1 + b] = epoch;
    return;
  }
  // offsets from the peers' split vectors (_exchange_row_offsets): remote's whole vector, and what every rank sends to this one
  if (tid < a.world) {
    const char* p = reinterpret_cast<const char*>(a.splits[remote] + tid);
    s_remote[tid] = (long long)(((unsigned long long)ld_sys4(p + 4) << 32) | ld_sys4(p));
    const char* q = reinterpret_cast<const char*>(a.splits[tid] + a.rank);
    s_to_me[tid] = (long long)(((unsigned long long)ld_sys4(q + 4) << 32) | ld_sys4(q));
  }
  __syncthreads();
  long long in_off = 0, out_off = 0;
  for (int r = 0; r < a.rank; ++r) in_off += s_remote[r];
  for (int q = 0; q < remote; ++q) out_off += s_to_me[q];
  long long n = s_remote[a.rank];
  if (n < 0) n = 0;
  if (in_off < 0 || in_off + n > a.max_in_rows) {  // a split vector whose prefix leaves the peer's staged rows: never read past them
    if (tid == 0) atomicOr(a.state, 4u);
    in_off = std::min<long long>(std::max<long long>(in_off, 0), a.max_in_rows);
    n = std::max<long long>(0, std::min<long long>(n, a.max_in_rows - in_off));
  }
  if (out_off + n > a.max_out_rows) {  // never write past the output: report it instead
    if (tid == 0) atomicOr(a.state, 2u);
    n = std::max<long long>(0, a.max_out_rows - out_off);
  }
  if (part == 0 && tid == 0) a.out_splits[remote] = n;
  // rows: 16-byte units (row_bytes % 16 == 0, 16-byte aligned buffers)
  {
    const char* src = a.data[remote] + in_off * a.row_bytes;
    char* dst = a.out_data + out_off * a.row_bytes;
    const long long units = n * a.row_bytes / 16;
    const long long u0 = units * part / kA2ABlocksPerRank, u1 = units * (part + 1) / kA2ABlocksPerRank;
    for (long long i = u0 + tid; i < u1; i += 256) *reinterpret_cast<u32x4*>(dst + i * 16) = ld_sys16(src + i * 16);
  }
  // scale rows: K / 32 bytes each -- 16-, 4- or 1-byte units, whatever the row size allows
  {
    const char* src = a.scales[remote] + in_off * a.scale_row_bytes;
    char* dst = a.out_scales + out_off * a.scale_row_bytes;
    const long long bytes = n * a.scale_row_bytes;
    const int unit = (a.scale_row_bytes % 16 == 0) ? 16 : (a.scale_row_bytes % 4 == 0) ? 4 : 1;
    const long long units = bytes / unit;
    const long long u0 = units * part / kA2ABlocksPerRank, u1 = units * (part + 1) / kA2ABlocksPerRank;
    if (unit == 16) {
      for (long long i = u0 + tid; i < u1; i += 256) *reinterpret_cast<u32x4*>(dst + i * 16) = ld_sys16(src + i * 16);
    } else if (unit == 4) {
      for (long long i = u0 + tid; i < u1; i += 256) *reinterpret_cast<uint32_t*>(dst + i * 4) = ld_sys4(src + i * 4);
    } else {
      for (long long i = u0 + tid; i < u1; i += 256) dst[i] = (char)ld_sys1(src + i);
    }
  }
  barrier(1);
  if (tid == 0) a.state[1 + b] = epoch;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int64_t ao_moe_a2a_flag_bytes(void) { return (int64_t)2 * kA2AMaxBlocks * kA2AMaxWorld * sizeof(unsigned); }
extern "C" int64_t ao_moe_a2a_state_bytes(void) { return (int64_t)(1 + kA2AMaxBlocks) * sizeof(unsigned); }

extern "C" int ao_moe_a2a_v(void* const* peer_data_host, void* const* peer_scales_host, void* const* peer_splits_host, void* const* peer_flags_host,
                            void* out_data, void* out_scales, int64_t* out_splits, void* local_state, int64_t row_bytes, int64_t scale_row_bytes,
                            int64_t max_in_rows, int64_t max_out_rows, int rank, int world, void* stream) {
  AO_REQUIRE_PTR(peer_data_host);
  AO_REQUIRE_PTR(peer_scales_host);
  AO_REQUIRE_PTR(peer_splits_host);
  AO_REQUIRE_PTR(peer_flags_host);
  AO_REQUIRE_PTR(out_data);
  AO_REQUIRE_PTR(out_scales);
  AO_REQUIRE_PTR(out_splits);
  AO_REQUIRE_PTR(local_state);
  AO_REQUIRE(world >= 1 && world <= kA2AMaxWorld && rank >= 0 && rank < world, "ao_moe_a2a_v: bad rank %d / world %d (at most %d ranks)", rank, world,
             kA2AMaxWorld);
  AO_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "ao_moe_a2a_v: rows must be a multiple of 16 bytes, got %lld", (long long)row_bytes);
  AO_REQUIRE(scale_row_bytes > 0 && max_out_rows >= 0 && max_in_rows >= 0, "ao_moe_a2a_v: bad scale row size %lld / staged rows %lld / output rows %lld",
             (long long)scale_row_bytes, (long long)max_in_rows, (long long)max_out_rows);
  AO_REQUIRE((uintptr_t)out_data % 16 == 0 && (uintptr_t)out_scales % 16 == 0, "ao_moe_a2a_v: 16-byte aligned outputs expected");
  A2AArgs a{};
  for (int r = 0; r < world; ++r) {
    AO_REQUIRE(peer_data_host[r] != nullptr && peer_scales_host[r] != nullptr && peer_splits_host[r] != nullptr && peer_flags_host[r] != nullptr,
               "ao_moe_a2a_v: rank %d's buffers are not mapped", r);
    AO_REQUIRE((uintptr_t)peer_data_host[r] % 16 == 0 && (uintptr_t)peer_scales_host[r] % 16 == 0, "ao_moe_a2a_v: 16-byte aligned staging expected");
    a.data[r] = static_cast<const char*>(peer_data_host[r]);
    a.scales[r] = static_cast<const char*>(peer_scales_host[r]);
    a.splits[r] = static_cast<const long long*>(peer_splits_host[r]);
    a.flags[r] = static_cast<unsigned*>(peer_flags_host[r]);
  }
  a.out_data = static_cast<char*>(out_data); a.out_scales = static_cast<char*>(out_scales); a.out_splits = reinterpret_cast<long long*>(out_splits);
  a.state = static_cast<unsigned*>(local_state);
  a.row_bytes = row_bytes; a.scale_row_bytes = scale_row_bytes; a.max_out_rows = max_out_rows; a.max_in_rows = max_in_rows;
  a.timeout_ticks = collective_timeout_ticks(); a.rank = rank; a.world = world;
  ao::launch(moe_a2a_v_kernel, dim3((unsigned)(world * kA2ABlocksPerRank)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  AO_LAUNCH_CHECK("moe_a2a_v_kernel launch");
  return AO_OK;
}
