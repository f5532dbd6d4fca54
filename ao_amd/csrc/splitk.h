// Split-K meeting shared by the batched kernels (int4 register-B, fp8 rowwise register-B).
#pragma once
#include "common.h"

namespace ao {

// Workspace: one per (device, stream); at most kSplitSlots streams per device hold one at a time (a further stream evicts the least
// recently used slot after a device synchronise); sized on demand (8 MiB steps of two) up to kSplitSlotFloats fp32 of parked partial tiles
// (kSplitMaxTiles tiles of 128 x 128; smaller tiles pack more of them) + kSplitMaxTickets tickets, one per output tile -- the
// smallest tile a kernel parks is 64 x 16, two parts at least, so 4096 tickets cover every launch the float budget admits only
// if the budget stays <= 4096 * 2 * 1024 floats; launchers check both.  Allocated on the first split-K launch on a stream
// (outside stream capture); runtime.hip.
constexpr int kSplitSlots = 8;
constexpr int kSplitMaxTiles = 2048;
constexpr int kSplitMaxTickets = 16384;
constexpr size_t kSplitSlotFloats = (size_t)kSplitMaxTiles * 128 * 128;  // 128 MiB
// need_floats: what this launch parks (tiles x parts x tile floats); the slot grows to the largest request seen on its stream
// The two-level meeting gathers <= 4 groups of <= 4 parts.  Every launcher passes its part count here and is refused beyond
// kSplitMaxParts -- a ticket that can never read "last" would skip the epilogue silently.
constexpr int kSplitMaxParts = 16;
int splitk_workspace(hipStream_t stream, float** part, unsigned** tickets, size_t need_floats, int parts = 1);
// ---------------------------------------------------------------------------
// Split-K meeting: every part parks its fp32 tile in the workspace
// ([tile][part][reg][thread], 16 B per thread and register: coalesced), takes a ticket, and the last one
// to arrive adds the parts in part order (so the sum does not depend on arrival order) and returns true:
// it stores the tile.  The parts sit on different XCDs (non-coherent L2s): the tiles are written through
// and read with agent-scope (sc1) accesses instead of device fences -- a fence writes back / invalidates
// the whole L2 and cost ~60 us per launch.  `flag` is any LDS word no wave is still using.
// ---------------------------------------------------------------------------
// INT: the registers hold int32 partial sums (exact integer adds) instead of fp32.
// (Round 5 built a same-XCD form -- all parts of a tile on one XCD, parked with plain stores in its L2, the placement checked per tile by
// the ticket -- and measured it at +- 2 % of this one on 40 cells, profiles/midm_sweep_r05.jsonl: the meeting costs its three dependent
// round trips whether they end in the XCD's L2 or at the fabric.  Removed in round 6; git history has it.)
template <int NREG, int NTHR, bool INT = false>
__device__ __forceinline__ bool split_k_meet(f32x4 (&acc)[NREG], float* ws, unsigned* tickets, int tile, int S, int ks, int tid, int* flag,
                                             bool active = true) {
  constexpr int kSc1 = 16;  // cache-policy bit 4 = sc1 on gfx950
  constexpr int kPark = kSc1;
  constexpr int kRegBytes = NTHR * 16;
  constexpr int kPartBytes = NREG * kRegBytes;
  const __amdgpu_buffer_rsrc_t rws =
      __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)tile * S * (kPartBytes / 4), 0, S * kPartBytes, 0x00020000);
  // `active` = false: a wave that holds no accumulators (a DMA-producer wave) only takes part in the workgroup barriers
  // (round 5) the stores read their data from VGPR copies that are pinned (asm operands) until the stores are out: with the accumulators in
  // AGPRs -- the 256-row slabs -- the compiler stages each store's data through scratch VGPRs and re-used one for the NEXT store's address
  // one instruction later (tests/test_isa_structure.py caught it): the same hazard as below, on registers the old pin did not cover
  f32x4 parked[NREG];
#pragma unroll
  for (int r = 0; r < NREG; ++r) {
    parked[r] = acc[r];
    asm volatile("" : "+v"(parked[r]));
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < NREG; ++r)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, parked[r]), rws, tid * 16 + r * kRegBytes, ks * kPartBytes, kPark);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // written through before the ticket is taken
#pragma unroll
  for (int r = 0; r < NREG; ++r) asm volatile("" ::"v"(parked[r]));
  // The accumulators stay LIVE until the stores have completed.  Round 3: hipcc re-used a data register of a just-issued
  // `buffer_store_dwordx4 v[6:9], v32, s[8:11], s1 offen sc1` for the next store's address in the very next instruction
  // (`v_or_b32 v8, 0x2000, v32`); LLVM knows that hazard only for stores WITHOUT an SGPR soffset, gfx950 showed it with one: now and
  // then the parked tile carried the address bits instead of component z of its first register (rows 4 kq + 2 of m-tile 0 off by one
  // part's contribution; the fp8 kernel at (33, 4096, 4096), cold launches; tools/stress_fp8_splitk.py).
#pragma unroll
  for (int r = 0; r < NREG; ++r) asm volatile("" ::"v"(acc[r]));
  __syncthreads();
  if (tid == 0) {
    const unsigned inc = 1u;
    const unsigned t = __hip_atomic_fetch_add(&tickets[tile], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (t == (unsigned)S - 1);
    *flag = last;
    // everyone has arrived: leave the ticket ready for the next launch
    if (last) {
      __hip_atomic_store(&tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!*flag || !active) return false;
  // parts are read in batches (<= 32 loads in flight per thread; indices past S re-read the last part and are
  // not added), summed in part order
  constexpr int U = (NREG >= 16) ? 2 : 4;
  f32x4 sum[NREG];
#pragma unroll
  for (int r = 0; r < NREG; ++r) sum[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < S; q0 += U) {
    f32x4 v[U][NREG];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < NREG; ++r)
        v[u][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, tid * 16 + r * kRegBytes, min(q0 + u, S - 1) * kPartBytes, kSc1));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool keep = q0 + u < S;
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        if constexpr (INT) {
          const i32x4 a = __builtin_bit_cast(i32x4, sum[r]), b = __builtin_bit_cast(i32x4, v[u][r]);
          sum[r] = __builtin_bit_cast(f32x4, i32x4{a.x + (keep ? b.x : 0), a.y + (keep ? b.y : 0), a.z + (keep ? b.z : 0), a.w + (keep ? b.w : 0)});
        } else {
          sum[r].x += keep ? v[u][r].x : 0.f; sum[r].y += keep ? v[u][r].y : 0.f;
          sum[r].z += keep ? v[u][r].z : 0.f; sum[r].w += keep ? v[u][r].w : 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NREG; ++r) acc[r] = sum[r];
  return true;
}

// ---------------------------------------------------------------------------
// Two-level meeting (round 4).  With S parts the meeting above makes the last arriver read S - 1 parked tiles through ONE CU's
// memory path in S / 4 dependent round trips (fp8 qkv shard at M = 128: 15 x 64 KiB, ~8 us of an 18 us launch).  Here the parts of a
// tile form groups of four: the last arriver of a GROUP adds its (<= 4) parts in part order and parks the group sum, takes a
// second-level ticket, and the last group adds the (<= 4, S <= 16) group sums in group order: two round trips on the critical
// path, the first level spread over S / 4 workgroups.  The sum is ((p0 + p1 + p2 + p3) + (p4 + ..) + ..) whatever the arrival
// order: reproducible.  Workspace per tile: S + ceil(S / 4) parked parts; tickets per tile: 1 + ceil(S / 4).
// ---------------------------------------------------------------------------
// UMAX: parts gathered per round trip (4 x NREG x 4 VGPRs: callers that run four waves per SIMD pass 2 for 8-register tiles)
// RC (round 5): registers per batch.  A 256 x 256 tile on 512 threads is 32 registers of 4 floats per thread (128 VGPRs): parked and gathered
// in batches of RC registers so that neither the pinned store copies nor the gather's landing registers double the tile.
template <int NREG, int NTHR, bool INT = false, int UMAX = 4, int RC = NREG>
__device__ __forceinline__ bool split_k_meet2(f32x4 (&acc)[NREG], float* ws, unsigned* tickets, int tile, int S, int ks, int tid, int* flag) {
  constexpr int kSc1 = 16;
  constexpr int kPark = kSc1;
  constexpr int kRegBytes = NTHR * 16;
  constexpr int kPartBytes = NREG * kRegBytes;
  const int NG = (S + 3) >> 2;          // groups
  const int slots = S + NG;             // parked tiles per output tile: parts, then group sums
  const __amdgpu_buffer_rsrc_t rws =
      __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)tile * slots * (kPartBytes / 4), 0, slots * kPartBytes, 0x00020000);
  unsigned* tk = tickets + (size_t)tile * (1 + NG);  // [0] second level, [1 + g] group g
  static_assert(NREG % RC == 0, "batches of RC registers");
  auto park = [&](int slot) {
    if constexpr (RC == NREG) {
      f32x4 parked[NREG];  // VGPR copies, pinned until the stores are out (see split_k_meet)
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        parked[r] = acc[r];
        asm volatile("" : "+v"(parked[r]));
      }
#pragma unroll
      for (int r = 0; r < NREG; ++r)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, parked[r]), rws, tid * 16 + r * kRegBytes, slot * kPartBytes, kPark);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // written through before the ticket is taken
#pragma unroll
      for (int r = 0; r < NREG; ++r) asm volatile("" ::"v"(parked[r]));  // the data registers stay untouched until the stores are out
    } else {
      // in batches: the copies of a batch stay pinned across 16 idle cycles behind its last store (what the stream-K kernel's in-loop
      // park does, DESIGN 4.10: the store has read its data registers by then), then the next batch may take the registers over
#pragma unroll
      for (int r0 = 0; r0 < NREG; r0 += RC) {
        f32x4 parked[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
          parked[r] = acc[r0 + r];
          asm volatile("" : "+v"(parked[r]));
        }
#pragma unroll
        for (int r = 0; r < RC; ++r)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, parked[r]), rws, tid * 16 + (r0 + r) * kRegBytes, slot * kPartBytes, kPark);
        asm volatile("s_nop 15" ::: "memory");
#pragma unroll
        for (int r = 0; r < RC; ++r) asm volatile("" ::"v"(parked[r]));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  };
  auto last_of = [&](unsigned* t, int n) {
    if (tid == 0) {
      const unsigned inc = 1u;
      const unsigned v = __hip_atomic_fetch_add(t, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = (v == (unsigned)n - 1);
      *flag = last;
      if (last) {
        __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      }
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();  // (flag is re-used by the second level)
    return last;
  };
  // acc = slot[first] + .. + slot[first + n - 1], n <= 4, in slot order, UMAX parts per round trip.  acc is dead here: the own part
  // is re-read from its slot like the others.
  auto gather = [&](int first, int n) {
    constexpr int U = UMAX;
#pragma unroll
    for (int r0 = 0; r0 < NREG; r0 += RC) {
#pragma unroll
      for (int u0 = 0; u0 < 4; u0 += U) {
        if (u0 > 0 && u0 >= n) break;  // uniform
        f32x4 v[U][RC];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < RC; ++r)
            v[u][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, tid * 16 + (r0 + r) * kRegBytes, (first + min(u0 + u, n - 1)) * kPartBytes, kSc1));
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool keep = u0 + u < n;
#pragma unroll
          for (int r = 0; r < RC; ++r) {
            f32x4& d = acc[r0 + r];
            if (u0 + u == 0) {
              d = v[0][r];
            } else if constexpr (INT) {
              const i32x4 a = __builtin_bit_cast(i32x4, d), b = __builtin_bit_cast(i32x4, v[u][r]);
              d = __builtin_bit_cast(f32x4, i32x4{a.x + (keep ? b.x : 0), a.y + (keep ? b.y : 0), a.z + (keep ? b.z : 0), a.w + (keep ? b.w : 0)});
            } else {
              d.x += keep ? v[u][r].x : 0.f; d.y += keep ? v[u][r].y : 0.f;
              d.z += keep ? v[u][r].z : 0.f; d.w += keep ? v[u][r].w : 0.f;
            }
          }
        }
      }
    }
  };
  const int g = ks >> 2, gn = min(4, S - 4 * g);
  park(ks);
  if (!last_of(tk + 1 + g, gn)) return false;
  gather(4 * g, gn);
  if (NG == 1) return true;
  park(S + g);
  if (!last_of(tk, NG)) return false;
  gather(S, NG);  // NG <= 4 (S <= 16)
  return true;
}

}  // namespace ao
