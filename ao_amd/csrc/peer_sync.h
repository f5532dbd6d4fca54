// Cross-GPU signalling shared by the one-shot all-reduce (allreduce_kernels.hip) and the on-device all-to-all-v (a2a_kernels.hip).
//
// Memory types (round 4).  The buffers a peer touches are allocated by ao_peer_alloc (runtime.hip) with
// hipExtMallocWithFlags, not by the caching allocator:
//   * flag blocks: hipDeviceMallocUncached -- every access goes to memory, on the owning GPU too.  A rank spins on a flag in its OWN
//     HBM that a REMOTE peer writes over xGMI; in ordinary (coarse-grained) device memory the line may sit in the local L2, and a
//     system-scope acquire does not invalidate local read-write lines, so the spin could read a stale value for ever.  (Two
//     processes on one GPU share that L2: the round-3 tests could not see this.)
//   * staging (data a peer reads): hipDeviceMallocFinegrained -- coherent at system scope without whole-cache write-backs; the
//     writer still releases it (system-scope fence) before raising its flag, the readers use sc0 sc1 loads.
// Waits are bounded by TIME (s_memrealtime, the 100 MHz constant clock), not by poll counts: ao_collective_set_timeout_ms, default
// 5 s.  A wait that expires returns false; the kernels then POISON their output (NaN / INT_MIN) and set a status bit -- a late rank
// never turns into a silently wrong sum.
#pragma once
#include "common.h"

namespace ao {

unsigned long long collective_timeout_ticks();  // runtime.hip: host side, 100 MHz ticks

__device__ __forceinline__ u32x4 ld_sys16(const char* p) {  // system-scope (sc0 sc1) loads: never served from a stale cache
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t ld_sys4(const char* p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t ld_sys1(const char* p) {
  uint32_t v;
  asm volatile("global_load_ubyte %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// spin until *w == epoch; false when `ticks` of the 100 MHz clock have passed without it
__device__ __forceinline__ bool wait_flag(const unsigned* w, unsigned epoch, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
    __builtin_amdgcn_s_sleep(8);
    if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) return false;
  }
  return true;
}

}  // namespace ao
