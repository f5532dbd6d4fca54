// LDS-DMA and agent-scope access helpers shared by the int4 kernels (gfx950 only).
//
// global_load_lds_*: each lane names a global address, the wave writes LDS[M0 + lane * size] -- no VGPR
// destination, so an inline-asm issue is register-safe (an inline-asm load INTO registers is not: the
// register allocator may move the destination before the data lands).  Retired by hand-counted
// s_waitcnt vmcnt(N); VMEM operations return in order, so "at most N outstanding" means everything
// older than the N youngest has landed.
#pragma once
#include "common.h"

namespace ao {

__device__ __forceinline__ uint32_t lds_offset(const void* p) {
  return (uint32_t)reinterpret_cast<uintptr_t>(p);  // flat address of LDS = aperture base (high dword) + offset
}
// each lane: 16 B from gsrc -> LDS[lds_dst + lane * 16]; lds_dst wave-uniform
__device__ __forceinline__ void dma_b128_nt(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// same, default cache policy (data other workgroups re-read from L2)
__device__ __forceinline__ void dma_b128(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// each lane: 4 B from gsrc -> LDS[lds_dst + lane * 4]
__device__ __forceinline__ void dma_b32(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// 16 B per lane, agent scope (sc1): reads past this XCD's L2.  One request per granule, so value
// and tag are read together (two 4-byte DMAs could pair a stale value with a fresh tag).
__device__ __forceinline__ void dma_b128_sc1(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// agent-scope write-through 16-byte store (no register destination: register-safe, fire and forget;
// the trailing s_nop keeps the next instruction off the data registers until the store has read them)
__device__ __forceinline__ void store_b128_sc1(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// scalar-base forms: address = sbase (wave-uniform, SGPR pair) + voff (32-bit unsigned byte offset per lane)
__device__ __forceinline__ void dma_b128_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_b128_nt_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_b32_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace ao
