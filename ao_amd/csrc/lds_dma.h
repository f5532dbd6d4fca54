// LDS-DMA helpers of the int4 kernels (gfx950 only).
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
// address = sbase (wave-uniform, SGPR pair) + voff (32-bit unsigned byte offset per lane); each lane's 16 (4) bytes land
// at LDS[lds_dst + lane * 16 (4)], lds_dst wave-uniform.  M0 is saved and restored around the instruction.
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
// N consecutive 1 KiB pieces with ONE M0 write: piece i lands at LDS[lds_dst + i * 1024 + lane * 16] and reads sbase + voff[i] (the
// instruction's offset field moves BOTH addresses, so piece i's scalar base is lowered by the same i * 1024).  Back-to-back
// global_load_lds behind separate M0 writes cost a wave ~170 cycles each in the batched kernels (round-3 trace); the M0 write has
// to wait for the previous DMA's address phase.
__device__ __forceinline__ void dma_b128_x4(const char* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %5\n\t"
               "global_load_lds_dwordx4 %2, %6 offset:1024\n\t"
               "global_load_lds_dwordx4 %3, %7 offset:2048\n\t"
               "global_load_lds_dwordx4 %4, %8 offset:3072\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(sbase - 1024), "s"(sbase - 2048), "s"(sbase - 3072), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void dma_b128_x2(const char* sbase, uint32_t v0, uint32_t v1, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %3\n\t"
               "global_load_lds_dwordx4 %2, %4 offset:1024\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sbase), "s"(sbase - 1024), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace ao
