// Shared host/device helpers for the gfx950 low-bit kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/ao_mi355.h"

namespace ao {

// ---- host side: error plumbing (thread-local message, never exit()) --------
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_failed(hipError_t e, const char* what);  // sets message, returns AO_ERR_HIP

// opt a kernel into > 48 KiB of dynamic LDS on the current device (once per kernel and device; runtime.hip)
int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what);

#define AO_REQUIRE(cond, ...)                \
  do {                                       \
    if (!(cond)) {                           \
      ::ao::set_error(__VA_ARGS__);          \
      return AO_ERR_INVALID_ARGUMENT;        \
    }                                        \
  } while (0)

#define AO_REQUIRE_PTR(p)                                        \
  do {                                                           \
    if ((p) == nullptr) {                                        \
      ::ao::set_error("%s: null pointer argument '%s'", __func__, #p); \
      return AO_ERR_NULL_POINTER;                                \
    }                                                            \
  } while (0)

#define AO_LAUNCH_CHECK(what)                          \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return ::ao::hip_failed(e__, what); \
  } while (0)

// ---- launch helper ---------------------------------------------------------
// While profiling is enabled (ao_prof_enable) every launch is bracketed by HIP
// extension events that timestamp the dispatch itself (kernel begin/end, no
// launch gaps) -- what bench.py uses for the live roofline numbers.
bool prof_next_events(hipEvent_t* start, hipEvent_t* stop);

template <typename K, typename... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t stream, Args... args) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof_next_events(&e0, &e1)) {
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)smem, stream, e0, e1, 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)smem, stream, args...);
  }
}

// ---- device side -----------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;  // gfx950 wavefront

__device__ __forceinline__ float bits_to_f32(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f32_to_bits(float f) { return __float_as_uint(f); }

// bf16 bit pattern (in the low 16 bits) -> fp32
__device__ __forceinline__ float bf16_lo_to_f32(uint32_t packed) { return bits_to_f32(packed << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(uint32_t packed) { return bits_to_f32(packed & 0xffff0000u); }

// two fp32 -> packed bf16 pair, round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}

// fp32 -> fp32 holding the nearest bf16 value (RNE), i.e. torch's ".to(bf16)"
__device__ __forceinline__ float round_bf16(float f) {
  __bf16 b = (__bf16)f;
  return (float)b;
}

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

// max over the 64 lanes without LDS traffic: DPP within rows of 16, then the four row results through SGPRs
__device__ __forceinline__ float wave_max(float m) {
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  m = fmaxf(m, dpp(m, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
  m = fmaxf(m, dpp(m, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
  m = fmaxf(m, dpp(m, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  m = fmaxf(m, dpp(m, std::integral_constant<int, 0x140>{}));  // row_mirror
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

}  // namespace ao
