// Register-only throughput lab for the exact int4 -> bf16 dequant (+ product) sequence of the decode kernel.
// No memory traffic: every wave dequantises the same four packed words over and over (made opaque to the compiler each
// iteration) and feeds the MFMA, so what is measured is pure issue cost per 1 KiB packed block (4 words per lane) per SIMD,
// for several formulations / instruction orders, at 1..8 waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-vectorize -o dequant_lab tools/dequant_lab.hip && ./dequant_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ s16x4 identity_fragment(int lane) {
  const int hot = lane & 3;
  s16x4 f;
  f.x = hot == 0 ? (short)0x3F80 : (short)0; f.y = hot == 1 ? (short)0x3F80 : (short)0;
  f.z = hot == 2 ? (short)0x3F80 : (short)0; f.w = hot == 3 ? (short)0x3F80 : (short)0;
  return f;
}
__device__ __forceinline__ f32x4 widen_add(s16x4 ident, uint32_t lo_pair, uint32_t hi_pair, f32x4 c) {
  const u32x2 bb = {lo_pair, hi_pair};
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ident, __builtin_bit_cast(s16x4, bb), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(u32x4 a, const uint32_t (&b)[4], f32x4 acc) {
  const u32x4 bv = {b[0], b[1], b[2], b[3]};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
}

struct Ctx {
  float s, n8s, z;
  s16x4 ident;
};

// per-word pipeline state
struct W {
  f32x2 r0, r1, r2, r3;
  uint32_t tp0, tp1, tp2, tp3;
  f32x4 w0, w1;
  uint32_t out[4];
};

__device__ __forceinline__ void st0(W& d, uint32_t p) {  // masks + fp8-trick conversion: 8 nibbles -> 8 fp32
  const uint32_t lo = p & 0x0F0F0F0Fu, hi = (p >> 4) & 0x0F0F0F0Fu;
  d.r0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, false);
  d.r1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, true);
  d.r2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, false);
  d.r3 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, true);
}
template <bool PK>
__device__ __forceinline__ void st1(W& d, const Ctx& c) {  // (q-8)*s exact, rounding #1
  f32x2 t0, t1, t2, t3;
  if constexpr (PK) {
    t0 = d.r0 * c.s + c.n8s; t1 = d.r1 * c.s + c.n8s; t2 = d.r2 * c.s + c.n8s; t3 = d.r3 * c.s + c.n8s;
  } else {
    auto f = [&](float q) { float t; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(q), "v"(c.s), "v"(c.n8s)); return t; };
    t0 = {f(d.r0.x), f(d.r0.y)}; t1 = {f(d.r1.x), f(d.r1.y)}; t2 = {f(d.r2.x), f(d.r2.y)}; t3 = {f(d.r3.x), f(d.r3.y)};
  }
  d.tp0 = pack_bf16x2(t0.x, t1.x); d.tp1 = pack_bf16x2(t2.x, t3.x);
  d.tp2 = pack_bf16x2(t0.y, t1.y); d.tp3 = pack_bf16x2(t2.y, t3.y);
}
__device__ __forceinline__ void st2a(W& d, const Ctx& c) { const f32x4 zz = {c.z, c.z, c.z, c.z}; d.w0 = widen_add(c.ident, d.tp0, d.tp1, zz); }
__device__ __forceinline__ void st2b(W& d, const Ctx& c) { const f32x4 zz = {c.z, c.z, c.z, c.z}; d.w1 = widen_add(c.ident, d.tp2, d.tp3, zz); }
__device__ __forceinline__ void st2_valu(W& d, const Ctx& c) {  // rounding-#2 add on the VALU (no 4x4x4 MFMA)
  auto lo = [](uint32_t p) { return __uint_as_float(p << 16); };
  auto hi = [](uint32_t p) { return __uint_as_float(p & 0xffff0000u); };
  d.w0 = f32x4{lo(d.tp0) + c.z, hi(d.tp0) + c.z, lo(d.tp1) + c.z, hi(d.tp1) + c.z};
  d.w1 = f32x4{lo(d.tp2) + c.z, hi(d.tp2) + c.z, lo(d.tp3) + c.z, hi(d.tp3) + c.z};
}
__device__ __forceinline__ void st3a(W& d) { d.out[0] = pack_bf16x2(d.w0.x, d.w0.y); d.out[1] = pack_bf16x2(d.w0.z, d.w0.w); }
__device__ __forceinline__ void st3b(W& d) { d.out[2] = pack_bf16x2(d.w1.x, d.w1.y); d.out[3] = pack_bf16x2(d.w1.z, d.w1.w); }

__device__ __forceinline__ void opaque(u32x4& w) { asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w)); }

// VARIANT:
//  0  product order: word by word (dequant_word_mfma + MFMA), what the compiler gets today
//  1  same with v_fma_f32 instead of v_pk_fma_f32
//  2  word by word, rounding-#2 add on the VALU (no 4x4x4 MFMA)
//  3  stage by stage over the four words (sched_group_barrier): 60 VALU, 4x4x4 MFMAs with the cvt_pk of earlier results in
//     their shadow, products last, two accumulators
//  4  software-pipelined across blocks, products of block b-1 and 4x4x4 of block b dealt evenly: (1 MFMA, 6 VALU) x 12
//  5  software-pipelined across blocks: products of block b-1 between st0/st1 of block b, 4x4x4 between the st3 cvt_pk
//  6  as 5 with v_fma_f32
//  7  VALU part only (st0, st1, st3 on stale data; no MFMA at all)        -- component
//  8  MFMA part only (8 x 4x4x4 + 4 x 16x16x32 per block)                  -- component
//  9  as 5, product by v_dot2_f32_bf16 instead of the 16x16x32 MFMA (M = 1 only)
template <int V>
__global__ __launch_bounds__(1024) void lab_kernel(const u32x4* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 w = in[threadIdx.x & 255];
  const u32x4 a0 = in[256 + lane], a1 = in[320 + lane], a2 = in[384 + lane], a3 = in[448 + lane];
  Ctx c;
  c.s = __uint_as_float((w.x & 0x7fu) << 16 | 0x3b000000u);
  c.n8s = -8.0f * c.s;
  c.z = __uint_as_float((w.y & 0x7fu) << 16 | 0x3a800000u);
  c.ident = identity_fragment(lane);
  f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  W d[4];
  const unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (V == 0 || V == 1 || V == 2) {
    for (int it = 0; it < iters; ++it) {
      opaque(w);
      const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
      const u32x4 av[4] = {a0, a1, a2, a3};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st0(d[j], wd[j]);
        st1<V != 1>(d[j], c);
        if constexpr (V == 2) st2_valu(d[j], c); else { st2a(d[j], c); st2b(d[j], c); }
        st3a(d[j]); st3b(d[j]);
        acc = mma(av[j], d[j].out, acc);
      }
    }
  } else if constexpr (V == 3) {
    // stage by stage over the four words; order left to the compiler inside the block
    for (int it = 0; it < iters; ++it) {
      opaque(w);
      const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
      const u32x4 av[4] = {a0, a1, a2, a3};
#pragma unroll
      for (int j = 0; j < 4; ++j) st0(d[j], wd[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) st1<true>(d[j], c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { st2a(d[j], c); st2b(d[j], c); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { st3a(d[j]); st3b(d[j]); }
      acc = mma(av[0], d[0].out, acc); acc2 = mma(av[1], d[1].out, acc2);
      acc = mma(av[2], d[2].out, acc); acc2 = mma(av[3], d[3].out, acc2);
      // 60 VALU (st0, st1), then 4x4x4 with the cvt_pk of earlier results in their shadow, then the products
      __builtin_amdgcn_sched_group_barrier(0x2, 60, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x2, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
    }
  } else if constexpr (V == 4 || V == 5 || V == 6 || V == 9) {
    // software-pipelined across blocks: the products of block b-1 and the 4x4x4 MFMAs of block b are dealt between the VALU
    // work of block b (V4: 1 MFMA : 5 VALU everywhere; V5/6/9: products between st0/st1, 4x4x4 between the st3 cvt_pk)
    constexpr bool PK = (V != 6);
    W e[4];  // previous block, ready for the product
#pragma unroll
    for (int j = 0; j < 4; ++j) { e[j].out[0] = w.x; e[j].out[1] = w.y; e[j].out[2] = w.z; e[j].out[3] = w.w; }
    auto prod = [&](const u32x4& a, const W& ww, f32x4& ac) {
      if constexpr (V == 9) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        ac.x = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a.x), __builtin_bit_cast(b2, ww.out[0]), ac.x, false);
        ac.y = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a.y), __builtin_bit_cast(b2, ww.out[1]), ac.y, false);
        ac.z = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a.z), __builtin_bit_cast(b2, ww.out[2]), ac.z, false);
        ac.w = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a.w), __builtin_bit_cast(b2, ww.out[3]), ac.w, false);
      } else {
        ac = mma(a, ww.out, ac);
      }
    };
    for (int it = 0; it < iters; ++it) {
      opaque(w);
      const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
      prod(a0, e[0], acc); prod(a1, e[1], acc2); prod(a2, e[2], acc); prod(a3, e[3], acc2);
#pragma unroll
      for (int j = 0; j < 4; ++j) st0(d[j], wd[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) st1<PK>(d[j], c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { st2a(d[j], c); st2b(d[j], c); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { st3a(d[j]); st3b(d[j]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { e[j].out[0] = d[j].out[0]; e[j].out[1] = d[j].out[1]; e[j].out[2] = d[j].out[2]; e[j].out[3] = d[j].out[3]; }
      if constexpr (V == 4) {
#pragma unroll
        for (int i = 0; i < 12; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 6, 0); }
        __builtin_amdgcn_sched_group_barrier(0x2, 16, 0);
      } else if constexpr (V == 9) {
        __builtin_amdgcn_sched_group_barrier(0x2, 76 + (PK ? 0 : 16), 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x2, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
      } else {
        constexpr int PER = PK ? 15 : 19;
#pragma unroll
        for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, PER, 0); }
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x2, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
      }
    }
  } else if constexpr (V == 7) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j].w0 = f32x4{c.s, c.z, c.s, c.z}; d[j].w1 = d[j].w0; }
    for (int it = 0; it < iters; ++it) {
      opaque(w);
      const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st0(d[j], wd[j]);
        st1<true>(d[j], c);
        asm volatile("" : "+v"(d[j].w0.x), "+v"(d[j].w0.y), "+v"(d[j].w0.z), "+v"(d[j].w0.w), "+v"(d[j].w1.x), "+v"(d[j].w1.y),
                     "+v"(d[j].w1.z), "+v"(d[j].w1.w));
        st3a(d[j]); st3b(d[j]);
        acc.x += __uint_as_float(d[j].out[0] ^ d[j].out[1] ^ d[j].out[2] ^ d[j].out[3] ^ d[j].tp0 ^ d[j].tp1 ^ d[j].tp2 ^ d[j].tp3);
      }
    }
  } else if constexpr (V == 8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j].tp0 = w.x; d[j].tp1 = w.y; d[j].tp2 = w.z; d[j].tp3 = w.w; d[j].out[0] = w.x; d[j].out[1] = w.y; d[j].out[2] = w.z; d[j].out[3] = w.w; }
    for (int it = 0; it < iters; ++it) {
      const u32x4 av[4] = {a0, a1, a2, a3};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st2a(d[j], c); st2b(d[j], c);
        acc2 += d[j].w0 + d[j].w1;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = mma(av[j], d[j].out, acc);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  acc += acc2;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (lane == 0) cyc[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
int run(const char* what, const u32x4* din, float* dout, unsigned long long* dcyc, hipStream_t s) {
  const int iters = 4000;
  printf("V%d %-72s", V, what);
  for (int wps : {1, 2, 4, 6, 8}) {
    const int grid = 256 * wps;  // 4-wave workgroups: one wave per SIMD each; `wps` workgroups per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(lab_kernel<V>, dim3(grid), dim3(256), 0, s, din, dout, dcyc, 64);  // warm
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(lab_kernel<V>, dim3(grid), dim3(256), 0, s, din, dout, dcyc, iters);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4);
    CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (auto v : h) sum += (double)v;
    const double wave_cyc_per_block = sum / h.size() / iters;           // shader clocks a wave spends per block
    const double us = ms * 1e3;
    const double ns_per_block_per_simd = us * 1e3 / ((double)iters * wps);  // wall time per block per SIMD
    printf(" | %dw: %6.1f cyc/blk/wave, %6.1f ns/blk/SIMD", wps, wave_cyc_per_block, ns_per_block_per_simd);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  printf("\n");
  return 0;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  u32x4* din; float* dout; unsigned long long* dcyc;
  std::vector<uint32_t> h(512 * 4);
  uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
  for (int i = 256 * 4; i < 512 * 4; ++i) h[i] = 0x3f803f80u;  // A fragments: bf16 1.0
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dout, 256 * 8 * 256 * 4)); CK(hipMalloc(&dcyc, 256 * 8 * 4 * 8));
  printf("dequant lab: 1 KiB packed block = 4 words/lane = 32 nibbles/lane.  HBM-rate budget at 6.3 TB/s: 1 block / SIMD / 166 ns\n");
  run<0>("word by word (product order)", din, dout, dcyc, s);
  run<1>("word by word, v_fma_f32 instead of v_pk_fma_f32", din, dout, dcyc, s);
  run<2>("word by word, rounding-2 add on the VALU (no 4x4x4)", din, dout, dcyc, s);
  run<3>("stage by stage over 4 words, 4x4x4 shadowed by cvt_pk", din, dout, dcyc, s);
  run<4>("pipelined across blocks, (1 MFMA, 6 VALU) x 12", din, dout, dcyc, s);
  run<5>("pipelined across blocks: product MFMAs between next block's st0/st1", din, dout, dcyc, s);
  run<6>("as V5 with v_fma_f32", din, dout, dcyc, s);
  run<9>("as V5 with v_dot2_f32_bf16 product (no 16x16x32)", din, dout, dcyc, s);
  run<7>("component: VALU only (st0 st1 st3)", din, dout, dcyc, s);
  run<8>("component: MFMA only (8 x 4x4x4 + 4 x 16x16x32)", din, dout, dcyc, s);
  return 0;
}
