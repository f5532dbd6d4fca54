/*
 * C restatement of the reference CPU dequant path for the int4 tinygemm linear.
 * TEST INFRASTRUCTURE ONLY (checker in tests/, bench.py's cpu_baseline leg).
 * "port" of: groupwise_affine_dequantize_tensor + bf16 F.linear
 *   torchao/quantization/utils.py:365-455, quant_primitives.py:999-1007
 * reading the tile-packed weight (layout: oracle/int4_ref.py:_tile_coords).
 * Validated against oracle/int4_ref.py in tests/test_oracle_c.py.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp)
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bf16_to_f32(uint16_t b) { return bits_f32((uint32_t)b << 16); }
/* fp32 -> bf16 round-to-nearest-even (finite inputs) */
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u = f32_bits(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

int ao_ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* nibble (n, k) of the tile-packed weight */
static inline uint32_t packed_nibble(const uint32_t* qdata, int64_t K, int64_t n, int64_t k) {
  const int64_t kblocks = K >> 7;
  const int64_t blk = (n >> 4) * kblocks + (k >> 7);
  const int kin = (int)(k & 127);
  const int tile = kin >> 4, off = kin & 15;
  const int t = (int)(n & 15) + 16 * (off >> 2);
  const int j = tile >> 1;
  const int v = (tile & 1) * 4 + (off & 3);
  static const int slot_of_v[8] = {0, 4, 1, 5, 2, 6, 3, 7};
  const uint32_t w = qdata[(blk * 64 + t) * 4 + j];
  return (w >> (4 * slot_of_v[v])) & 0xFu;
}

/* w_dq[n][k] = bf16(bf16((q-8)*s) + z), written as bf16 bits */
void ao_ref_int4_dequantize(const int32_t* qdata, const uint16_t* sz, uint16_t* w, int64_t N, int64_t K, int G) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t k = 0; k < K; ++k) {
      const uint16_t* p = sz + ((k / G) * N + n) * 2;
      const float s = bf16_to_f32(p[0]), z = bf16_to_f32(p[1]);
      const float q = (float)((int)packed_nibble((const uint32_t*)qdata, K, n, k) - 8);
      w[n * K + k] = f32_to_bf16(bf16_round(q * s) + z);
    }
  }
}

/* y[M][N] = x[M][K] @ dequant(qdata)^T, fp32 accumulate, bf16 out: the
 * reference's "dequantize the whole weight, then bf16 matmul" CPU path. */
void ao_ref_int4_linear(const uint16_t* x, const int32_t* qdata, const uint16_t* sz, uint16_t* y, int64_t M,
                        int64_t N, int64_t K, int G) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    float acc[16];
    for (int64_t m0 = 0; m0 < M; m0 += 16) {
      const int mb = (int)((M - m0) < 16 ? (M - m0) : 16);
      for (int i = 0; i < mb; ++i) acc[i] = 0.f;
      for (int64_t k = 0; k < K; ++k) {
        const uint16_t* p = sz + ((k / G) * N + n) * 2;
        const float s = bf16_to_f32(p[0]), z = bf16_to_f32(p[1]);
        const float q = (float)((int)packed_nibble((const uint32_t*)qdata, K, n, k) - 8);
        const float wv = bf16_round(bf16_round(q * s) + z);
        for (int i = 0; i < mb; ++i) acc[i] += bf16_to_f32(x[(m0 + i) * K + k]) * wv;
      }
      for (int i = 0; i < mb; ++i) y[(m0 + i) * N + n] = f32_to_bf16(acc[i]);
    }
  }
}
