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

/* ------------------------------------------------------------------------------------------------
 * 8-bit paths (cpu_baseline legs of bench.py's secondary configs; validated against oracle/int8_ref.py,
 * fp8_ref.py, mx_ref.py in tests/test_oracle_c.py).
 * ---------------------------------------------------------------------------------------------- */
#include <math.h>

/* fp32 (|f| <= 448 or NaN) -> OCP e4m3fn code, round-to-nearest-even: what .to(torch.float8_e4m3fn) does in range
 * (oracle/fp8_ref.py:f32_to_e4m3) */
static inline uint8_t f32_to_e4m3(float f) {
  const uint32_t u = f32_bits(f);
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return 0x7f;
  const float af = bits_f32(a);
  if (af >= 0.015625f) { /* >= 2^-6: normal in e4m3 */
    uint32_t r = a + 0x7FFFFu + ((a >> 20) & 1u); /* RNE to 3 mantissa bits */
    r >>= 20;
    const int e = (int)(r >> 3) - 127 + 7;
    return (uint8_t)(sign | (uint8_t)((e << 3) | (int)(r & 7u)));
  }
  return (uint8_t)(sign | (uint8_t)(int)rintf(af * 512.0f)); /* subnormal: m * 2^-9, m = 8 encodes 2^-6 */
}
static inline float e4m3_to_f32(uint8_t c) {
  const int e = (c >> 3) & 15, m = c & 7;
  float v;
  if ((c & 0x7f) == 0x7f) return NAN;
  if (e == 0) v = (float)m * 0.001953125f; else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return (c & 0x80) ? -v : v;
}

/* Int8Tensor dynamic linear (int8_tensor.py:191-230 + :305-359, oracle/int8_ref.py): per-row symmetric int8 cast of x,
 * int32 dot products, t = bf16(c * sx), y = bf16(t * sw). */
void ao_ref_int8_dynamic_linear(const uint16_t* x, const int8_t* wq, const float* ws, uint16_t* y, int64_t M, int64_t N, int64_t K) {
  const float eps = bf16_round(1.1920928955078125e-07f);
#pragma omp parallel
  {
    int8_t* xq = (int8_t*)__builtin_alloca((size_t)K);
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
      float mn = 0.f, mx = 0.f;
      for (int64_t k = 0; k < K; ++k) { const float v = bf16_to_f32(x[m * K + k]); mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
      const float amax = -mn > mx ? -mn : mx;
      float sx = bf16_round(amax / 127.5f);
      sx = sx > eps ? sx : eps;
      const float inv = 1.0f / sx;
      for (int64_t k = 0; k < K; ++k) {
        float q = rintf(bf16_to_f32(x[m * K + k]) * inv);
        q = q < -128.f ? -128.f : (q > 127.f ? 127.f : q);
        xq[k] = (int8_t)q;
      }
      for (int64_t n = 0; n < N; ++n) {
        int32_t c = 0;
        const int8_t* w = wq + n * K;
        for (int64_t k = 0; k < K; ++k) c += (int32_t)xq[k] * (int32_t)w[k];
        const float t = bf16_round((float)c * sx);
        y[m * N + n] = f32_to_bf16(t * ws[n]);
      }
    }
  }
}

/* Float8Tensor rowwise dynamic linear (float8_tensor.py:167-253, float8/inference.py:104-123, oracle/fp8_ref.py):
 * sx = bf16(amax / 448), codes = e4m3(clamp(x / sx)), y = bf16(sum * sx * sw). */
void ao_ref_fp8_rowwise_linear(const uint16_t* x, const uint8_t* wq, const float* ws, uint16_t* y, int64_t M, int64_t N, int64_t K) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = e4m3_to_f32((uint8_t)i);
#pragma omp parallel
  {
    float* xf = (float*)__builtin_alloca((size_t)K * sizeof(float));
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
      float amax = 0.f;
      for (int64_t k = 0; k < K; ++k) { const float v = fabsf(bf16_to_f32(x[m * K + k])); amax = v > amax ? v : amax; }
      const float sx = bf16_round(amax / 448.0f);
      for (int64_t k = 0; k < K; ++k) {
        float t = bf16_to_f32(x[m * K + k]) / sx;
        t = t < -448.f ? -448.f : (t > 448.f ? 448.f : t);
        xf[k] = lut[f32_to_e4m3(t)];
      }
      for (int64_t n = 0; n < N; ++n) {
        double acc = 0.0;
        const uint8_t* w = wq + n * K;
        for (int64_t k = 0; k < K; ++k) acc += (double)xf[k] * (double)lut[w[k]];
        y[m * N + n] = f32_to_bf16((float)(acc * (double)sx * (double)ws[n]));
      }
    }
  }
}

/* to_mx RCEIL scale byte of one 32-block (mx_tensor.py:111-129, oracle/mx_ref.py) */
static inline uint8_t e8m0_rceil(float amax) {
  const float d = amax * (1.0f / 448.0f);
  const uint32_t b = f32_bits(d);
  if ((b & 0x7f800000u) == 0x7f800000u) return 255;
  const int be = (int)((b >> 23) & 0xFF);
  const uint32_t mant = b & 0x7FFFFFu;
  const int up = (be == 0) ? (mant > 0x400000u) : (mant != 0);
  return (uint8_t)(be + up);
}
static inline float e8m0_reciprocal(uint8_t e) {
  const uint32_t r = (uint32_t)((254 - (int)e) & 0xFF);
  if (r == 0) return bits_f32(0x00400000u);
  if (r == 255) return bits_f32(0x7F800001u);
  return bits_f32(r << 23);
}
static inline float e8m0_value(uint8_t e) { return e == 255 ? NAN : ldexpf(1.0f, (int)e - 127); }

/* _to_mxfp8_then_scaled_grouped_mm forward, emulated path (mxfp8_grouped_mm.py:552-594, :959-1023, oracle/mx_ref.py):
 * A rows cast to MXFP8 (RCEIL), both operands dequantised to bf16, bf16 grouped matmul with fp32 accumulation. */
void ao_ref_mxfp8_grouped_mm(const uint16_t* a, const uint8_t* wq, const uint8_t* wscale, const int32_t* offs, uint16_t* y, int64_t M,
                             int64_t E, int64_t N, int64_t K) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = e4m3_to_f32((uint8_t)i);
  const int64_t KB = K / 32;
#pragma omp parallel
  {
    float* ad = (float*)__builtin_alloca((size_t)K * sizeof(float));
#pragma omp for schedule(dynamic, 1)
    for (int64_t m = 0; m < M; ++m) {
      int64_t e = 0;
      while (e < E && m >= offs[e]) ++e;
      if (e >= E) { for (int64_t n = 0; n < N; ++n) y[m * N + n] = 0; continue; }
      for (int64_t kb = 0; kb < KB; ++kb) {
        float amax = 0.f;
        for (int k = 0; k < 32; ++k) { const float v = fabsf(bf16_to_f32(a[m * K + kb * 32 + k])); amax = v > amax ? v : amax; }
        const uint8_t se = e8m0_rceil(amax);
        const float r = e8m0_reciprocal(se), sv = e8m0_value(se);
        for (int k = 0; k < 32; ++k) ad[kb * 32 + k] = bf16_round(lut[f32_to_e4m3(bf16_to_f32(a[m * K + kb * 32 + k]) * r)] * sv);
      }
      const uint8_t* we = wq + e * N * K;
      const uint8_t* wse = wscale + e * N * KB;
      for (int64_t n = 0; n < N; ++n) {
        float acc = 0.f;
        for (int64_t kb = 0; kb < KB; ++kb) {
          const float sv = e8m0_value(wse[n * KB + kb]);
          float part = 0.f;
          for (int k = 0; k < 32; ++k) part += ad[kb * 32 + k] * bf16_round(lut[we[n * K + kb * 32 + k]] * sv);
          acc += part;
        }
        y[m * N + n] = f32_to_bf16(acc);
      }
    }
  }
}
