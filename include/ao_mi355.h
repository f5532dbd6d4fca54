/*
 * ao_mi355.h -- C ABI of the MI355X (gfx950) low-bit linear backend.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  All
 * pointers are DEVICE pointers (HBM) unless a name ends in `_host`; `stream` is
 * a hipStream_t passed as void* (NULL = the default stream).  Every entry point
 * is asynchronous with respect to the host: it validates its arguments on the
 * host, enqueues kernels on `stream` and returns.  Inputs are borrowed and never
 * written; outputs must be caller-allocated (the Python host layer in
 * ao_amd/ops.py allocates them the way the reference ops allocate-and-return).
 *
 * Return value: AO_OK (0) or a negative AO_ERR_* code; ao_last_error() gives
 * the thread-local human readable message (the host layer raises RuntimeError /
 * ValueError from it, mirroring STD_TORCH_CHECK in
 * torchao/csrc/cuda/mx_kernels/mxfp8_extension.cpp:95-134).  The library never
 * calls exit().
 *
 * Library state: the only device memory the library owns is a split-K scratch
 * buffer per (device, stream) that launches split-K shapes (batched int4 on
 * narrow N, 8-bit linears with few output tiles).  It is allocated on that
 * stream's first such launch and grows in powers of two (8 MiB .. 128 MiB) to the
 * largest request seen on the stream; at most 8 streams per device hold one at a
 * time -- a ninth evicts the least recently used one after a device
 * synchronise.  Allocation, growth and eviction are impossible inside stream
 * capture: call the op once on the stream with its largest shape before
 * capturing, and re-capture graphs of a stream that lost its buffer.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the torchao checkout, lines from the 0.19.0 snapshot).
 * bf16 tensors are passed as uint16_t*, fp8 e4m3fn / e8m0 as uint8_t*.
 */
#ifndef AO_MI355_H
#define AO_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AO_OK 0
#define AO_ERR_INVALID_ARGUMENT (-1) /* shape / alignment / enum not supported */
#define AO_ERR_NULL_POINTER (-2)
#define AO_ERR_HIP (-3) /* a HIP runtime call failed; message has hipGetErrorString */

#define AO_MI355_ABI_VERSION 2 /* 2: ao_moe_a2a_v takes max_in_rows; peer memory + MAX all-reduce entry points */

/* Library / ABI version and last error of the calling thread. */
int ao_abi_version(void);
const char* ao_last_error(void);

/* Pre-size the split-K scratch buffer of `stream` (see "Library state" above) outside stream capture: afterwards every op on that stream is
 * capture-safe whatever its shape, without a warm-up call per shape.  bytes <= 0 reserves the cap (128 MiB); smaller requests are rounded up
 * to the next power of two >= 8 MiB.  The reference's ops own no workspace (SURVEY 8(b) "stateless"); its split-K lives inside hipBLASLt /
 * cuBLAS, which pre-allocate theirs per handle the same way (ATen's cublas workspace, CUDABlas / CublasHandlePool). */
int ao_splitk_reserve(void* stream, int64_t bytes);

/* Per-launch kernel timing for benchmarks: after ao_prof_enable(n) the next n
 * kernel launches of this library are bracketed with HIP extension events that
 * timestamp the dispatch itself (begin/end of the kernel, no launch gaps).
 * ao_prof_collect() waits for them, writes the durations in milliseconds in
 * launch order to a HOST array and disables profiling.  Not thread-safe; do not
 * use during graph capture. */
int ao_prof_enable(int max_records);
int ao_prof_collect(float* ms_out_host, int capacity, int* n_out_host);

/* ------------------------------------------------------------------------- *
 * int4 weight-only, tinygemm "tile packed to 4d" format
 * ------------------------------------------------------------------------- */

/* Replaces aten::_convert_weight_to_int4pack(Tensor self, int innerKTiles) as
 * called at torchao/quantization/quantize_/workflows/int4/
 * int4_tile_packed_to_4d_tensor.py:202.
 *   w_u8   uint8 [N][K/2], even k in the high nibble (:199-201)
 *   qdata  int32 [N/8][K/(inner_k_tiles*16)][32][inner_k_tiles/2], gfx950 tile
 *          order (one wavefront owns a 16(n) x 128(k) tile)
 * Requires N % 16 == 0, K % (inner_k_tiles*16) == 0, inner_k_tiles == 8. */
int ao_int4_convert_weight_to_int4pack(const uint8_t* w_u8, int32_t* qdata,
                                       int64_t N, int64_t K, int inner_k_tiles,
                                       void* stream);

/* Inverse of the above (bit-exact): qdata -> uint8 [N][K/2].  Used by
 * Int4TilePackedTo4dTensor.dequantize()/slicing tests; no reference op
 * (the reference unpacks by multiplying with an identity matrix,
 * torchao/quantization/utils.py:198-226). */
int ao_int4_unpack_int4pack(const int32_t* qdata, uint8_t* w_u8, int64_t N,
                            int64_t K, int inner_k_tiles, void* stream);

/* Replaces aten::_weight_int4pack_mm(Tensor self, Tensor mat2, int qGroupSize,
 * Tensor qScaleAndZeros) as called at int4_tile_packed_to_4d_tensor.py:287.
 *   x               bf16 [M][K]
 *   qdata           int32 4-D as above (N x K logical)
 *   scale_and_zero  bf16 [K/group_size][N][2]   (torchao/quantization/utils.py:299-313)
 *   y               bf16 [M][N]
 * y[m][n] = sum_k x[m][k] * bf16(bf16((q[n][k]-8) * s[k/g][n]) + z[k/g][n]),
 * fp32 accumulation (the dequant->bf16-matmul oracle, quant_primitives.py:999-1007).
 * group_size in {32, 64, 128, 256}; N % 16 == 0; K % 128 == 0; K % group_size == 0. */
int ao_int4_weight_int4pack_mm(const uint16_t* x, const int32_t* qdata,
                               const uint16_t* scale_and_zero, uint16_t* y,
                               int64_t M, int64_t N, int64_t K, int group_size,
                               void* stream);

/* Dequantize the packed weight to bf16 [N][K] with the oracle's rounding
 * sequence.  Replaces groupwise_affine_dequantize_tensor
 * (torchao/quantization/utils.py:445-455) for the packed format. */
int ao_int4_dequantize(const int32_t* qdata, const uint16_t* scale_and_zero,
                       uint16_t* w_bf16, int64_t N, int64_t K, int group_size,
                       void* stream);

/* Fused weight preparation: _choose_qparams_affine_tinygemm +
 * _quantize_affine_tinygemm + nibble pack + _convert_weight_to_int4pack +
 * pack_tinygemm_scales_and_zeros (int4_tile_packed_to_4d_tensor.py:129-236,
 * quant_primitives.py:1299-1335,577-599; utils.py:299-313) in one pass.
 *   w  bf16 [N][K] (already padded: N % 16 == 0, K % 128 == 0, K % g == 0). */
int ao_int4_quantize_tinygemm(const uint16_t* w, int32_t* qdata,
                              uint16_t* scale_and_zero, int64_t N, int64_t K,
                              int group_size, void* stream);

/* Launch-shape override for tuning sweeps (bench/tools only): waves per
 * workgroup (0 = heuristic) and an A/B mode of the int4 mm (0 = product dispatch).  Every mode this library honours computes the
 * SAME result as the product (other ring depths, tile shapes, K splits, trace stamps); the ablation builds that drop parts of the
 * kernel -- and so return wrong numbers -- exist only in the laboratory build (`python -m ao_amd.build --lab` ->
 * tools/bin/_C_mi355_lab.so, compiled with -DAO_LAB; tools select it through AO_MI355_LIB).  Thread-local. */
int ao_int4_set_tuning(int waves_per_block, int mode);
/* Profiling only: 0 = product dispatch of the 8-bit GEMMs (LDS-DMA staged kernel when K % 128 == 0),
 * 1 = force the register-staged kernel, 2 / 4 / 8 = force the LDS-DMA kernel with 128x128, 256x128 (4 waves), 256x256 (8 waves) tiles,
 * 32 = the phase-interleaved 256x256 kernel, 33 = its 256x128 form (gemm8_p8h_kernel); 100 / 101 / 102 = the fp8 weight-streaming kernel never / always / always with 64-column tiles; 103 = its round-3 wave arrangement (1 x 8);
 * MXFP8 grouped mm: 110 always the LDS-staged kernels, 111 never (A-stationary / per-tile kernels); decode-size groups: 113 one workgroup
 * per tile instead of the stream-K kernel, 129 the stream-K kernel's per-step-scales form (what K % 512 != 0 takes) on every K
 * (ao_amd/csrc/rb8_kernels.hip); the decode kernel (dec8_kernel): 201 .. 208 its ring depth, 290 half-line loads, 291 / 292 never /
 * always 8-row tiles, 293 the round-4 bound of 64 KiB of activation codes (product since round 6: the CU's whole LDS), 299 never;
 * 300 / 301 / 31S the register-ring mid-M kernel never / wherever the shape allows / with S K parts.  Thread-local, like ao_int4_set_tuning. */
int ao_gemm8_set_variant(int variant);
/* Profiling only, key / value (every setting computes the SAME result as the product; 0 = product rule; thread-local):
 *   key 1  column-tile width of the rowwise weight-streaming kernel (rb8_kernel): 32, 64 or 128
 *   key 2  its K parts (1 .. 16)
 *   key 3  slab rows of rb8_kernel: 64 or 128 at any M (product: the cost model's pick above 64 rows, 64 up to 64)
 *   key 4  tile rows an XCD's workgroups of gemm8_p8_kernel walk together (product: 4)
 *   key 5  timing probes of the TRACED build of rb8_kernel only (ao_int4_set_trace set; results are wrong): bit 0 no MFMAs, 1 no fragment
 *          reads, 2 no weight DMAs, 3 no activation DMAs, 4 DMA wait + barrier on even steps only -- the product build ignores it
 *   key 6  the persistent form of the 256 x 256 GEMM (gemm8_p8p_kernel): 1 never, 2 wherever the shape allows
 *   key 7  K parts of gemm8_p8h_kernel (1 .. 16, clamped to what fits one round of the chip and the split-K workspace)
 *   key 8  loop form of gemm8_p8h_kernel, laboratory build only (the product build ignores it)
 *   key 9  the MXFP8 stream-K kernel's meeting, A/B: bit 0 the head piece's ticket after the loop, bit 1 no early read of the tail ticket
 *          (3 = the round-3 protocol)
 * An unknown key is an error.  DESIGN.md 4.5h. */
int ao_gemm8_set_tuning(int key, int value);
/* Name of the kernel ao_int4_weight_int4pack_mm launches for this problem (product dispatch, no
 * tuning override): what a profiler's kernel table should be matched against.  Static string. */
const char* ao_int4_mm_kernel_name(int64_t M, int64_t N, int64_t K, int group_size);
/* Which kernel ao_fp8_scaled_mm (int8 = 0) / ao_int8_scaled_mm (int8 = 1) dispatches a shape to: "dec8_kernel" (M <= 16), "mid8_kernel",
 * "stream8_kernel", "rb8_kernel" (up to 256 tiles of 128 x 128), "gemm8_p8_kernel" / "gemm8_dma_kernel<...>" / "gemm8_kernel" (tiled), or
 * "invalid".  Host logic only (no launch): bench / tests label their measurements with it.  DESIGN.md 4.4-4.5g. */
const char* ao_gemm8_kernel_name(int int8, int64_t M, int64_t N, int64_t K);
/* The launch shape behind that name: column-tile width and K parts of the product dispatch for the shape (rb8_kernel: the cost model's
 * pick; gemm8_p8h_kernel: 128 columns and 1 .. 4 parts; others: their tile width, one part).  Host logic only.  DESIGN.md 4.5h. */
int ao_gemm8_plan(int int8, int64_t M, int64_t N, int64_t K, int* tile_cols, int* k_parts);
/* The rows of that launch's tile: rb8_kernel's slab height (64 up to 64 rows; 64 or 128 beyond, by the cost model -- round 6: 64-row slabs
 * cut M instead of K where the fixed costs of a launch outweigh its loop), 256 for the 256 x 128 / 256 x 256 GEMMs, 128 / 16 otherwise.
 * Host logic only.  DESIGN.md 4.5. */
int ao_gemm8_plan_rows(int int8, int64_t M, int64_t N, int64_t K, int* tile_rows);
/* Which form of fp8_int4_mm_kernel ao_fp8_int4_linear launches for a shape: "<m-tiles x n-tiles>" of 16 x 16 per workgroup -- "<1x1>" up to
 * 16 rows, "<2x1>" / "<2x2>" beyond (round 5: the weights stream once per 32 rows, the staged activations serve 32 columns), or "invalid".
 * Host logic only.  DESIGN.md 4.9. */
const char* ao_fp8_int4_kernel_name(int64_t M, int64_t N, int64_t K, int group_size);
/* Profiling only: device buffer [workgroups][16] of s_memtime stamps written by the trace builds of the batched
 * kernels (int4: tuning mode 65S; fp8 rowwise mid-M kernel: whenever the pointer is set); NULL disables. */
int ao_int4_set_trace(unsigned long long* trace_dev);

/* ------------------------------------------------------------------------- *
 * int8 dynamic activation x int8 weight
 * ------------------------------------------------------------------------- */

/* Per-row symmetric int8 quantization of a bf16 matrix: replaces
 * Int8Tensor.from_hp(x, PerRow) = choose_qparams_affine(SYMMETRIC, eps=fp32 eps)
 * + quantize_affine (torchao/quantization/quantize_/workflows/int8/
 * int8_tensor.py:176-248; quant_primitives.py:1534-1583,463-485).
 *   x bf16 [M][K] -> q int8 [M][K], scale fp32 [M] */
int ao_int8_quantize_rowwise(const uint16_t* x, int8_t* q, float* scale,
                             int64_t M, int64_t K, void* stream);

/* Replaces _int_scaled_matmul/safe_int_mm -> aten::_int_mm plus the scale
 * epilogue of the Int8Tensor linear (int8/kernels.py:114-144, int8_tensor.py:
 * 315-359):  y = bf16( bf16( (xq @ wq^T)_i32 * x_scale[m] ) * w_scale[n] ) (+bias).
 *   xq int8 [M][K]; wq int8 [N][K] (row-major weight, i.e. the reference's
 *   weight.qdata before its .contiguous().t()); x_scale fp32 [M];
 *   w_scale fp32 [N]; bias bf16 [N] or NULL; y bf16 [M][N].  K % 16 == 0. */
int ao_int8_scaled_mm(const int8_t* xq, const float* x_scale, const int8_t* wq,
                      const float* w_scale, const uint16_t* bias, uint16_t* y,
                      int64_t M, int64_t N, int64_t K, void* stream);

/* Plain aten::_int_mm (int8/kernels.py:38-40,70): c int32 [M][N] = a[M][K] @ b_t[N][K]^T. */
int ao_int8_int_mm(const int8_t* a, const int8_t* b_t, int32_t* c, int64_t M,
                   int64_t N, int64_t K, void* stream);

/* Asymmetric per-row activation quantization: Int8Tensor.from_hp(x, PerRow(), mapping_type=ASYMMETRIC), the activation side
 * of Int8DynamicActivationInt8WeightConfig(act_mapping_type=ASYMMETRIC) (int8_tensor.py:191-236; choose_qparams_affine's
 * ASYMMETRIC branch quant_primitives.py:1568-1574; quantize_affine :463-485), bf16 input:
 *   mn = min(row, 0), mx = max(row, 0); scale = max(bf16(bf16(mx - mn) / 255), f32_eps);
 *   zero_point = clamp(-128 - rint(bf16(mn / scale)), -128, 127); q = clamp(rint(x / scale) + zero_point, -128, 127)
 *   x bf16 [M][K] -> q int8 [M][K], scale fp32 [M], zero_point int8 [M] */
int ao_int8_quantize_rowwise_asym(const uint16_t* x, int8_t* q, float* scale, int8_t* zero_point,
                                  int64_t M, int64_t K, void* stream);
/* Static activation quantization: Int8Tensor.from_hp(x, granularity, mapping_type, scale=..., zero_point=...) with the qparams GIVEN
 * (Int8StaticActivationInt8WeightConfig, quant_api.py:919-1012; int8_tensor.py:212-231): q = clamp(rint(x / scale) + zp, -128, 127).
 *   x bf16 [M][K]; scale fp32 [1] (per_row = 0) or [M] (per_row = 1); zero_point int8, same shape, or NULL (symmetric) -> q int8 [M][K] */
int ao_int8_quantize_static(const uint16_t* x, const float* scale, const int8_t* zero_point, int per_row,
                            int8_t* q, int64_t M, int64_t K, void* stream);
/* rowsum(W_int8) of the zero-point correction (int8_tensor.py:326 `weight_tensor.qdata.sum(dim=-1)`): q int8 [N][K] ->
 * sums int32 [N].  K % 16 == 0.  Computed once per weight by the host mirror. */
int ao_int8_row_sums(const int8_t* q, int32_t* sums, int64_t N, int64_t K, void* stream);
/* The Int8Tensor linear's epilogue with an asymmetric activation (int8_tensor.py:305-346) over the int32 accumulator of
 * ao_int8_int_mm:  t = bf16(f32(acc) * x_scale[m]);  t = bf16(t - bf16((zp[m] * x_scale[m]) * w_row_sums[n]));
 *   y = bf16(t * w_scale[n] (+ bias[n])). */
int ao_int8_scale_epilogue_asym(const int32_t* acc, const float* x_scale, const int8_t* x_zero_point,
                                const int32_t* w_row_sums, const float* w_scale, const uint16_t* bias,
                                uint16_t* y, int64_t M, int64_t N, void* stream);

/* ------------------------------------------------------------------------- *
 * float8 (OCP e4m3fn) rowwise
 * ------------------------------------------------------------------------- */

/* Replaces Float8Tensor.from_hp(x, e4m3, PerRow): _choose_scale_float8 +
 * _quantize_affine_float8 (float8_tensor.py:167-253; quant_primitives.py:
 * 2172-2212,2271-2287): scale[r] = f32(bf16(amax_r / 448)), q = e4m3(clamp(x/scale)).
 *   x bf16 [M][K] -> q e4m3fn [M][K], scale fp32 [M] */
int ao_fp8_quantize_rowwise(const uint16_t* x, uint8_t* q, float* scale,
                            int64_t M, int64_t K, void* stream);

/* Replaces aten::_scaled_mm with rowwise scales as called from
 * addmm_float8_unwrapped_inference (torchao/float8/inference.py:86-123):
 *   y = bf16( (a @ b^T)_f32 * scale_a[m] * scale_b[n] + bias[n] )
 *   a e4m3fn [M][K] row-major; b e4m3fn [N][K] row-major (== the reference's
 *   column-major [K][N] mat2); scale_a fp32 [M]; scale_b fp32 [N];
 *   bias bf16 [N] or NULL; y bf16 [M][N].  K % 16 == 0, N % 16 == 0. */
int ao_fp8_scaled_mm(const uint8_t* a, const uint8_t* b, const float* scale_a,
                     const float* scale_b, const uint16_t* bias, uint16_t* y,
                     int64_t M, int64_t N, int64_t K, void* stream);

/* HQQ qparams + codes for the tinygemm format: Int4TilePackedTo4dTensor.from_hp(..., HQQ)
 * (int4_tile_packed_to_4d_tensor.py:149-168 -> quant_primitives.py:1891-1997, optimizer :1797-1866 in float16 as on a GPU).
 *   w bf16 [N][K] -> nibble_bytes uint8 [N][K/2] (even k in the HIGH nibble: the input of
 *   ao_int4_convert_weight_to_int4pack), scale_and_zero bf16 [K/g][N][2].
 *   workspace: ao_int4_hqq_workspace_bytes(N, K, g) device bytes.  Host-synchronous like the reference: the optimizer's
 *   early stop reads a global error every iteration (<= 20 stream synchronisations); not capturable in a graph. */
int64_t ao_int4_hqq_workspace_bytes(int64_t N, int64_t K, int group_size);
int ao_int4_quantize_hqq(const uint16_t* w, uint8_t* nibble_bytes, uint16_t* scale_and_zero, void* workspace,
                         int64_t N, int64_t K, int group_size, void* stream);

/* Int4Tensor.from_hp, PLAIN packing (quantize_/workflows/int4/int4_tensor.py:130-186): the arithmetic of mslk's
 * int4_row_quantize_zp / int4_row_quantize + pack_int4 (un-vendored; restated by the reference at
 * quantization/qat/fake_quantizer.py:148-190 and prototype/gptq/api.py:167-221), fp32 math:
 *   symmetric = 0: scale = max(max - min, 1e-6) / 15, zero = min + 8 scale, q = clamp(rint((w - min) / scale), 0, 15) - 8
 *   symmetric = 1: scale = max(max|w| / 8, 1e-6), zero = 0, q = clamp(rint(w / scale), -8, 7)   (fp8-activation flavour)
 *   w bf16 [N][K] -> qdata uint8 [N][K/2] (even k in the LOW nibble), scale / zero_point bf16 [K/g][N]. */
int ao_int4_plain_quantize(const uint16_t* w, uint8_t* qdata, uint16_t* scale, uint16_t* zero_point,
                           int64_t N, int64_t K, int group_size, int symmetric, void* stream);

/* ------------------------------------------------------------------------- *
 * MXFP8 (e4m3 elements, E8M0 scale per 32 along the contraction dim)
 * ------------------------------------------------------------------------- */

#define AO_MX_SCALE_FLOOR 0
#define AO_MX_SCALE_RCEIL 1

/* Replaces torchao::mxfp8_quantize (rowwise 1x32 cast; schema
 * torchao/prototype/mx_formats/kernels.py:1022-1026; semantics to_mx,
 * torchao/prototype/mx_formats/mx_tensor.py:228-409).
 *   x bf16 [R][C] (C % 32 == 0) -> q e4m3fn [R][C], scale e8m0 [R][C/32]. */
int ao_mxfp8_quantize_rowwise(const uint16_t* x, uint8_t* q, uint8_t* scale_e8m0,
                              int64_t R, int64_t C, int scaling_mode,
                              void* stream);

/* The colwise half of torchao::mxfp8_quantize (32 x 1 blocks: one scale per 32 ROWS of a column;
 * csrc/cuda/mx_kernels/mxfp8_quantize.cuh:460-820, mxfp8_extension.cpp:160-175; == to_mx(x.t()).t()).
 *   x bf16 [R][C] (R % 32 == 0, C % 32 == 0) -> q_t e4m3fn [C][R] (the column-major data the reference
 *   returns as a {R, C} tensor with strides {1, R}), scale e8m0 [R/32][C] (its {C, R/32} tensor with strides {1, C}). */
int ao_mxfp8_quantize_colwise(const uint16_t* x, uint8_t* q_t, uint8_t* scale_e8m0,
                              int64_t R, int64_t C, int scaling_mode, void* stream);

/* The 3-D (per-expert) form: mxfp8_quantize_cuda_3d with (scale_block_dim1, scale_block_dim2) = (32, 1), logical scales
 * (torchao/prototype/moe_training/kernels/mxfp8/quant.py:1413-1440; csrc/cuda/mx_kernels/mxfp8_quantize.cuh:822-1290) --
 * every [R][C] matrix of the batch cast on its own, "column-major-per-expert" data.
 *   x bf16 [E][R][C] -> q_t e4m3fn [E][C][R], scale e8m0 [E][R/32][C]. */
int ao_mxfp8_quantize_colwise_3d(const uint16_t* x, uint8_t* q_t, uint8_t* scale_e8m0,
                                 int64_t E, int64_t R, int64_t C, int scaling_mode, void* stream);

/* Replaces aten::_scaled_grouped_mm as called from _compute_fwd_sm100
 * (torchao/prototype/moe_training/mxfp8_grouped_mm.py:541), with the numerics of
 * _emulated_mxfp8_scaled_grouped_mm_2d_3d (:959-1023):
 *   out[offs[e-1]:offs[e]] = dq(a_rows) @ dq(b[e])^T,  dq = fp8 * 2^(scale-127)
 *   a e4m3 [M_total][K]; a_scale e8m0 [M_total][K/32];
 *   b e4m3 [E][N][K] (each expert row-major [N][K]); b_scale e8m0 [E][N][K/32];
 *   offs int32 [E] cumulative group ends; out bf16 [M_total][N].
 * Scale layout is plain row-major (CDNA4 scaled-MFMA takes scales in VGPRs; the
 * cuBLAS 128x4 "blocked" swizzle is not read by any GEMM here; ao_mx_block_rearrange_2d_m_groups writes it as a data format). */
int ao_mxfp8_grouped_mm(const uint8_t* a, const uint8_t* a_scale,
                        const uint8_t* b, const uint8_t* b_scale,
                        const int32_t* offs, uint16_t* out, int64_t M_total,
                        int64_t N, int64_t K, int64_t E, void* stream);

/* _to_mxfp8_then_scaled_grouped_mm's forward in ONE launch (torchao/prototype/moe_training/mxfp8_grouped_mm.py:330-371, 552-594: to_mx(A, 32,
 * scaling_mode) followed by the grouped mm above; SURVEY.md 8 f1 for the MX format; native analogue of the cast: torchao/csrc/cuda/mx_kernels/
 * mxfp8_quantize.cuh:460-820).  a is the BF16 activation matrix [M_total][K]; the 1 x 32 cast runs inside the kernel's A-fill with the
 * arithmetic of ao_mxfp8_quantize_rowwise (same codes, same scales, so out is bit-identical to cast + ao_mxfp8_grouped_mm).
 * Decode-size groups only: ao_mxfp8_grouped_mm_dyn_fits(M_total, N, K, E) (host logic, no launch) is 1 when M_total <= 48 E, E <= 64,
 * K % 512 == 0, N % 16 == 0; other shapes return AO_ERR_INVALID_ARGUMENT -- cast and call ao_mxfp8_grouped_mm.  a and b_scale 16-byte aligned. */
int ao_mxfp8_grouped_mm_dyn_fits(int64_t M_total, int64_t N, int64_t K, int64_t E);
int ao_mxfp8_grouped_mm_dyn(const uint16_t* a, const uint8_t* b, const uint8_t* b_scale, const int32_t* offs, uint16_t* out,
                            int64_t M_total, int64_t N, int64_t K, int64_t E, int scaling_mode, void* stream);

/* Two expert-weight tensors of ONE shape [E][N][K] against the same activations in ONE launch: an MoE layer's w1 and w3 (the reference's
 * experts compute x @ w1 and x @ w3 by two calls of _to_mxfp8_then_scaled_grouped_mm, mxfp8_grouped_mm.py:56-239, casting x twice).
 * out1 / out3 bf16 [M_total][N] hold the values of two single calls: the same bits where the stream-K shares cut the tiles at the same k steps
 * (a cut tile's pieces are added in k order; another cut is another fp32 summation order -- single elements one bf16 ulp apart, the same from
 * launch to launch).  _dyn_pair: a is BF16, cast fused (as ao_mxfp8_grouped_mm_dyn);
 * _pair: a / a_scale are the caller's e4m3 codes / E8M0 scales.  Shapes: ao_mxfp8_grouped_mm_pair_fits (host logic; the _dyn conditions
 * with twice the tiles). */
int ao_mxfp8_grouped_mm_pair_fits(int64_t M_total, int64_t N, int64_t K, int64_t E);
int ao_mxfp8_grouped_mm_dyn_pair(const uint16_t* a, const uint8_t* b1, const uint8_t* b1_scale, const uint8_t* b3, const uint8_t* b3_scale,
                                 const int32_t* offs, uint16_t* out1, uint16_t* out3, int64_t M_total, int64_t N, int64_t K, int64_t E,
                                 int scaling_mode, void* stream);
int ao_mxfp8_grouped_mm_pair(const uint8_t* a, const uint8_t* a_scale, const uint8_t* b1, const uint8_t* b1_scale, const uint8_t* b3,
                             const uint8_t* b3_scale, const int32_t* offs, uint16_t* out1, uint16_t* out3, int64_t M_total, int64_t N,
                             int64_t K, int64_t E, void* stream);

/* Float8Tensor's aten::_grouped_mm with rowwise scales (float8_tensor.py:1085-1122 -> scaled_grouped_mm, RowWise recipes):
 *   out[offs[e-1]:offs[e]] = bf16( (a_rows @ b[e]^T)_f32 * scale_a[m] * scale_b[e][n] )
 *   a e4m3 [M_total][K]; scale_a fp32 [M_total]; b e4m3 [E][N][K]; scale_b fp32 [E][N]; offs int32 [E]; out bf16 [M_total][N].
 *   Rows past offs[E-1] are not written. */
int ao_fp8_grouped_mm(const uint8_t* a, const float* scale_a, const uint8_t* b, const float* scale_b,
                      const int32_t* offs, uint16_t* out, int64_t M_total, int64_t N, int64_t K,
                      int64_t E, void* stream);

/* ------------------------------------------------------------------------- *
 * MoE token-group padding (glue either side of the MXFP8 grouped GEMM)
 * ------------------------------------------------------------------------- */

/* Rows of the padded buffer: align_up(num_tokens + num_groups * alignment, alignment) -- the upper bound the
 * reference allocates so that no host sync is needed (quant.py:414-420).  Host-only helper. */
int64_t ao_moe_padded_rows(int64_t num_tokens, int64_t num_groups, int alignment);

/* Replaces torchao::fused_pad_token_groups (schema torchao/prototype/moe_training/kernels/mxfp8/quant.py:1244-1246;
 * semantics torch_pad_token_groups, quant.py:368-430).
 *   inputs  [num_tokens][dim] bf16 or fp32 (elem_bytes 2 or 4); offsets int32 [num_groups] cumulative group ends;
 *   padded  [ao_moe_padded_rows(...)][dim], every row written (groups copied to aligned starts, the rest zero);
 *   padded_starts / padded_ends int32 [num_groups]. */
int ao_moe_pad_token_groups(const void* inputs, const int32_t* offsets, void* padded,
                            int32_t* padded_starts, int32_t* padded_ends, int64_t num_tokens,
                            int64_t dim, int elem_bytes, int64_t num_groups, int alignment,
                            void* stream);

/* Replaces torchao::fused_unpad_token_groups (schema quant.py:1319-1321; semantics torch_unpad_token_groups,
 * quant.py:433-480): out[t] = padded[padded_starts[g] + t - offsets[g-1]] for the group g holding token t.
 * Tokens past offsets[num_groups-1] (the reference raises after a host sync) are written as zeros. */
int ao_moe_unpad_token_groups(const void* padded, const int32_t* offsets,
                              const int32_t* padded_starts, void* out, int64_t num_tokens,
                              int64_t dim, int elem_bytes, int64_t num_groups, void* stream);

/* Replaces torchao::mx_block_rearrange_2d_M_groups (schema torchao/prototype/moe_training/kernels/mxfp8/quant.py:969-973; host wrapper
 * csrc/cuda/mx_kernels/mxfp8_extension.cpp:178-300; semantics torch_to_blocked_2d_M_groups, quant.py:136-196, over to_blocked,
 * prototype/mx_formats/utils.py:31-72): E8M0 scales [rows][cols] whose rows are grouped by `offsets` (int32 [num_groups] cumulative
 * ends) -> the reference's "128 x 4 blocked" layout, every group padded to 128-row blocks:
 *   out [ao_mx_blocked_rows(rows, num_groups)][4 ceil(cols / 4)] bytes, every byte written (zero where no scale lands), 16-byte aligned;
 *   group g starts at out row sum_{h<g} 128 ceil(size_h / 128); tile (rb, cb) of a group at + (rb ncb + cb) 512 bytes, element (r, c)
 *   of a tile at (r % 32) 16 + (r / 32) 4 + c.
 * The GEMMs of this library take row-major scales; this is a data-format op for callers that hold the blocked layout.
 * ao_mx_blocked_rows = rows + 128 num_groups, the reference's no-host-sync upper bound (mxfp8_extension.cpp:221). */
int64_t ao_mx_blocked_rows(int64_t rows, int64_t num_groups);
int ao_mx_block_rearrange_2d_m_groups(const uint8_t* scales, const int32_t* offsets, uint8_t* out, int64_t rows, int64_t cols,
                                      int64_t num_groups, void* stream);
/* to_blocked of one matrix (prototype/mx_formats/utils.py:31-72): out [128 ceil(rows / 128)][4 ceil(cols / 4)] bytes, the same tiles. */
int ao_mx_to_blocked(const uint8_t* scales, uint8_t* out, int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------- *
 * One-shot all-reduce over peer-mapped buffers (TP row-parallel linears at decode sizes)
 * ------------------------------------------------------------------------- */

/* Peer-visible device memory for the flag blocks and staging buffers below, and its IPC handles.  The caching allocator's memory is
 * coarse-grained: a flag a REMOTE GPU writes may be served to the spinning owner from its local L2 for ever (two processes on ONE GPU
 * share that L2 and cannot show it).  kind 0 = hipDeviceMallocUncached (flag blocks), kind 1 = hipDeviceMallocFinegrained (staging);
 * zero-filled.  ao_peer_export writes ao_peer_handle_bytes() (64) opaque bytes any process of the node can ao_peer_import (maps the
 * peer's allocation, enabling peer access lazily); ao_peer_close unmaps an imported pointer, ao_peer_free releases an own one.  All
 * HOST-side, synchronous, not capturable. */
int ao_peer_alloc(void** ptr_out_host, int64_t bytes, int kind);
int ao_peer_free(void* ptr);
int ao_peer_handle_bytes(void);
int ao_peer_export(void* ptr, void* handle_out_host);
int ao_peer_import(const void* handle_host, void** ptr_out_host);
int ao_peer_close(void* imported_ptr);
/* How long a collective kernel waits for a peer before it gives up (1 ms .. 10 min, default 5000 ms; measured with the 100 MHz
 * constant clock on the device).  A launch that gives up sets bit 0 of its local_state[0] AND poisons its output (NaN / INT_MIN for the
 * all-reduce, zero rows for the all-to-all): a late rank never yields a plausible wrong result.  Process-wide. */
int ao_collective_set_timeout_ms(int ms);
int ao_collective_timeout_ms(void);

/* Bytes of the flag block every rank allocates (zero-filled) and maps into its peers, and of the rank-local state block (zero-filled,
 * never shared: a status word + one epoch counter per block).  Host-only helpers. */
int64_t ao_allreduce_flag_bytes(void);
int64_t ao_allreduce_state_bytes(void);

/* SUM (or, _op with op = 1, elementwise MAX) all-reduce of one small vector in ONE launch per rank ("direct / one-shot
 * algorithm for S <= ~1 MiB, never a ring on the fully connected 8-GPU xGMI mesh"; the caller-issued all-reduce of the reference's TP
 * harness, torchao/testing/utils.py:370-467; MAX is the row-amax exchange of the exact row-parallel protocol).  Every rank stages its
 * vector in device memory all peers have mapped (IPC), raises a flag in every peer's flag block, waits (bounded) for all flags, then
 * reads all `world` staged vectors and combines them in rank order (fp32 accumulation for bf16 / fp32, exact for int32):
 * bit-identical results on every rank.
 *   peer_data_host / peer_flags_host: HOST arrays of `world` DEVICE pointers, index = rank (own buffers at [rank]); every staging
 *     buffer is 2 x slot_bytes (double-buffered by epoch parity), every flag block ao_allreduce_flag_bytes() bytes, zero-filled once
 *     (ao_peer_alloc kinds 1 and 0);
 *   input / output: count elements of dtype (0 fp32, 1 bf16, 2 int32), count x size a multiple of 16 and <= slot_bytes; may alias;
 *   local_state: device uint32 [ao_allreduce_state_bytes() / 4]: word 0 is set to 1 if a peer did not arrive within the timeout
 *     (the output is then NaN / INT_MIN) -- check it wherever the host synchronises anyway; the rest are the per-block epoch counters
 *     the kernel bumps on every call (they live on the device so that a launch captured into a hipGraph replays correctly).  Every
 *     rank must make the same sequence of calls. */
int ao_allreduce_oneshot(void* const* peer_data_host, void* const* peer_flags_host, const void* input,
                         void* output, void* local_state, int64_t count, int dtype, int64_t slot_bytes,
                         int rank, int world, void* stream);
int ao_allreduce_oneshot_op(void* const* peer_data_host, void* const* peer_flags_host, const void* input,
                            void* output, void* local_state, int64_t count, int dtype, int op,
                            int64_t slot_bytes, int rank, int world, void* stream);

/* On-device all-to-all-v of MXFP8 token rows (expert-parallel dispatch without a host round trip for the split sizes).  Replaces the
 * Triton kernel torchao/prototype/moe_training/kernels/mxfp8/comms.py:318-460 (_mxfp8_all_to_all_v_kernel, _exchange_row_offsets) that
 * MXFP8OnDeviceAllToAllV.forward (:52-167) launches over symmetric memory.  Every rank has staged its e4m3 rows (ordered by destination
 * rank), its E8M0 scale rows and its int64 split vector (rows it sends to rank r) in buffers all peers have mapped; this rank pulls the
 * rows addressed to it: from rank q, rows [sum_{r < rank} splits_q[r], + splits_q[rank]) land at local rows [sum_{p < q} splits_p[rank], ..),
 * and out_splits[q] = splits_q[rank].  Barriers before (inputs staged everywhere) and after (staging may be overwritten) are inside
 * the launch; epochs live on the device (hipGraph-replayable); a peer that never arrives sets bit 0 of local_state[0] after the
 * collective timeout (nothing is read from it: out_splits of that peer = 0), more rows than max_out_rows sets bit 1 (the rows past the
 * end are not written), a peer's split prefix that reaches past its max_in_rows staged rows sets bit 2 (the read is clamped to them).
 *   peer_*_host: HOST arrays of `world` DEVICE pointers, index = rank (own buffers at [rank]); data / scale staging 16-byte aligned,
 *     flag blocks ao_moe_a2a_flag_bytes() bytes, zero-filled once; local_state: ao_moe_a2a_state_bytes() bytes, zero-filled once;
 *   row_bytes = D (e4m3), a multiple of 16; scale_row_bytes = D / 32.  Every rank must make the same sequence of calls. */
int64_t ao_moe_a2a_flag_bytes(void);
int64_t ao_moe_a2a_state_bytes(void);
int ao_moe_a2a_v(void* const* peer_data_host, void* const* peer_scales_host, void* const* peer_splits_host,
                 void* const* peer_flags_host, void* out_data, void* out_scales, int64_t* out_splits,
                 void* local_state, int64_t row_bytes, int64_t scale_row_bytes, int64_t max_in_rows,
                 int64_t max_out_rows, int rank, int world, void* stream);

/* ------------------------------------------------------------------------- *
 * fp8 activations x int4 weights (Float8DynamicActivationInt4WeightConfig)
 * ------------------------------------------------------------------------- */

/* Replaces mslk.f8i4bf16_rowwise as called by Int4Tensor's F.linear with activation_dtype = float8_e4m3fn
 * (torchao/quantization/quantize_/workflows/int4/int4_tensor.py:213-229; quant_api.py:630-699):
 *   xq e4m3 [M][K], x_scale fp32 [M] (per-row dynamic cast: ao_fp8_quantize_rowwise);
 *   qdata / scale_and_zero: the int4 weight in the tinygemm tile order with offset-8 codes (what Int4Tensor.tile_packed() builds from
 *   the PLAIN data; scale_and_zero bf16 [K/group_size][N][2], zero = 0 for the reference's symmetric flavour); bias bf16 [N] or NULL.
 *   y[m][n] = bf16( x_scale[m] * sum_g ( s[g][n] * sum_{k in g} xq[m][k] (q[n][k] - 8) + z[g][n] * sum_{k in g} xq[m][k] ) + bias[n] )
 * fp8 MFMA on the codes themselves, the scale applied per group to fp32 sums (no per-weight rounding).  N % 16 == 0, K % 128 == 0. */
int ao_fp8_int4_linear(const uint8_t* xq, const float* x_scale, const int32_t* qdata,
                       const uint16_t* scale_and_zero, const uint16_t* bias, uint16_t* y, int64_t M,
                       int64_t N, int64_t K, int group_size, void* stream);
/* The same linear on the bf16 activation, its per-row e4m3 cast (ao_fp8_quantize_rowwise's arithmetic: float8_tensor.py:167-253 as
 * Int4Tensor's F.linear applies it, int4_tensor.py:205-212) fused into the launch -- SURVEY.md 8 f1 for this path.  Shapes:
 * ao_fp8_int4_dynamic_fits (M <= 16, M * (K + 16) <= 65536).  Same bits as cast + ao_fp8_int4_linear. */
int ao_fp8_int4_dynamic_fits(int64_t M, int64_t N, int64_t K);
int ao_fp8_int4_dynamic_linear(const uint16_t* x, const int32_t* qdata, const uint16_t* scale_and_zero,
                               const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K,
                               int group_size, void* stream);

/* ------------------------------------------------------------------------- *
 * Expert-parallel token regrouping (between the all-to-all and the grouped GEMM)
 * ------------------------------------------------------------------------- */

/* Replaces generate_permute_indices (torchao/prototype/moe_training/ep/kernels.py:132-214; Triton `_fill_indices_kernel` :13-60,
 * semantics of fill_indices_cpu :94-129).
 *   tokens_per_expert_group int32 [num_ranks * experts_per_rank], rank-major counts of the tokens this rank received;
 *   start_workspace         int32 [num_ranks * experts_per_rank] scratch (exclusive prefix sum of the counts);
 *   permuted_indices        int32 [max_len]: source row of every expert-major position, -1 for alignment padding and the tail;
 *   m_sizes / m_offsets     int32 [experts_per_rank]: align_up(max(tokens of expert, alignment), alignment) and its inclusive cumsum. */
int ao_moe_permute_indices(const int32_t* tokens_per_expert_group, int32_t* start_workspace,
                           int32_t* permuted_indices, int32_t* m_sizes, int32_t* m_offsets,
                           int64_t experts_per_rank, int64_t num_ranks, int64_t max_len,
                           int alignment, void* stream);

/* Replaces the row gather `vstack(x, 0)[permuted_indices]` of permute_and_pad / _PermuteMXFP8FwdHPBwd.forward (ep/permute.py:86-96,
 * 195-198) and of _UnpermuteHPFwdMXFP8Bwd.backward (ep/unpermute.py:57-98): out[i] = inputs[indices[i]] when 0 <= indices[i] <
 * num_rows_in, else zeros (index -1 and index num_rows_in are the reference's appended zero row).  Rows are row_bytes bytes of any
 * dtype (bf16 tokens, e4m3 data, e8m0 scales). */
int ao_moe_gather_rows(const void* inputs, const int32_t* indices, void* out, int64_t num_rows_in,
                       int64_t num_rows_out, int64_t row_bytes, void* stream);

/* Replaces the row scatter `out[permuted_indices] = y; out[:-1]` of _UnpermuteHPFwdMXFP8Bwd.forward / _unpermute_bf16
 * (ep/unpermute.py:36-41, 152-158): out[indices[i]] = inputs[i] when 0 <= indices[i] < num_rows_out (rows sent to the dummy row are
 * dropped; rows of `out` no index names are left untouched, as in the reference's new_empty). */
int ao_moe_scatter_rows(const void* inputs, const int32_t* indices, void* out, int64_t num_rows_in,
                        int64_t num_rows_out, int64_t row_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Dynamic-activation linears with the activation cast fused in (decode sizes)
 * ------------------------------------------------------------------------- */

/* 1 when the fused entry points below accept the shape: 0 < M <= 16, M * (K + 16) <= 65536 (the cast activation lives in
 * LDS), N % 16 == 0, K % 128 == 0.  Host-only helper. */
int ao_dyn_linear_fits(int64_t M, int64_t N, int64_t K);

/* Replaces the Int8Tensor F.linear with dynamic activation quantisation as ONE launch: Int8Tensor.from_hp(x, PerRow)
 * (int8_tensor.py:176-248) followed by _int_scaled_matmul + weight scale (int8/kernels.py:114-144, int8_tensor.py:305-359).
 * Same bits as ao_int8_quantize_rowwise + ao_int8_scaled_mm.
 *   x bf16 [M][K]; wq int8 [N][K]; w_scale fp32 [N]; bias bf16 [N] or NULL; y bf16 [M][N]. */
int ao_int8_dynamic_linear(const uint16_t* x, const int8_t* wq, const float* w_scale,
                           const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K,
                           void* stream);

/* Same for float8 rowwise: Float8Tensor.from_hp(x, PerRow) (float8_tensor.py:167-253) + aten::_scaled_mm with rowwise scales
 * (float8/inference.py:104-123).  Same bits as ao_fp8_quantize_rowwise + ao_fp8_scaled_mm at these sizes.
 *   x bf16 [M][K]; wq e4m3fn [N][K]; w_scale fp32 [N]; bias bf16 [N] or NULL; y bf16 [M][N]. */
int ao_fp8_dynamic_linear(const uint16_t* x, const uint8_t* wq, const float* w_scale,
                          const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K,
                          void* stream);

/* ---- K-sharded (row-parallel) 8-bit linears: the unsharded result from sharded operands ---------------------------------
 * torchao delegates TP to its caller (DTensor / vLLM slice the subclasses: int8_tensor.py:362-422, float8_tensor.py:732-839;
 * harness torchao/testing/utils.py:370-519).  A caller that wants the UNSHARDED linear's bits from K shards needs the
 * activation scale over the full K and the scale epilogue applied once, after the partial sums are added:
 *   amax_local = ao_rowwise_amax(x_shard);  all_reduce(MAX, amax);  q_shard = ao_*_quantize_rowwise_amax(x_shard, amax);
 *   acc = ao_int8_int_mm / ao_fp8_mm_f32 (q_shard, w_shard);  all_reduce(SUM, acc);  y = ao_*_scale_epilogue(acc).
 * `ldx` is the row stride of x in elements (a column slice of the full activation needs no copy). */
int ao_rowwise_amax(const uint16_t* x, int64_t ldx, float* amax, int64_t M, int64_t K, void* stream);
/* Int8Tensor.from_hp arithmetic (int8_tensor.py:191-230) with the row's amax given instead of reduced over x's K columns. */
int ao_int8_quantize_rowwise_amax(const uint16_t* x, int64_t ldx, const float* amax, int8_t* q,
                                  float* scale, int64_t M, int64_t K, void* stream);
/* Float8Tensor.from_hp arithmetic (float8_tensor.py:167-253) with the row's amax given. */
int ao_fp8_quantize_rowwise_amax(const uint16_t* x, int64_t ldx, const float* amax, uint8_t* q,
                                 float* scale, int64_t M, int64_t K, void* stream);
/* c fp32 [M][N] = a e4m3fn [M][K] @ b e4m3fn [N][K]^T, no scales: the accumulator of aten::_scaled_mm
 * (float8/inference.py:104-123) before its scale epilogue. */
int ao_fp8_mm_f32(const uint8_t* a, const uint8_t* b, float* c, int64_t M, int64_t N, int64_t K,
                  void* stream);
/* y = bf16( bf16(f32(acc) * x_scale[m]) * w_scale[n] (+ bias[n]) ): the epilogue of int8_tensor.py:315-359 on its own. */
int ao_int8_scale_epilogue(const int32_t* acc, const float* x_scale, const float* w_scale,
                           const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, void* stream);
/* y = bf16( acc * scale_a[m] * scale_b[n] (+ bias[n]) ): the epilogue of aten::_scaled_mm (rowwise) on its own. */
int ao_fp8_scale_epilogue(const float* acc, const float* scale_a, const float* scale_b,
                          const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AO_MI355_H */
