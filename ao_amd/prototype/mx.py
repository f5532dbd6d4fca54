"""MXFP8 (block 32, e4m3 elements, E8M0 scales): cast and MoE grouped GEMM forward, MI355X-native.

Host-side mirror of
  * torchao/prototype/mx_formats/mx_tensor.py:228-409  to_mx(x, float8_e4m3fn, 32, mode)
  * torchao/prototype/moe_training/mxfp8_grouped_mm.py:56-239,330-371
    _to_mxfp8_then_scaled_grouped_mm -- FORWARD only, numerics of the reference's
    emulated path (:959-1023): both operands dequantised per 32-block, fp32 accumulate, bf16 out.
CDNA4's scaled MFMA takes the E8M0 bytes as register operands, so scales stay in plain [rows, K/32]
layout: no 128x4 "blocked" swizzle and no per-group row padding are needed on this path
(torchao::mx_block_rearrange_2d_M_groups / fused_pad_token_groups have no work to do on this path; both exist as ops for callers that hold
the reference's padded / blocked formats: ops.mx_block_rearrange_2d_M_groups, ops.fused_pad_token_groups).
"""
from enum import Enum
from typing import Optional

import torch

from .. import ops

__all__ = ["ScaleCalculationMode", "to_mx", "mx_dequantize", "_to_mxfp8_then_scaled_grouped_mm", "pad_token_groups", "unpad_token_groups",
           "to_blocked", "mx_block_rearrange_2d_M_groups_cuda", "compute_blocked_scale_offsets_for_M_groups"]

BLOCK = 32


class ScaleCalculationMode(Enum):
    """reference mx_formats/config.py: FLOOR (QuantizeTensorToMXKwargs default) and RCEIL (inference
    and MoE default, inference_workflow.py:112, mxfp8_grouped_mm.py:64)"""

    FLOOR = "floor"
    RCEIL = "rceil"


def to_mx(data_hp: torch.Tensor, elem_dtype: torch.dtype = torch.float8_e4m3fn, block_size: int = BLOCK,
          scaling_mode: ScaleCalculationMode = ScaleCalculationMode.FLOOR):
    """(scale_e8m0 [..., C/32], data_e4m3 [..., C]) -- the reference's return order (mx_tensor.py:409)."""
    if elem_dtype != torch.float8_e4m3fn:
        raise NotImplementedError(f"to_mx on MI355X implements float8_e4m3fn elements only, got {elem_dtype}")
    if block_size != BLOCK:
        raise NotImplementedError(f"to_mx on MI355X implements block_size 32 only, got {block_size}")
    assert data_hp.dtype in (torch.bfloat16,), f"{data_hp.dtype} is not supported yet (bfloat16 only on MI355X)"
    assert data_hp.shape[-1] % block_size == 0, (
        f"the last dimension of shape {data_hp.shape} must be divisible by block_size {block_size}"
    )
    assert data_hp.is_contiguous(), "unsupported"
    q, s = ops.mxfp8_quantize(data_hp, scaling_mode)
    return s, q


def mx_dequantize(scale_e8m0: torch.Tensor, data_lp: torch.Tensor, output_dtype=torch.bfloat16) -> torch.Tensor:
    """reference mx_tensor.py:412-471 (torch ops on the GPU; a checker, not a hot path)"""
    s = torch.exp2(scale_e8m0.view(torch.uint8).to(torch.float32) - 127.0)
    x = data_lp.to(torch.float32).reshape(*data_lp.shape[:-1], -1, BLOCK) * s.unsqueeze(-1)
    return x.reshape(data_lp.shape).to(output_dtype)


class MXFP8ExpertWeights:
    """Expert weights cast to MXFP8 once (1 x 32 blocks along K of the [E, N, K] tensor): what the grouped GEMM streams.
    The reference casts B_t inside every forward (mxfp8_grouped_mm.py:330-371) because it also trains them; for inference
    the cast of Mixtral's w1 alone re-reads 940 MB of bf16 per call, so it is hoisted here."""

    def __init__(self, data: torch.Tensor, scale: torch.Tensor):
        self.data, self.scale = data, scale  # e4m3 [E, N, K], e8m0 [E, N, K/32]

    @classmethod
    def from_hp(cls, B_t: torch.Tensor, scale_calculation_mode: "ScaleCalculationMode" = None):
        assert B_t.ndim == 3 and B_t.dtype == torch.bfloat16, "B_t must be a 3-D bfloat16 tensor [E, K, N]"
        mode = ScaleCalculationMode.RCEIL if scale_calculation_mode is None else scale_calculation_mode
        return cls(*ops.mxfp8_quantize(B_t.transpose(-2, -1).contiguous(), mode))

    @property
    def shape(self):  # the [E, K, N] shape of the B_t it stands for
        e, n, k = self.data.shape
        return torch.Size((e, k, n))


_WEIGHT_MEMO = {}  # (data_ptr, version, shape, stride, mode) -> (weakref to the weight tensor, MXFP8ExpertWeights), bounded
_WEIGHT_MEMO_MAX = 256


def _cached_expert_weights(B_t, mode):
    """Memo of the one-time MXFP8 cast of an expert weight.  The key alone (address, version counter, shape) is not an identity: a freed
    weight's address can be handed to a new same-shape tensor (hot-swapped checkpoint, LoRA merge) with an equal version counter, so
    every entry also holds a WEAK reference to the tensor it was cast from (its base, for the usual `w.transpose(-2, -1)` view that is
    made anew on every call) -- a hit counts only when that very tensor is alive and is the one asked about; entries of dead tensors
    are dropped (and no longer pin their casts on the GPU)."""
    import weakref

    anchor = B_t._base if B_t._base is not None else B_t
    key = (B_t.data_ptr(), B_t._version, tuple(B_t.shape), tuple(B_t.stride()), str(mode))
    hit = _WEIGHT_MEMO.get(key)
    if hit is not None:
        ref, cast = hit
        if ref() is anchor:
            return cast
        del _WEIGHT_MEMO[key]  # same address, another (or a dead) tensor: stale
    for k_ in [k_ for k_, (r, _c) in _WEIGHT_MEMO.items() if r() is None]:
        del _WEIGHT_MEMO[k_]
    if len(_WEIGHT_MEMO) >= _WEIGHT_MEMO_MAX:
        _WEIGHT_MEMO.pop(next(iter(_WEIGHT_MEMO)))
    cast = MXFP8ExpertWeights.from_hp(B_t, mode)
    _WEIGHT_MEMO[key] = (weakref.ref(anchor), cast)
    return cast


FUSE_ACTIVATION_CAST = True  # (A/B and tests: False = the two-launch path)


def _to_mxfp8_then_scaled_grouped_mm(
    A: torch.Tensor,
    B_t,
    offs: Optional[torch.Tensor] = None,
    block_size: int = BLOCK,
    out_dtype: Optional[torch.dtype] = torch.bfloat16,
    scale_calculation_mode: ScaleCalculationMode = ScaleCalculationMode.RCEIL,
    cache_weights: bool = False,
) -> torch.Tensor:
    """Forward of the reference's MXFP8 MoE grouped GEMM.

    A     bf16 [M_total, K]   tokens, grouped by expert
    B_t   bf16 [E, K, N]      expert weights, "transposed" view of [E, N, K] (strides (N*K, 1, N))
    offs  int32 [E]           cumulative group ends along M
    ->    bf16 [M_total, N]
    B_t may also be an MXFP8ExpertWeights (cast once, MXFP8ExpertWeights.from_hp(B_t)); `cache_weights=True` memoises that
    cast per weight tensor (keyed on its storage and version counter: an in-place update re-casts) -- inference only.
    Raises like the reference for unsupported arguments (:167-200)."""
    from .ep import MXFP8Tokens
    if isinstance(A, MXFP8Tokens):
        # pre-quantized tokens (the output of the EP dispatch; reference: MXTensor input, mxfp8_grouped_mm.py:173-178, 482-486)
        assert block_size == BLOCK and offs is not None and A.shape[-1] == B_t.shape[-2], f"shape {A.shape} and {B_t.shape} are not compatible"
        w = B_t if isinstance(B_t, MXFP8ExpertWeights) else (
            _cached_expert_weights(B_t, scale_calculation_mode) if cache_weights else MXFP8ExpertWeights.from_hp(B_t, scale_calculation_mode))
        return ops.mxfp8_grouped_mm(A.data, A.scale, w.data, w.scale, offs.to(torch.int32))
    assert A.ndim == 2, "A must be 2D"
    if isinstance(B_t, MXFP8ExpertWeights):
        assert block_size == BLOCK, "Only block_size=32 is supported"
        assert offs is not None, "offs must be provided for 2d-2d and 2d-3d grouped mm"
        assert A.dtype == torch.bfloat16 and A.shape[-1] == B_t.shape[-2], f"shape {A.shape} and {B_t.shape} are not compatible"
        return _cast_then_grouped_mm(A, B_t, offs, scale_calculation_mode)
    assert B_t.ndim == 3, "B must be 3D"
    assert block_size == BLOCK, "Only block_size=32 is supported"
    assert offs is not None, "offs must be provided for 2d-2d and 2d-3d grouped mm"
    assert out_dtype == torch.bfloat16, "Only bfloat16 out_dtype is supported"
    assert A.dtype == torch.bfloat16 and B_t.dtype == torch.bfloat16, "A and B_t must be bfloat16"
    assert A.shape[-1] == B_t.shape[-2], f"shape {A.shape} and {B_t.shape} are not compatible for _scaled_grouped_mm"
    # weights: 1x32 blocks along K of the [E, N, K] tensor (the reference quantises B_t.transpose(-2, -1))
    w = _cached_expert_weights(B_t, scale_calculation_mode) if cache_weights else MXFP8ExpertWeights.from_hp(B_t, scale_calculation_mode)
    return _cast_then_grouped_mm(A, w, offs, scale_calculation_mode)


def _to_mxfp8_then_scaled_grouped_mm_pair(A: torch.Tensor, B1_t, B3_t, offs: torch.Tensor, scale_calculation_mode: ScaleCalculationMode = ScaleCalculationMode.RCEIL,
                                          cache_weights: bool = False):
    """(A @ w1, A @ w3) of an MoE layer's experts -- the reference computes them by two calls of _to_mxfp8_then_scaled_grouped_mm
    (mxfp8_grouped_mm.py:56-239), casting A twice; here decode-size groups take ONE launch for both products with the cast fused in
    (ops.mxfp8_grouped_mm_pair), other shapes two calls.  Same values either way (the same bits unless the pair launch's stream-K shares cut a tile at other k steps than
    the single launches: then single elements may round the other way -- include/ao_mi355.h).  B1_t / B3_t: bf16 [E, K, N] views or MXFP8ExpertWeights."""
    assert A.ndim == 2 and A.dtype == torch.bfloat16 and offs is not None
    ws = []
    for B_t in (B1_t, B3_t):
        if isinstance(B_t, MXFP8ExpertWeights):
            ws.append(B_t)
        else:
            assert B_t.ndim == 3 and B_t.dtype == torch.bfloat16, "B must be a 3-D bfloat16 tensor or MXFP8ExpertWeights"
            ws.append(_cached_expert_weights(B_t, scale_calculation_mode) if cache_weights else MXFP8ExpertWeights.from_hp(B_t, scale_calculation_mode))
    w1, w3 = ws
    assert w1.data.shape == w3.data.shape and A.shape[-1] == w1.data.shape[-1], f"shapes {A.shape}, {w1.data.shape}, {w3.data.shape} are not compatible"
    E, N, K = w1.data.shape
    offs = offs.to(torch.int32)
    if FUSE_ACTIVATION_CAST and A.is_cuda and ops.mxfp8_grouped_mm_pair_fits(A.shape[0], N, K, E):
        return ops.mxfp8_grouped_mm_pair(A, w1.data, w1.scale, w3.data, w3.scale, offs, scale_calculation_mode)
    return _cast_then_grouped_mm(A, w1, offs, scale_calculation_mode), _cast_then_grouped_mm(A, w3, offs, scale_calculation_mode)


def _cast_then_grouped_mm(A, w, offs, scale_calculation_mode):
    """to_mx(A) then the grouped mm (mxfp8_grouped_mm.py:330-371).  Decode-size groups take ONE launch with the cast fused into the kernel's
    A-fill (round 6, SURVEY 8 f1: ops.mxfp8_grouped_mm_dyn, bit-identical); other shapes the cast kernel and the GEMM."""
    E, N, K = w.data.shape
    offs = offs.to(torch.int32)
    if FUSE_ACTIVATION_CAST and A.is_cuda and ops.mxfp8_grouped_mm_dyn_fits(A.shape[0], N, K, E):
        return ops.mxfp8_grouped_mm_dyn(A, w.data, w.scale, offs, scale_calculation_mode)
    a_q, a_s = ops.mxfp8_quantize(A.contiguous(), scale_calculation_mode)
    return ops.mxfp8_grouped_mm(a_q, a_s, w.data, w.scale, offs)


def pad_token_groups(input_act: torch.Tensor, group_end_offsets: torch.Tensor, alignment_size: int = 32):
    """Mirror of torchao.prototype.moe_training.utils.pad_token_groups (utils.py:412-445): pad every token group to
    the next multiple of `alignment_size` (32 for MXFP8, 16 for FP8) so that grouped-GEMM tiles never straddle two
    experts.  Returns (padded_input_act, padded_group_start_offsets, padded_group_end_offsets)."""
    return ops.fused_pad_token_groups(input_act, group_end_offsets, alignment_size)


def unpad_token_groups(padded_output: torch.Tensor, original_group_end_offsets: torch.Tensor,
                       padded_group_start_offsets: torch.Tensor, num_tokens: int, alignment_size: int = 32) -> torch.Tensor:
    """Mirror of torchao.prototype.moe_training.utils.unpad_token_groups (utils.py:448-490)."""
    return ops.fused_unpad_token_groups(padded_output, original_group_end_offsets, padded_group_start_offsets, num_tokens, alignment_size)


def to_blocked(input_matrix: torch.Tensor, use_triton_kernel: bool = False) -> torch.Tensor:
    """Mirror of torchao.prototype.mx_formats.utils.to_blocked (utils.py:31-72): E8M0 scales [H, W] -> the flat 128 x 4 blocked layout,
    32 ceil(H / 128) x 16 ceil(W / 4) bytes.  `use_triton_kernel` picks between two implementations of the same bytes upstream; one kernel here."""
    return ops.mx_to_blocked(input_matrix)


def mx_block_rearrange_2d_M_groups_cuda(scales_tensor: torch.Tensor, input_offsets: torch.Tensor, chunks_per_tb: int = 4) -> torch.Tensor:
    """Mirror of torchao.prototype.moe_training.kernels.mxfp8.quant.mx_block_rearrange_2d_M_groups_cuda (quant.py:1183-1225): per token group,
    the blocked layout of its scales at the row where the previous groups' 128-row-padded blocks end."""
    return ops.mx_block_rearrange_2d_M_groups(scales_tensor, input_offsets, chunks_per_tb)


def compute_blocked_scale_offsets_for_M_groups(offsets: torch.Tensor):
    """Mirror of quant.py:309-335: (group sizes, starting row of every group's scales after padding each group to 128 rows, leading 0) --
    where mx_block_rearrange_2d_M_groups_cuda put each group.  Index arithmetic on [num_groups] integers (device-side torch ops, no sync)."""
    zero = torch.zeros(1, dtype=offsets.dtype, device=offsets.device)
    group_sizes = torch.diff(offsets, prepend=zero)
    starts = torch.cumsum((group_sizes + 127) // 128 * 128, dim=0)
    return group_sizes, torch.cat([zero, starts.to(offsets.dtype)])
