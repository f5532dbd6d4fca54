"""Expert-parallel token exchange in front of / behind the MXFP8 grouped GEMM, MI355X-native (forward only).

Host-side mirror of torchao/prototype/moe_training/ep/:
  * a2a_dispatch.py:18-112  _A2ADispatchMXFP8FwdHPBwd.forward -- cast the bf16 tokens to MXFP8 FIRST (1 x 32 blocks, RCEIL), then
    exchange the e4m3 bytes and the E8M0 scale bytes with two all_to_all_single calls: 33 bytes per 32 elements cross xGMI
    instead of 64;
  * a2a_combine.py:18-95    _A2ACombineHPFwdMXFP8Bwd.forward   -- the way back is a plain bf16 all-to-all.
One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI; the byte tensors are exchanged as uint8 because
neither RCCL nor gloo knows the fp8 dtypes -- the reference does the same at :78-83).  xGMI is point-to-point, so an
all-to-all of S bytes per peer moves over all 7 links at once: the exchange is sized by the busiest rank's splits.
The result feeds `_to_mxfp8_then_scaled_grouped_mm` (ao_amd/prototype/mx.py) directly: pre-quantized tokens skip its cast,
like the reference's MXTensor input (mxfp8_grouped_mm.py:173-178, 482-486).

  * permute.py / unpermute.py / kernels.py -- regrouping the received tokens from rank-major to expert-major order with aligned
    groups (`generate_permute_indices`, `permute_and_pad`, `permute_mxfp8_fwd`) and back (`unpermute_hp_fwd`): HIP index + row
    kernels (csrc/moe_permute_kernels.hip) instead of Triton.
Backward passes are training-only (outside SURVEY.md section 8).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .mx import BLOCK, ScaleCalculationMode

__all__ = ["MXFP8Tokens", "a2a_dispatch_mxfp8_fwd", "a2a_combine_hp_fwd", "exchange_split_sizes", "generate_permute_indices",
           "permute_and_pad", "permute_mxfp8_fwd", "unpermute_hp_fwd"]


def _round_up(x: int, y: int) -> int:
    return (x + y - 1) // y * y


def generate_permute_indices(tokens_per_expert_group: torch.Tensor, experts_per_rank: int, num_ranks: int, max_len: int, alignment: int):
    """reference ep/kernels.py:132-214 -> (permuted_indices int32 [max_len], m_sizes int32 [E], m_offsets int32 [E])."""
    from .. import ops
    return ops.generate_permute_indices(tokens_per_expert_group, experts_per_rank, num_ranks, max_len, alignment)


def permute_and_pad(x: torch.Tensor, num_tokens_per_expert: torch.Tensor, ep_degree: int, num_local_experts: int, alignment: int):
    """reference permute_and_pad (ep/permute.py:170-204): bf16 tokens from rank-major to expert-major order, every expert's group padded
    to `alignment` rows (padding rows are zero).  Returns (input_shape incl. the reference's dummy row, permuted x, permuted_indices,
    num_tokens_per_expert_padded, group_offsets)."""
    from .. import ops
    padded_max_len = _round_up(x.shape[0] + num_local_experts * alignment, alignment)
    idx, m_sizes, m_offsets = generate_permute_indices(num_tokens_per_expert, num_local_experts, ep_degree, padded_max_len, alignment)
    input_shape = torch.Size((x.shape[0] + 1, x.shape[1]))
    return input_shape, ops.gather_rows(x, idx), idx, m_sizes, m_offsets


def permute_mxfp8_fwd(tokens: "MXFP8Tokens", num_tokens_per_expert: torch.Tensor, ep_degree: int, num_local_experts: int,
                      group_size_multiple_of: int = 32):
    """reference _PermuteMXFP8FwdHPBwd.forward (ep/permute.py:60-125): the same regrouping applied to the e4m3 bytes and the E8M0 scale
    bytes separately.  Returns (padded_shape, MXFP8Tokens, permuted_indices, num_tokens_per_expert_padded, group_offsets)."""
    from .. import ops
    data, scale = tokens.data, tokens.scale
    padded_max_len = _round_up(data.shape[0] + num_local_experts * group_size_multiple_of, group_size_multiple_of)
    idx, m_sizes, m_offsets = generate_permute_indices(num_tokens_per_expert, num_local_experts, ep_degree, padded_max_len, group_size_multiple_of)
    d = ops.gather_rows(data.view(torch.uint8), idx).view(data.dtype)
    s = ops.gather_rows(scale.view(torch.uint8), idx).view(scale.dtype)
    padded_shape = torch.Size((data.shape[0] + 1, data.shape[1]))
    return padded_shape, MXFP8Tokens(d, s, tokens.orig_dtype), idx, m_sizes, m_offsets


def unpermute_hp_fwd(input: torch.Tensor, permuted_indices: torch.Tensor, padded_shape) -> torch.Tensor:
    """reference _UnpermuteHPFwdMXFP8Bwd.forward / _unpermute_bf16 (ep/unpermute.py:23-47, 140-158): scatter the expert-major rows back
    to their rank-major positions; `padded_shape` is the shape WITH the dummy row, the result has padded_shape[0] - 1 rows."""
    from .. import ops
    return ops.scatter_rows(input, permuted_indices, int(padded_shape[0]) - 1)


class MXFP8Tokens:
    """Token rows already cast to MXFP8: data e4m3 [T, D], scale e8m0 [T, D / 32] (plain row-major: what the CDNA4 scaled MFMA
    takes).  The stand-in for the reference's MXTensor on this path (a2a_dispatch.py:93-104)."""

    def __init__(self, data: torch.Tensor, scale: torch.Tensor, orig_dtype: torch.dtype = torch.bfloat16):
        assert data.dim() == 2 and scale.dim() == 2 and data.shape[0] == scale.shape[0] and data.shape[1] == scale.shape[1] * BLOCK
        self.data, self.scale, self.orig_dtype = data, scale, orig_dtype

    @property
    def shape(self):
        return self.data.shape

    def dequantize(self, output_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        from .mx import mx_dequantize
        return mx_dequantize(self.scale, self.data, output_dtype or self.orig_dtype)


def _gpu_cast(x: torch.Tensor, mode: ScaleCalculationMode) -> Tuple[torch.Tensor, torch.Tensor]:
    from .. import ops
    return ops.mxfp8_quantize(x, mode)


def _a2a_bytes(x_u8: torch.Tensor, out_rows: int, output_splits, input_splits, group) -> torch.Tensor:
    out = torch.empty((out_rows, x_u8.shape[1]), dtype=torch.uint8, device=x_u8.device)
    dist.all_to_all_single(out, x_u8.contiguous(), list(output_splits), list(input_splits), group=group)
    return out


def a2a_dispatch_mxfp8_fwd(input: torch.Tensor, output_splits: Sequence[int], input_splits: Sequence[int], group=None,
                           scaling_mode: ScaleCalculationMode = ScaleCalculationMode.RCEIL, block_size: int = BLOCK,
                           cast: Optional[Callable] = None) -> MXFP8Tokens:
    """reference a2a_dispatch_mxfp8_fwd_hp_bwd, forward (a2a_dispatch.py:40-112).

    input bf16 [T_local, D] (rows ordered by destination rank, `input_splits[r]` rows for rank r) ->
    MXFP8Tokens of sum(output_splits) rows (ordered by source rank).  `cast` (tests on CPU): the MXFP8 cast to use instead of
    the HIP kernel."""
    assert input.dtype == torch.bfloat16, f"Expected bf16 on MI355X, got {input.dtype}"
    assert block_size == BLOCK, "Only block_size=32 is supported"
    assert input.dim() == 2 and input.shape[1] % BLOCK == 0, f"tokens must be [T, D] with D a multiple of {BLOCK}"
    assert sum(input_splits) == input.shape[0], f"input_splits {list(input_splits)} do not cover {input.shape[0]} rows"
    group = dist.group.WORLD if group is None else group
    data, scale = (cast or _gpu_cast)(input.contiguous(), scaling_mode)
    rows = int(sum(output_splits))
    out_data = _a2a_bytes(data.view(torch.uint8), rows, output_splits, input_splits, group)
    out_scale = _a2a_bytes(scale.view(torch.uint8), rows, output_splits, input_splits, group)
    return MXFP8Tokens(out_data.view(torch.float8_e4m3fn), out_scale.view(torch.float8_e8m0fnu), input.dtype)


def a2a_combine_hp_fwd(input: torch.Tensor, output_splits: Sequence[int], input_splits: Sequence[int], group=None) -> torch.Tensor:
    """reference a2a_combine_hp_fwd_mxfp8_bwd, forward (a2a_combine.py:36-95): the expert outputs travel back in bf16."""
    assert input.dim() == 2 and sum(input_splits) == input.shape[0]
    group = dist.group.WORLD if group is None else group
    out = torch.empty((int(sum(output_splits)), input.shape[1]), dtype=input.dtype, device=input.device)
    dist.all_to_all_single(out, input.contiguous(), list(output_splits), list(input_splits), group=group)
    return out


def exchange_split_sizes(num_tokens_per_expert: torch.Tensor, group=None) -> Tuple[List[int], List[int], torch.Tensor]:
    """What the reference's callers compute in front of the dispatch (test_a2a_dispatch.py:69-92, torchtitan's token dispatcher):
    `num_tokens_per_expert` int [E_global] on this rank -> (input_splits, output_splits, num_tokens_per_expert_group), the
    last one int [world * E_local]: how many tokens each source rank sends for each of this rank's local experts.  One
    device-to-host sync (the split sizes are host integers for the collective), like the reference."""
    group = dist.group.WORLD if group is None else group
    world = dist.get_world_size(group)
    assert num_tokens_per_expert.numel() % world == 0, "experts must divide evenly over the EP ranks"
    counts = num_tokens_per_expert.to(torch.int64).contiguous()
    recv = torch.empty_like(counts)
    dist.all_to_all_single(recv, counts, group=group)
    input_splits = counts.view(world, -1).sum(dim=1).tolist()
    output_splits = recv.view(world, -1).sum(dim=1).tolist()
    return input_splits, output_splits, recv
