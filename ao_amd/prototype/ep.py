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
  * kernels/mxfp8/comms.py:25-400 -- the ON-DEVICE all-to-all-v (`OnDeviceAllToAllV`, `mxfp8_on_device_all_to_all_v`): split sizes stay
    on the device, every rank pulls its rows from peer-mapped staging buffers with one HIP kernel (csrc/a2a_kernels.hip) instead of
    the reference's Triton kernel over symmetric memory.  Opt-in (verified with two ranks on one GPU); RCCL's all_to_all_single is
    the default exchange.
Backward passes are training-only (outside SURVEY.md section 8).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .mx import BLOCK, ScaleCalculationMode

__all__ = ["MXFP8Tokens", "a2a_dispatch_mxfp8_fwd", "a2a_combine_hp_fwd", "exchange_split_sizes", "generate_permute_indices",
           "permute_and_pad", "permute_mxfp8_fwd", "unpermute_hp_fwd", "OnDeviceAllToAllV", "mxfp8_on_device_all_to_all_v"]


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


class OnDeviceAllToAllV:
    """All-to-all-v of MXFP8 token rows with the split sizes ON THE DEVICE (reference MXFP8OnDeviceAllToAllV / _mxfp8_on_device_all_to_all_v,
    kernels/mxfp8/comms.py:25-167, 271-316).  One instance per (group, D, max rows): it owns this rank's staging buffers -- e4m3 rows,
    E8M0 scale rows, the int64 split vector -- and a flag block, allocated through the C ABI as fine-grained / uncached device memory and
    exported to the peers as raw IPC handles once (ao_amd/peer_mem.py; HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack), like the reference's
    symmetric-memory buffers.
    `__call__` stages the inputs and launches `ao_moe_a2a_v`: no host synchronisation, capturable into a hipGraph.
    `ok` False / `why`: the set-up failed (callers fall back to `a2a_dispatch_mxfp8_fwd` over RCCL)."""

    def __init__(self, max_rows: int, dim: int, group=None, device=None):
        import ctypes

        from .. import _lib

        self.group = dist.group.WORLD if group is None else group
        self.max_rows, self.dim = int(max_rows), int(dim)
        self.ok, self.why = False, None
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.device = device
        try:
            assert dim % BLOCK == 0 and dim % 16 == 0, f"D = {dim} must be a multiple of {BLOCK}"
            lib = _lib.lib()
            world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
            if world > 8:
                raise RuntimeError("the on-device all-to-all handles at most 8 ranks (one xGMI-connected node)")
            from .. import peer_mem

            # staging (rows, scale rows, the split vector: read by the peers) fine-grained, the flag block uncached -- peer_mem.py
            own, ptrs, self._keep, self.memory = peer_mem.exchange(
                [(self.max_rows * dim, peer_mem.FINEGRAINED), (self.max_rows * (dim // BLOCK), peer_mem.FINEGRAINED),
                 (8 * world, peer_mem.FINEGRAINED), (lib.ao_moe_a2a_flag_bytes(), peer_mem.UNCACHED)], self.group, device)
            peer_mem.require_coherent(self.memory, self.group, device)  # (raises -> ok stays False: the RCCL all-to-all serves the calls)
            self._data = own[0].view(self.max_rows, dim)
            self._scales = own[1].view(self.max_rows, dim // BLOCK)
            self._splits = own[2].view(torch.int64)
            self._flags = own[3]
            self._state = torch.zeros(lib.ao_moe_a2a_state_bytes() // 4, dtype=torch.int32, device=device)
            torch.cuda.synchronize(device)
            self._arrs = [(ctypes.c_void_p * world)(*p) for p in ptrs]
            self._lib, self._check = lib, _lib.check
            self.rank, self.world = rank, world
            dist.barrier(group=self.group)  # everybody has mapped everybody before the first flag is raised
            self.ok = True
        except Exception as e:  # noqa: BLE001 -- any failure means "use the RCCL exchange"
            self.why = f"{type(e).__name__}: {e}"

    def status(self) -> int:
        """bit 0: a peer did not arrive within the collective timeout (its rows were not read); bit 1: more rows arrived than max_rows;
        bit 2: a peer's split vector reached past its staged rows (host-synchronising read)."""
        return int(self._state[0].item())

    def check(self):
        """Raise on any status bit -- call it where the host synchronises anyway."""
        st = self.status()
        if st:
            why = [m for b, m in ((1, "a peer did not arrive within the collective timeout"), (2, f"more than max_rows = {self.max_rows} rows arrived"),
                                  (4, "a peer's input_splits reach past its staged rows")) if st & b]
            raise RuntimeError("on-device all-to-all-v: " + "; ".join(why))

    def __call__(self, data: torch.Tensor, scales: torch.Tensor, input_splits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """data e4m3 / uint8 [T, D] ordered by destination rank, scales e8m0 / uint8 [T, D / 32], input_splits int64 [world] ON THE DEVICE
        -> (out_data uint8 [max_rows, D], out_scales uint8 [max_rows, D / 32], output_splits int64 [world]); rows past
        output_splits.sum() are undefined, like the reference's."""
        assert self.ok, f"on-device all-to-all is not set up: {self.why}"
        t = data.shape[0]
        assert data.dim() == 2 and data.shape[1] == self.dim and t <= self.max_rows and scales.shape == (t, self.dim // BLOCK)
        assert input_splits.dtype == torch.int64 and input_splits.numel() == self.world and input_splits.is_cuda
        self._data[:t].copy_(data.view(torch.uint8))
        self._scales[:t].copy_(scales.view(torch.uint8))
        self._splits.copy_(input_splits)
        out = torch.empty((self.max_rows, self.dim), dtype=torch.uint8, device=self.device)
        out_s = torch.empty((self.max_rows, self.dim // BLOCK), dtype=torch.uint8, device=self.device)
        out_splits = torch.empty(self.world, dtype=torch.int64, device=self.device)
        self._check(self._lib.ao_moe_a2a_v(*self._arrs, out.data_ptr(), out_s.data_ptr(), out_splits.data_ptr(), self._state.data_ptr(),
                                          self.dim, self.dim // BLOCK, self.max_rows, self.max_rows, self.rank, self.world,
                                          torch.cuda.current_stream(self.device).cuda_stream))
        return out, out_s, out_splits


_ON_DEVICE_A2A = {}  # (group id, D, max rows) -> OnDeviceAllToAllV: the reference keeps its symmetric buffers as class attributes


def mxfp8_on_device_all_to_all_v(input: torch.Tensor, input_splits: torch.Tensor, max_output_rows_per_rank: int, group=None,
                                 cast: Optional[Callable] = None):
    """reference MXFP8OnDeviceAllToAllV.forward (comms.py:52-167): cast the tokens to MXFP8 (to_mx defaults: FLOOR, 1 x 32), exchange
    data and scales on the device as `input_splits` (an int64 DEVICE tensor, rows for every rank) says, dequantize what arrived.
    Returns (tokens in input.dtype [sum(output_splits), D], output_splits int64 [world] on the device).  The only host sync is the
    final slice by output_splits.sum(), as in the reference (:164-166)."""
    assert input.dtype in (torch.float32, torch.bfloat16) and input.dim() == 2
    group = dist.group.WORLD if group is None else group
    key = (id(group), input.shape[1], int(max_output_rows_per_rank))
    ex = _ON_DEVICE_A2A.get(key)
    if ex is None:
        ex = _ON_DEVICE_A2A[key] = OnDeviceAllToAllV(max_output_rows_per_rank, input.shape[1], group, input.device)
    if not ex.ok:
        raise RuntimeError(f"on-device all-to-all unavailable ({ex.why}); use a2a_dispatch_mxfp8_fwd")
    data, scale = (cast or _gpu_cast)(input.to(torch.bfloat16).contiguous(), ScaleCalculationMode.FLOOR)
    out, out_s, output_splits = ex(data, scale, input_splits)
    from .mx import mx_dequantize

    hp = mx_dequantize(out_s.view(torch.float8_e8m0fnu), out.view(torch.float8_e4m3fn), input.dtype)
    rows = int(output_splits.sum().item())  # the reference's one host sync (:164-166); the status word is read behind it
    ex.check()
    return hp[:rows], output_splits
