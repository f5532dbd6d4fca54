"""Tensor-parallel linears over RCCL for the TP-linear benchmark (BASELINE config 4).

torchao itself has no TP code: its subclasses implement aten.slice so that a caller (vLLM, DTensor)
can shard them (int4_tile_packed_to_4d_tensor.py:302-385, float8_tensor.py:732-839,
int8_tensor.py:362-422; harness torchao/testing/utils.py:370-519).  This module is that caller,
MI355X-first: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI),
Megatron pairing -- {q,k,v,gate,up} column-parallel (output features N split, no exchange),
{o,down} row-parallel (input features K split, ONE all-reduce(sum) of the bf16 [M, hidden] partial
per linear, issued on the compute stream right behind the GEMM).

Sharding units: N in multiples of 16 (one int4 n-tile; also the fp8/int8 kernels' store width),
K in multiples of lcm(128, group_size) for int4 (one packed k-block) and of 128 for the 8-bit
GEMMs.

Row-parallel 8-bit linears reproduce the UNSHARDED oracle: the per-row weight scale
is the full-K one (a slice of a tensor quantized before sharding keeps it), the per-row ACTIVATION
scale is made the full-K one by an all_reduce(MAX) of the rows' amax ([M] fp32) before the cast, the
raw accumulators (int32 / fp32 [M, N]) are all-reduced, and the scale epilogue runs once on the sum --
int8 bit-exact, fp8 within fp32 summation order of the unsharded linear.  From M = 128 rows on the all-reduce
is split around the epilogue: reduce-scatter (fp32 / int32) -> epilogue on M / world rows -> all-gather (bf16),
25 % fewer bytes over xGMI and a world-times smaller epilogue, the same bits for int8.  `reduce="bf16"` is the
cheaper protocol a caller of F.linear(x_shard, w_shard) + all_reduce gets from the reference
subclasses (locally scaled activation shards, bf16 partials; half the exchange bytes, ~1e-2 rel).
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["shard_bounds", "shard_unit", "ColumnParallelLinear", "RowParallelLinear", "shard_linear_", "tp_mlp", "OneShotAllReduce"]


class OneShotAllReduce:
    """SUM all-reduce of SMALL tensors (decode: a bf16 [1, 8192] partial is 16 KiB, the exact protocol's fp32 accumulator 32 KiB) in one
    hop: every rank stages its vector in a buffer all peers have mapped, raises a flag in every peer, waits for all flags, reads all
    staged vectors over xGMI and adds them in rank order ("direct / one-shot algorithm for S <= ~1 MiB, never a ring
    on the fully connected 8-GPU xGMI mesh").

    backend "hip" (default): the hand-written kernel `ao_allreduce_oneshot_op` (csrc/allreduce_kernels.hip), SUM or MAX, over buffers
    allocated through the C ABI as fine-grained (staging) / uncached (flags) device memory and exchanged as raw IPC handles
    (ao_amd/peer_mem.py; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack; `memory` names what was really allocated) -- fp32 / bf16 /
    int32, bit-identical on every rank, capturable into a hipGraph (epochs live on the device).  A rank that waits longer than
    `ao_collective_set_timeout_ms` for a peer poisons its output with NaN / INT_MIN and `check()` raises.  backend "symm_mem": `torch.distributed._symmetric_memory`
    + `symm_mem::one_shot_all_reduce` (round 2's prototype).  RCCL's all_reduce stays the path for anything larger, for MAX
    reductions, and whenever the set-up fails (`ok` False, `why` says what happened).  Verified on one GPU with two processes
    (tests/test_oneshot_allreduce_gpu.py); no multi-GPU node was available to this build.
    """

    _DT = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2}

    def __init__(self, group=None, max_bytes: int = 1 << 20, device=None, backend: str = "hip", force_coarse_grained: bool = False):
        self.group = dist.group.WORLD if group is None else group
        self._force_coarse = force_coarse_grained  # tests: the round-3 allocation path (torch allocator + storage sharing)
        self.memory = None
        self.max_bytes = (max_bytes + 15) // 16 * 16
        self.ok = False
        self.why = None
        self.backend = backend
        self.calls = 0
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.device = device
        try:
            if backend == "hip":
                self._setup_hip(device)
            else:
                import torch.distributed._symmetric_memory as symm_mem

                self.group_name = self.group.group_name
                if hasattr(symm_mem, "enable_symm_mem_for_group"):
                    symm_mem.enable_symm_mem_for_group(self.group_name)
                self.buf = symm_mem.empty(self.max_bytes, dtype=torch.uint8, device=device)
                self.handle = symm_mem.rendezvous(self.buf, self.group_name)
            self.ok = True
        except Exception as e:  # noqa: BLE001 -- any failure means "use RCCL"
            self.why = f"{type(e).__name__}: {e}"

    def _setup_hip(self, device):
        import ctypes

        from . import _lib, peer_mem

        lib = _lib.lib()
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        if world > 8:
            raise RuntimeError("the one-shot kernel handles at most 8 ranks (one xGMI-connected node)")
        # own buffers: staging fine-grained, flags uncached (peer_mem.py says why), zero-filled, exported as raw IPC handles
        own, ptrs, self._keep, self.memory = peer_mem.exchange(
            [(2 * self.max_bytes, peer_mem.FINEGRAINED), (lib.ao_allreduce_flag_bytes(), peer_mem.UNCACHED)], self.group, device,
            force_fallback=self._force_coarse)
        peer_mem.require_coherent(self.memory, self.group, device)  # (raises -> ok stays False, `why` says why: RCCL serves the calls)
        self._staging, self._flags = own
        self._state = torch.zeros(lib.ao_allreduce_state_bytes() // 4, dtype=torch.int32, device=device)
        self._scratch = torch.empty(self.max_bytes, dtype=torch.uint8, device=device)  # contiguous, 16-byte aligned stand-in
        torch.cuda.synchronize(device)
        self._data_arr = (ctypes.c_void_p * world)(*ptrs[0])
        self._flag_arr = (ctypes.c_void_p * world)(*ptrs[1])
        self._lib, self._check = lib, _lib.check
        self.rank, self.world = rank, world
        dist.barrier(group=self.group)  # everybody has mapped everybody before the first flag is raised

    def fits(self, t: torch.Tensor) -> bool:
        """Whether `t` goes through the one-shot kernel.  Decided from RANK-INVARIANT properties only (dtype, element count): a
        layout or alignment that differs between ranks must never send one rank to RCCL while the others spin in the kernel --
        non-contiguous or misaligned tensors are copied through a scratch buffer instead."""
        if not (self.ok and t.is_cuda):
            return False
        nbytes = t.numel() * t.element_size()
        if nbytes == 0 or nbytes > self.max_bytes:
            return False
        if self.backend == "hip":
            return t.dtype in self._DT  # (any size: the last 16-byte unit is padded in the scratch buffer)
        return nbytes % 16 == 0 and t.is_contiguous() and t.dtype in (torch.bfloat16, torch.float32)

    def timed_out(self) -> bool:
        """True if any call so far gave up waiting for a peer (host-synchronising read of the status word)."""
        return self.backend == "hip" and self.ok and bool(int(self._state[0].item()))

    def check(self):
        """Raise if a call gave up waiting for a peer -- its output was poisoned (NaN / INT_MIN), not summed.  Host-synchronising:
        call it where the host waits for the device anyway (end of a step, before results leave the GPU)."""
        if self.timed_out():
            raise RuntimeError(f"one-shot all-reduce: a peer of rank {self.rank} did not arrive within {self._lib.ao_collective_timeout_ms()} ms "
                               "(ao_collective_set_timeout_ms); the affected outputs were filled with NaN / INT_MIN")

    def _run(self, t: torch.Tensor, op: int) -> torch.Tensor:
        self.calls += 1
        nbytes = t.numel() * t.element_size()
        direct = t.is_contiguous() and t.data_ptr() % 16 == 0 and nbytes % 16 == 0
        if direct:
            buf, count = t, t.numel()
        else:  # pad to 16 bytes with zeros (SUM: neutral; MAX: the padding is never copied back)
            padded = (nbytes + 15) // 16 * 16
            raw = self._scratch[:padded]
            raw[nbytes:].zero_()
            buf = raw[:nbytes].view(t.dtype)
            buf.copy_(t.reshape(-1))
            count = padded // t.element_size()
        self._check(self._lib.ao_allreduce_oneshot_op(self._data_arr, self._flag_arr, buf.data_ptr(), buf.data_ptr(), self._state.data_ptr(),
                                                      count, self._DT[t.dtype], op, self.max_bytes, self.rank, self.world,
                                                      torch.cuda.current_stream(t.device).cuda_stream))
        if not direct:
            t.copy_(buf.view(t.shape) if t.is_contiguous() else buf.reshape(t.shape))
        return t

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM over the group; returns t."""
        if not self.fits(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t
        if self.backend == "hip":
            return self._run(t, 0)
        nbytes = t.numel() * t.element_size()
        stage = self.buf[:nbytes].view(t.dtype).view(t.shape)
        stage.copy_(t)
        t.copy_(torch.ops.symm_mem.one_shot_all_reduce(stage, "sum", self.group_name))
        return t

    def max_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place elementwise MAX over the group (the row-amax exchange of the exact row-parallel protocol: [M] fp32)."""
        if self.backend == "hip" and self.fits(t):
            return self._run(t, 1)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t


def shard_bounds(size: int, world: int, rank: int, unit: int = 1) -> Tuple[int, int]:
    """[start, end) of rank's contiguous shard of `size`, in multiples of `unit`; the first
    (size/unit) % world ranks get one extra unit.  Raises if size is not a multiple of unit."""
    if size % unit != 0:
        raise ValueError(f"cannot shard {size} in units of {unit}")
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    units = size // unit
    base, extra = divmod(units, world)
    start = rank * base + min(rank, extra)
    return start * unit, (start + base + (1 if rank < extra else 0)) * unit


def shard_unit(weight: torch.Tensor, dim: int) -> int:
    """Smallest slice the weight's packed layout allows along `dim` (0 = N, 1 = K)."""
    block = getattr(weight, "block_size", None)
    name = type(weight).__name__
    if name == "Int4TilePackedTo4dTensor":
        return 16 if dim == 0 else max(128, int(block[-1]))
    if name in ("Int8Tensor", "Float8Tensor"):
        return 16 if dim == 0 else 128
    return 1


class ColumnParallelLinear(nn.Module):
    """y_local = x @ W[n0:n1].T (+ b[n0:n1]); no collective (gather_output=False, Megatron style)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], group=None):
        super().__init__()
        self.group = group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n0, n1 = shard_bounds(weight.shape[0], world, rank, shard_unit(weight, 0))
        self.weight = nn.Parameter(weight[n0:n1], requires_grad=False)
        self.bias = None if bias is None else nn.Parameter(bias[n0:n1].clone(), requires_grad=False)
        self.rows = (n0, n1)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class _GpuBlocks:
    """The exact row-parallel protocol's compute steps on the MI355X kernels (ao_amd.ops)."""

    @staticmethod
    def amax(x):
        from . import ops
        return ops.rowwise_amax(x)

    @staticmethod
    def quantize(kind, x, amax):
        from . import ops
        return (ops.int8_quantize_rowwise_amax if kind == "int8" else ops.fp8_quantize_rowwise_amax)(x, amax)

    @staticmethod
    def partial_mm(kind, xq, w):
        from . import ops
        return ops.int_mm(xq, w.qdata.t()) if kind == "int8" else ops.fp8_mm_f32(xq, w.qdata.t())

    @staticmethod
    def epilogue(kind, acc, xs, w, bias):
        from . import ops
        return (ops.int8_scale_epilogue if kind == "int8" else ops.fp8_scale_epilogue)(acc, xs, w.scale, bias)


def _exact_protocol_blocker(weight) -> Optional[str]:
    """None when `weight`'s activation recipe is the one RowParallelLinear._forward_exact implements, else the reason it is not."""
    act = weight.act_quant_kwargs
    gran = getattr(act, "granularity", None)
    if type(gran).__name__ != "PerRow":
        return f"activation granularity {gran} is not PerRow"
    mapping = getattr(act, "mapping_type", None)
    if mapping is not None and str(getattr(mapping, "name", mapping)).upper() != "SYMMETRIC":
        return f"activation mapping_type {mapping} is not SYMMETRIC"
    if getattr(weight, "act_quant_scale", None) is not None:
        return "static activation scale (act_quant_scale) is set"
    if getattr(act, "hp_value_lb", None) is not None or getattr(act, "hp_value_ub", None) is not None:
        return "activation value bounds (hp_value_lb / hp_value_ub) are set"
    if getattr(weight, "zero_point", None) is not None:
        return "asymmetric weight (zero_point)"
    if weight.scale.numel() != weight.shape[0]:
        return f"weight scale has {weight.scale.numel()} elements, not one per output row (PerTensor / blockwise weights)"
    return None


class RowParallelLinear(nn.Module):
    """y = all_reduce_sum_r( x[..., k0:k1] @ W[:, k0:k1].T ) + b.  `input_is_parallel`: x already
    holds only this rank's K shard (the output of a ColumnParallelLinear).

    8-bit dynamic-activation weights (Int8Tensor / Float8Tensor with act_quant_kwargs) take the exact
    protocol of the module docstring unless reduce="bf16"; everything else (int4 weight-only, plain
    tensors) all-reduces the bf16 partial of F.linear."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], group=None, input_is_parallel: bool = True,
                 reduce: str = "exact", blocks=None, one_shot: Optional["OneShotAllReduce"] = None, scatter_min_rows: int = 128):
        super().__init__()
        # exact protocol at M >= scatter_min_rows (and M % world == 0): reduce-scatter -> epilogue on M / world rows -> all-gather
        self.scatter_min_rows = scatter_min_rows
        self._native_rs = dist.get_backend(group) != "gloo"
        if reduce not in ("exact", "bf16"):
            raise ValueError(f"reduce must be 'exact' or 'bf16', got {reduce!r}")
        self.group = group
        self.input_is_parallel = input_is_parallel
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        k0, k1 = shard_bounds(weight.shape[1], world, rank, shard_unit(weight, 1))
        self.weight = nn.Parameter(weight[:, k0:k1], requires_grad=False)
        self.bias = None if bias is None else nn.Parameter(bias.clone(), requires_grad=False)
        self.cols = (k0, k1)
        name = type(weight).__name__
        self.kind = {"Int8Tensor": "int8", "Float8Tensor": "fp8"}.get(name)
        if self.kind is not None and getattr(weight, "act_quant_kwargs", None) is None:
            self.kind = None
        self.exact = reduce == "exact" and self.kind is not None
        if self.exact:
            # the exact protocol reproduces ONE activation recipe: dynamic, symmetric, one scale per row, no value bounds, per-row
            # weight scales.  Every other variant the subclasses accept (ASYMMETRIC mapping, PerTensor granularity on either operand,
            # static act_quant_scale, hp_value_lb / _ub) would silently compute something else than the unsharded linear -- those
            # take the reference callers' protocol instead (F.linear on the shard + bf16 all-reduce) and say so.
            why = _exact_protocol_blocker(weight)
            if why is not None:
                import warnings

                warnings.warn(f"RowParallelLinear(reduce='exact'): {why}; falling back to reduce='bf16' (locally quantized activation "
                              "shard, bf16 partial sums)", stacklevel=2)
                self.exact = False
        self.blocks = blocks or _GpuBlocks
        self.one_shot = one_shot  # float SUM all-reduces of <= 1 MiB go through it when given (decode-size partials)
        # act_pre_scale (AWQ / SmoothQuant, per input feature): the full vector for a replicated input, the K shard's part else
        self.pre_full = getattr(weight, "act_pre_scale", None)
        self.pre_shard = self.pre_full
        if self.pre_full is not None and self.pre_full.numel() == weight.shape[1]:
            self.pre_shard = self.pre_full.reshape(-1)[k0:k1]

    def _forward_exact(self, x):
        w = self.weight
        k0, k1 = self.cols
        out_dtype = x.dtype
        x2 = x.reshape(-1, x.shape[-1]).to(torch.bfloat16)
        pre = self.pre_shard if self.input_is_parallel else self.pre_full
        if pre is not None:
            x2 = (x2 * pre).to(torch.bfloat16)
        if self.input_is_parallel:
            amax = self.blocks.amax(x2)                       # over this rank's K shard
            if self.one_shot is not None:                     # [M] fp32: the full-K amax, one launch over the peers' buffers
                self.one_shot.max_(amax)
            else:
                dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=self.group)
            x_sh = x2
        else:
            amax = self.blocks.amax(x2)                       # x is replicated: the full-K amax is local
            x_sh = x2[:, k0:k1]
        xq, xs = self.blocks.quantize(self.kind, x_sh, amax)  # the shard of the unsharded qdata, the unsharded scale
        acc = self.blocks.partial_mm(self.kind, xq, w)        # int32 / fp32 [M, N], unscaled
        world = dist.get_world_size(self.group)
        m = acc.shape[0]
        if world > 1 and m >= self.scatter_min_rows and m % world == 0:
            # large M: reduce-scatter the accumulator (each rank ends with the summed rows [r M / W, (r + 1) M / W)), run the scale
            # epilogue on that 1 / W of the rows only, all-gather the bf16 result.  Against all-reduce + full epilogue: the exchange
            # moves (W - 1) / W x M x N x (4 + 2) bytes per rank instead of (W - 1) / W x M x N x 8 (-25 %), the epilogue pass is W times
            # smaller, and the sums are the same sums (int32 exact; fp32 in whatever order the collective adds, as before).
            rows = m // world
            r0 = dist.get_rank(self.group) * rows
            part = self._reduce_scatter(acc, rows, r0)
            y_part = self.blocks.epilogue(self.kind, part, xs[r0 : r0 + rows], w, self.bias)
            y = self._all_gather(y_part.contiguous(), m)
        else:
            self._sum(acc)
            y = self.blocks.epilogue(self.kind, acc, xs, w, self.bias)
        return y.reshape(*x.shape[:-1], y.shape[-1]).to(out_dtype)

    def _reduce_scatter(self, acc, rows, r0):
        """SUM over the group of `acc` [M, N], this rank's row block [r0, r0 + rows) returned.  RCCL: reduce_scatter_tensor (two-shot
        over all xGMI links for large buffers); gloo has no reduce-scatter: all-reduce + slice (CPU tests)."""
        acc = acc.contiguous()
        if self._native_rs:
            part = torch.empty((rows, acc.shape[1]), dtype=acc.dtype, device=acc.device)
            dist.reduce_scatter_tensor(part, acc, op=dist.ReduceOp.SUM, group=self.group)
            return part
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
        return acc[r0 : r0 + rows]

    def _all_gather(self, y_part, m):
        y = torch.empty((m, y_part.shape[1]), dtype=y_part.dtype, device=y_part.device)
        if self._native_rs:
            dist.all_gather_into_tensor(y, y_part, group=self.group)
            return y
        # gloo (CPU tests, the one-GPU dry run of bench.py --gpus 2) knows neither bf16 nor 16-bit integers: move the bytes
        yb, pb = y.view(torch.uint8), y_part.view(torch.uint8)
        try:
            dist.all_gather_into_tensor(yb, pb, group=self.group)
        except RuntimeError:  # a gloo build without the flat variant for this device
            parts = [torch.empty_like(pb) for _ in range(dist.get_world_size(self.group))]
            dist.all_gather(parts, pb, group=self.group)
            yb.copy_(torch.cat(parts, dim=0))
        return y

    def forward(self, x):
        if self.exact:
            return self._forward_exact(x)
        if not self.input_is_parallel:
            x = x[..., self.cols[0] : self.cols[1]]
        y = F.linear(x, self.weight, None)
        # one exchange step per row-parallel linear: bf16 [M, N] partial sums.  On 8 MI355X over
        # xGMI RCCL picks a direct (all links) algorithm for these sizes; the call is asynchronous
        # on the current stream, the next kernel on that stream orders behind it.
        self._sum(y)
        if self.bias is not None:
            y = y + self.bias.to(y.dtype)
        return y

    def _sum(self, t):
        if self.one_shot is not None:
            self.one_shot(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


def shard_linear_(module: nn.Linear, style: str, group=None, input_is_parallel: bool = True, reduce: str = "exact",
                  one_shot: Optional["OneShotAllReduce"] = None) -> nn.Module:
    """Replace an (already quantized or plain) nn.Linear by its TP shard; style "colwise" | "rowwise"
    (the names of torchao/testing/utils.py:370-467's DTensor harness)."""
    bias = module.bias.detach() if module.bias is not None else None
    w = module.weight.detach() if type(module.weight.data) is torch.Tensor else module.weight
    if style == "colwise":
        return ColumnParallelLinear(w, bias, group)
    if style == "rowwise":
        return RowParallelLinear(w, bias, group, input_is_parallel, reduce, one_shot=one_shot)
    raise ValueError(f"unknown TP style {style!r} (colwise | rowwise)")


def tp_mlp(gate_up: nn.Linear, down: nn.Linear, group=None) -> nn.Module:
    """Megatron MLP: merged gate_up column-parallel (each rank holds its gate rows and its up rows),
    SiLU(gate) * up locally, down row-parallel + all-reduce."""

    class _MLP(nn.Module):
        def __init__(self):
            super().__init__()
            w, b = gate_up.weight, gate_up.bias
            half = w.shape[0] // 2
            self.gate = ColumnParallelLinear(w[:half], None if b is None else b[:half], group)
            self.up = ColumnParallelLinear(w[half:], None if b is None else b[half:], group)
            self.down = shard_linear_(down, "rowwise", group)

        def forward(self, x):
            return self.down(F.silu(self.gate(x)) * self.up(x))

    return _MLP()
