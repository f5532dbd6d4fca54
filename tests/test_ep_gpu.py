"""GPU: the EP dispatch on the HIP cast + RCCL (a world of one: the collective path is real, the exchange degenerate) and the
pre-quantized-token entry of the grouped GEMM (reference: MXTensor input of _to_mxfp8_then_scaled_grouped_mm)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import mx_ref as MX

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def world1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_dispatch_then_grouped_mm_matches_the_unexchanged_path(world1):
    from ao_amd.prototype.ep import MXFP8Tokens, a2a_combine_hp_fwd, a2a_dispatch_mxfp8_fwd, exchange_split_sizes
    from ao_amd.prototype.mx import MXFP8ExpertWeights, _to_mxfp8_then_scaled_grouped_mm

    g = torch.Generator().manual_seed(7)
    sizes = [24, 0, 40, 8]
    E, T, K, N = len(sizes), sum(sizes), 512, 256
    x = torch.randn(T, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(E, N, K, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    counts = torch.tensor(sizes, device=DEV)
    input_splits, output_splits, per_group = exchange_split_sizes(counts)
    assert input_splits == [T] and output_splits == [T] and per_group.tolist() == sizes
    toks = a2a_dispatch_mxfp8_fwd(x, output_splits, input_splits)
    assert isinstance(toks, MXFP8Tokens) and toks.data.is_cuda
    q, s = MX.to_mx(x.float().cpu().numpy(), MX.RCEIL)
    assert np.array_equal(toks.data.view(torch.uint8).cpu().numpy(), q) and np.array_equal(toks.scale.view(torch.uint8).cpu().numpy(), s)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
    B_t = w.transpose(-2, -1)
    y_tok = _to_mxfp8_then_scaled_grouped_mm(toks, B_t, offs)
    y_hp = _to_mxfp8_then_scaled_grouped_mm(x, B_t, offs)
    assert torch.equal(y_tok, y_hp)  # same cast, same kernel
    assert torch.equal(_to_mxfp8_then_scaled_grouped_mm(toks, MXFP8ExpertWeights.from_hp(B_t), offs), y_hp)
    back = a2a_combine_hp_fwd(y_tok, input_splits, output_splits)
    assert torch.equal(back, y_tok)
    sq = 20 * torch.log10(torch.linalg.norm(x.float()) / torch.linalg.norm(x.float() - toks.dequantize().float()))
    assert float(sq) > 30.0  # reference bar, test_a2a_dispatch.py:107


def test_one_shot_all_reduce_world1_or_clean_fallback(world1):
    """The symmetric-memory one-shot all-reduce (ao_amd.parallel.OneShotAllReduce) on a world of one GPU: either the symmetric
    buffer is set up and the op returns the input (sum over one rank), or set-up fails cleanly and the call falls through to
    RCCL -- never a wrong result, never a hang."""
    from ao_amd.parallel import OneShotAllReduce, RowParallelLinear

    red = OneShotAllReduce(max_bytes=1 << 16)
    print("one-shot all-reduce available:", red.ok, red.why)
    for dtype, n in ((torch.bfloat16, 8192), (torch.float32, 4096), (torch.float32, 1 << 20), (torch.int32, 64)):
        t = torch.arange(n, device=DEV).to(dtype) if dtype != torch.bfloat16 else torch.randn(n, device=DEV).to(dtype)
        want = t.clone()
        assert red(t) is t and torch.equal(t, want)
    # through a row-parallel linear (plain bf16 weight: the partial is all-reduced, bias added after)
    w = torch.randn(64, 256, device=DEV, dtype=torch.bfloat16)
    b = torch.randn(64, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(3, 256, device=DEV, dtype=torch.bfloat16)
    lin = RowParallelLinear(w, b, one_shot=red)
    assert torch.allclose(lin(x).float(), torch.nn.functional.linear(x, w, b).float(), atol=2e-2, rtol=2e-2)
