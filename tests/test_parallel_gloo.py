"""CPU, world_size 2, gloo: the TP shard planning and the one all-reduce per row-parallel linear.
The per-rank GEMM here is plain torch on CPU tensors (the sharding/collective logic is what is under
test; the quantized kernels need a GPU and are covered by the -m gpu suite)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ao_amd.parallel import ColumnParallelLinear, RowParallelLinear, shard_bounds, shard_linear_, tp_mlp


def test_shard_bounds_cover_and_align():
    for size, world, unit in [(4096, 8, 16), (14336, 8, 128), (6144, 4, 16), (28672, 8, 128), (1792 * 2, 3, 128)]:
        spans = [shard_bounds(size, world, r, unit) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == size
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0 and a0 % unit == 0 and a1 % unit == 0
        assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= unit
    with pytest.raises(ValueError):
        shard_bounds(100, 4, 0, 16)
    with pytest.raises(ValueError):
        shard_bounds(128, 4, 4, 16)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)  # same weights on every rank
        hidden, ffn = 256, 512
        gate_up = torch.nn.Linear(hidden, 2 * ffn, bias=False)
        down = torch.nn.Linear(ffn, hidden, bias=True)
        qkv = torch.nn.Linear(hidden, 384, bias=True)
        x = torch.randn(3, hidden)
        # reference: unsharded
        g, u = gate_up(x).chunk(2, dim=-1)
        y_ref = down(torch.nn.functional.silu(g) * u)
        # column-parallel: local rows, no collective
        col = shard_linear_(qkv, "colwise")
        n0, n1 = col.rows
        assert torch.allclose(col(x), qkv(x)[:, n0:n1], atol=1e-5)
        # row-parallel with a replicated input: slices its K shard, all-reduces
        row = RowParallelLinear(down.weight.detach(), down.bias.detach(), input_is_parallel=False)
        h = torch.randn(3, ffn)
        assert torch.allclose(row(h), down(h), atol=1e-4)
        # Megatron MLP: column-parallel gate/up, row-parallel down, ONE all-reduce
        mlp = tp_mlp(gate_up, down)
        y = mlp(x)
        q.put((rank, bool(torch.allclose(y, y_ref, atol=1e-4)), float((y - y_ref).abs().max())))
    finally:
        dist.destroy_process_group()


def test_tp_mlp_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=10) for _ in range(world))
    assert [r[0] for r in results] == [0, 1]
    assert all(r[1] for r in results), results
