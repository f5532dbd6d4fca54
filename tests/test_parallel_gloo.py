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


class _OracleBlocks:
    """The exact row-parallel protocol's compute steps restated with the numpy oracle (CPU): what the HIP kernels compute
    on the GPU (tests/test_subclass_gpu.py checks those against the same oracle)."""

    @staticmethod
    def amax(x):
        return x.float().abs().amax(dim=1)

    @staticmethod
    def quantize(kind, x, amax):
        import numpy as np
        from oracle import fp8_ref, int8_ref
        f = int8_ref.quantize_rowwise if kind == "int8" else fp8_ref.quantize_rowwise
        q, s = f(x.float().numpy(), amax.numpy())
        return torch.from_numpy(q), torch.from_numpy(np.ascontiguousarray(s)).reshape(-1, 1)

    @staticmethod
    def partial_mm(kind, xq, w):
        from oracle import fp8_ref, int8_ref
        if kind == "int8":
            return torch.from_numpy(int8_ref.int_mm(xq.numpy(), w.qdata.numpy()))
        a = fp8_ref.e4m3_to_f32(xq.numpy()).astype("float64")
        b = fp8_ref.e4m3_to_f32(w.qdata.numpy()).astype("float64")
        return torch.from_numpy((a @ b.T).astype("float32"))

    @staticmethod
    def epilogue(kind, acc, xs, w, bias):
        import numpy as np
        from oracle import bf16
        c = acc.numpy().astype(np.float32)
        xs, ws = xs.numpy().reshape(-1, 1).astype(np.float32), w.scale.numpy().reshape(1, -1).astype(np.float32)
        y = bf16.bf16_round(c * xs) * ws if kind == "int8" else c * xs * ws
        if bias is not None:
            y = y + bias.float().numpy()[None, :]
        return torch.from_numpy(bf16.bf16_round(y.astype(np.float32))).to(torch.bfloat16)


def _worker_8bit(rank, world, port, q):
    """Row-parallel int8 / fp8 dynamic linears over K shards == the UNSHARDED oracle linear (int8 bit for bit)."""
    import numpy as np
    from ao_amd.quantization.float8_tensor import Float8Tensor, QuantizeTensorToFloat8Kwargs
    from ao_amd.quantization.int8_tensor import Int8Tensor, QuantizeTensorToInt8Kwargs
    from oracle import bf16, fp8_ref, int8_ref

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        n, k = 32, 512
        w = bf16.bf16_round((rng.standard_normal((n, k)) * 0.05).astype(np.float32))
        bias = bf16.bf16_round(rng.standard_normal(n).astype(np.float32))
        bt = torch.from_numpy(bias).to(torch.bfloat16)
        out = []
        # m = 5: all-reduce + one epilogue;  m = 6 with scatter_min_rows = 4: reduce-scatter -> epilogue on m / world rows -> all-gather
        for m, scatter_min in ((5, 128), (6, 4)):
            x = bf16.bf16_round(rng.standard_normal((m, k)).astype(np.float32))
            x[:, 300] *= 32.0  # the full-K amax sits in one rank's shard only (power of two: stays bf16)
            xt = torch.from_numpy(x).to(torch.bfloat16)
            for kind, ref, cls, kw in (("int8", int8_ref, Int8Tensor, QuantizeTensorToInt8Kwargs()),
                                       ("fp8", fp8_ref, Float8Tensor, QuantizeTensorToFloat8Kwargs())):
                wq, ws = ref.quantize_rowwise(w)
                wt = cls(torch.from_numpy(wq), torch.from_numpy(np.ascontiguousarray(ws)).reshape(-1, 1), [1, k], torch.bfloat16,
                         act_quant_kwargs=kw)
                want = ref.linear(x, w, bias)
                for parallel_in in (False, True):
                    row = RowParallelLinear(wt, bt, input_is_parallel=parallel_in, blocks=_OracleBlocks, scatter_min_rows=scatter_min)
                    k0, k1 = row.cols
                    y = row(xt[:, k0:k1] if parallel_in else xt).float().numpy()
                    exact = bool(np.array_equal(y, want))
                    rel = float(np.linalg.norm(y - want) / np.linalg.norm(want))
                    out.append((kind, parallel_in, exact, rel))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_row_parallel_8bit_matches_unsharded_oracle_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_8bit, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for _ in range(world):
        rank, out = q.get(timeout=10)
        for kind, parallel_in, exact, rel in out:
            if kind == "int8":
                assert exact, (rank, kind, parallel_in, rel)  # integer partial sums: bit for bit the unsharded linear
            else:
                assert rel <= 1e-6, (rank, kind, parallel_in, rel)  # fp32 partial sums instead of one float64 sum


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


def test_exact_protocol_is_only_taken_for_the_recipe_it_reproduces():
    """ASYMMETRIC / PerTensor / static / bounded activation variants (and PerTensor weights) must not enter the exact
    row-parallel protocol silently -- `_exact_protocol_blocker` names the reason and RowParallelLinear falls back to reduce='bf16'."""
    from ao_amd.parallel import _exact_protocol_blocker
    from ao_amd.quantization.float8_tensor import Float8Tensor, QuantizeTensorToFloat8Kwargs
    from ao_amd.quantization.granularity import PerRow, PerTensor
    from ao_amd.quantization.int8_tensor import Int8Tensor, QuantizeTensorToInt8Kwargs
    from ao_amd.quantization.quant_primitives import MappingType

    n, k = 32, 256
    q8, s8 = torch.zeros(n, k, dtype=torch.int8), torch.ones(n, 1)
    mk8 = lambda kw, **extra: Int8Tensor(q8, extra.pop("scale", s8), [1, k], torch.bfloat16, act_quant_kwargs=kw, **extra)  # noqa: E731
    assert _exact_protocol_blocker(mk8(QuantizeTensorToInt8Kwargs())) is None
    assert "SYMMETRIC" in _exact_protocol_blocker(mk8(QuantizeTensorToInt8Kwargs(mapping_type=MappingType.ASYMMETRIC)))
    assert "PerRow" in _exact_protocol_blocker(mk8(QuantizeTensorToInt8Kwargs(granularity=PerTensor())))
    assert "static" in _exact_protocol_blocker(mk8(QuantizeTensorToInt8Kwargs(), act_quant_scale=torch.ones(1)))
    assert "per output row" in _exact_protocol_blocker(mk8(QuantizeTensorToInt8Kwargs(), scale=torch.ones(1, 1)))
    qf = torch.zeros(n, k, dtype=torch.float8_e4m3fn)
    mkf = lambda kw, sc=s8: Float8Tensor(qf, sc, [1, k], torch.bfloat16, act_quant_kwargs=kw)  # noqa: E731
    assert _exact_protocol_blocker(mkf(QuantizeTensorToFloat8Kwargs(granularity=PerRow()))) is None
    assert "bounds" in _exact_protocol_blocker(mkf(QuantizeTensorToFloat8Kwargs(granularity=PerRow(), hp_value_lb=1e-3)))
    assert "PerRow" in _exact_protocol_blocker(mkf(QuantizeTensorToFloat8Kwargs(granularity=PerTensor())))
