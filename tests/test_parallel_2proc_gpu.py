"""GPU, two processes sharing cuda:0, `gloo` collectives on device tensors: the TP linears of ao_amd/parallel.py with the REAL kernels and a
REAL world of two (RCCL refuses two ranks on one device, so the transport is gloo; everything else -- quantize_, shard_linear_, the
exact row-parallel protocol, the fp32 / int32 accumulator all-reduce -- is what `bench.py --gpus 2` runs).  Rank 0 checks the sharded
result against the UNSHARDED oracle linear."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    per_gpu = os.environ.get("AO_TEST_ONE_GPU_PER_RANK") == "1"  # tests/test_multigpu_gpu.py: one process per GPU over RCCL
    torch.cuda.set_device(rank if per_gpu else 0)
    torch.set_num_threads(2)
    dist.init_process_group("nccl" if per_gpu else "gloo", rank=rank, world_size=world)
    try:
        from ao_amd import parallel
        from ao_amd.quantization import (Float8DynamicActivationFloat8WeightConfig, Int4WeightOnlyConfig,
                                         Int8DynamicActivationInt8WeightConfig, PerRow, quantize_)
        from oracle import bf16, fp8_ref as F8, int4_ref as R4, int8_ref as I8

        dev = "cuda"
        g = torch.Generator().manual_seed(11)  # same tensors on every rank
        hidden, ffn = 512, 1024
        w_up = (torch.randn(ffn, hidden, generator=g) * 0.05).to(torch.bfloat16)
        w_down = (torch.randn(hidden, ffn, generator=g) * 0.05).to(torch.bfloat16)
        res = {}
        # m = 5: amax all-reduce + accumulator all-reduce + one epilogue;  m = 128 (round 4): reduce-scatter -> epilogue on m / world rows ->
        # all-gather, with the real kernels on both ranks (gloo has no reduce-scatter: all-reduce + slice stands in for it)
        for m in (5, 128):
          x = torch.randn(m, hidden, generator=g).to(torch.bfloat16)
          for kind, cfg in (("int8", Int8DynamicActivationInt8WeightConfig()), ("fp8", Float8DynamicActivationFloat8WeightConfig(granularity=PerRow())),
                          ("int4", Int4WeightOnlyConfig(group_size=128, int4_packing_format="tile_packed_to_4d"))):
              up = torch.nn.Linear(hidden, ffn, bias=False, device=dev, dtype=torch.bfloat16)
              down = torch.nn.Linear(ffn, hidden, bias=False, device=dev, dtype=torch.bfloat16)
              with torch.no_grad():
                  up.weight.copy_(w_up)
                  down.weight.copy_(w_down)
              quantize_(up, cfg)
              quantize_(down, cfg)
              col = parallel.shard_linear_(up, "colwise")                              # N split, no exchange
              row = parallel.shard_linear_(down, "rowwise", input_is_parallel=True)    # K split, exact protocol for the 8-bit kinds
              h_local = col(x.to(dev))                                                  # [m, ffn / world]
              y = row(h_local)                                                          # all ranks: [m, hidden]
              parts = [torch.empty_like(h_local) for _ in range(world)]
              dist.all_gather(parts, h_local.contiguous())
              h_seen = torch.cat(parts, dim=1).float().cpu().numpy()                    # the activation the row-parallel linear was given
              # reference on the host: unsharded linears with the oracle's arithmetic
              xn, un, dn = x.float().numpy(), w_up.float().numpy(), w_down.float().numpy()
              if kind == "int8":
                  h = I8.linear(xn, un)
                  yr = I8.linear(h_seen, dn)
              elif kind == "fp8":
                  h = F8.linear(xn, un)
                  yr = F8.linear(h_seen, dn)
              else:
                  def lin4(a, w):
                      s, z = R4.choose_qparams_tinygemm(w, 128)
                      qd = R4.convert_weight_to_int4pack(R4.nibble_pack(R4.quantize_tinygemm(w, s, z, 128)))
                      return R4.weight_int4pack_mm(a, qd, 128, R4.pack_scales_and_zeros(s, z))
                  h = lin4(xn, un)
                  yr = lin4(h_seen, dn)
              n0, n1 = col.rows
              hl = h_local.float().cpu().numpy()
              yn = y.float().cpu().numpy()
              rel_h = float(np.linalg.norm(hl - h[:, n0:n1]) / np.linalg.norm(h[:, n0:n1]))
              rel_y = float(np.linalg.norm(yn - yr) / np.linalg.norm(yr))
              exact = bool(np.array_equal(yn, yr.astype(np.float32)))
              res[(kind, m)] = (rel_h, rel_y, exact)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_tp_linears_two_ranks_one_gpu_vs_unsharded_oracle():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        if p.is_alive():
            p.kill()
            pytest.fail("TP worker did not finish")
        assert p.exitcode == 0
    out = dict(q.get() for _ in range(world))
    for rank in range(world):
        for m in (5, 128):
            r = {k: out[rank][(k, m)] for k in ("int8", "fp8", "int4")}
            assert r["int8"][0] == 0.0 and r["int8"][2], (m, r["int8"])        # int8: column shard and exact row-parallel result bit-exact
            assert r["fp8"][0] <= 1e-3 and r["fp8"][1] <= 1e-3, (m, r["fp8"])  # fp8: within the BASELINE tolerance of the unsharded oracle
            # int4 weight-only: the op's contract is a bf16 output, so each rank's partial sum is rounded to bf16 before the all-reduce adds
            # them in bf16 -- what a caller of the reference's F.linear(x_shard, w_shard) + all_reduce gets too; one bf16 ulp is 3.9e-3
            assert r["int4"][0] <= 1e-3 and r["int4"][1] <= 4e-3, (m, r["int4"])


def _ep_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    per_gpu = os.environ.get("AO_TEST_ONE_GPU_PER_RANK") == "1"  # tests/test_multigpu_gpu.py: one process per GPU over RCCL
    torch.cuda.set_device(rank if per_gpu else 0)
    torch.set_num_threads(2)
    dist.init_process_group("nccl" if per_gpu else "gloo", rank=rank, world_size=world)
    try:
        from ao_amd.prototype.ep import a2a_combine_hp_fwd, a2a_dispatch_mxfp8_fwd, exchange_split_sizes
        from ao_amd.prototype.mx import MXFP8ExpertWeights, _to_mxfp8_then_scaled_grouped_mm
        from oracle import mx_ref as MX

        dev = "cuda"
        torch.manual_seed(5 + rank)
        tokens, dim, n_out = 96, 256, 128
        x = torch.randn(tokens, dim).to(torch.bfloat16).to(dev)
        per_expert = torch.tensor([[32, 0, 40, 24], [16, 48, 0, 32]][rank], device=dev)  # 4 global experts, 2 per rank; rows sorted by expert
        input_splits, output_splits, per_group = exchange_split_sizes(per_expert)
        toks = a2a_dispatch_mxfp8_fwd(x, output_splits, input_splits)           # HIP cast, then the byte exchange
        ref = torch.empty(sum(output_splits), dim, dtype=torch.bfloat16, device=dev)
        dist.all_to_all_single(ref, x, output_splits, input_splits)
        rq, rs = MX.to_mx(ref.float().cpu().numpy(), MX.RCEIL)
        same = bool(np.array_equal(toks.data.view(torch.uint8).cpu().numpy(), rq) and np.array_equal(toks.scale.view(torch.uint8).cpu().numpy(), rs))
        # local experts: received rows are ordered (source rank, local expert); with ONE local expert per call the groups are the sources
        g = torch.Generator().manual_seed(9)
        w = (torch.randn(world, n_out, dim, generator=g) * 0.1).to(torch.bfloat16).to(dev)  # treat each source chunk as a group
        offs = torch.tensor(np.cumsum(output_splits), dtype=torch.int32, device=dev)
        y = _to_mxfp8_then_scaled_grouped_mm(toks, MXFP8ExpertWeights.from_hp(w.transpose(-2, -1)), offs)
        y_hp = _to_mxfp8_then_scaled_grouped_mm(ref, w.transpose(-2, -1), offs)
        back = a2a_combine_hp_fwd(y, input_splits, output_splits)
        q.put((rank, same, bool(torch.equal(y, y_hp)), tuple(back.shape), per_group.tolist()))
    finally:
        dist.destroy_process_group()


def test_ep_dispatch_two_ranks_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_ep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        if p.is_alive():
            p.kill()
            pytest.fail("EP worker did not finish")
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(world))
    for rank, same, same_mm, back_shape, per_group in res:
        assert same, f"rank {rank}: HIP cast + exchange differs from exchange + oracle cast"
        assert same_mm, f"rank {rank}: grouped GEMM on dispatched MXFP8 tokens differs from the bf16-input path"
        assert back_shape == (96, 128)
    assert res[0][4] == [32, 0, 16, 48] and res[1][4] == [40, 24, 0, 32]


def _oneshot_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    per_gpu = os.environ.get("AO_TEST_ONE_GPU_PER_RANK") == "1"  # tests/test_multigpu_gpu.py: one process per GPU over RCCL
    torch.cuda.set_device(rank if per_gpu else 0)
    torch.set_num_threads(2)
    dist.init_process_group("nccl" if per_gpu else "gloo", rank=rank, world_size=world)
    try:
        from ao_amd.parallel import OneShotAllReduce

        red = OneShotAllReduce(max_bytes=1 << 16)  # symmetric memory refuses two ranks on one device: ok must come back False, not raise
        t = torch.arange(4096, device="cuda", dtype=torch.float32) + 1000.0 * rank
        red(t)                                      # ... and the call must fall through to the group's all_reduce
        want = torch.arange(4096, device="cuda", dtype=torch.float32) * 2 + 1000.0
        q.put((rank, red.ok, bool(torch.equal(t, want)), red.why))
    finally:
        dist.destroy_process_group()


def test_one_shot_all_reduce_falls_back_cleanly_when_symmetric_memory_is_unavailable():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_oneshot_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        if p.is_alive():
            p.kill()
            pytest.fail("one-shot worker did not finish")
        assert p.exitcode == 0
    for rank, ok, summed, why in sorted(q.get() for _ in range(world)):
        assert summed, f"rank {rank}: wrong sum (one-shot ok={ok}, {why})"
