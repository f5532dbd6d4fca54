"""CPU, world_size 2, gloo: the expert-parallel token exchange of ao_amd/prototype/ep.py (reference
torchao/prototype/moe_training/ep/a2a_dispatch.py, a2a_combine.py; test model: test/prototype/moe_training/ep/test_a2a_dispatch.py).
The MXFP8 cast is the numpy oracle here (the HIP cast needs a GPU and is pinned to the same oracle by the -m gpu suite); what is
under test is the split bookkeeping and that quantize-then-exchange equals exchange-then-quantize row for row."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mx_ref as MX


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_cast(x, mode):
    q, s = MX.to_mx(x.float().numpy(), MX.RCEIL if "rceil" in str(mode).lower() else MX.FLOOR)
    return torch.from_numpy(q).view(torch.float8_e4m3fn), torch.from_numpy(s).view(torch.float8_e8m0fnu)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ao_amd.prototype.ep import MXFP8Tokens, a2a_combine_hp_fwd, a2a_dispatch_mxfp8_fwd, exchange_split_sizes

        torch.manual_seed(42 + rank)
        tokens, dim, e_global = 64, 128, 4  # two local experts per rank
        x = torch.randn(tokens, dim).to(torch.bfloat16)
        x[3] *= 300.0
        x[5] = 0
        # ragged routing, different on every rank, an empty expert included; rows already ordered by (destination rank, expert)
        per_expert = torch.tensor([[10, 0, 30, 24], [17, 25, 21, 1]][rank])
        assert int(per_expert.sum()) == tokens
        input_splits, output_splits, per_group = exchange_split_sizes(per_expert)
        assert input_splits == per_expert.view(world, -1).sum(1).tolist()
        out = a2a_dispatch_mxfp8_fwd(x, output_splits, input_splits, cast=_oracle_cast)
        assert isinstance(out, MXFP8Tokens) and out.data.dtype == torch.float8_e4m3fn and out.scale.dtype == torch.float8_e8m0fnu
        assert out.shape == (sum(output_splits), dim) and out.scale.shape == (sum(output_splits), dim // 32)
        # reference of the reference's test: all-to-all of the bf16 tokens
        ref = torch.empty(sum(output_splits), dim, dtype=torch.bfloat16)
        dist.all_to_all_single(ref, x, output_splits, input_splits)
        rq, rs = _oracle_cast(ref, "rceil")
        same = bool(torch.equal(out.data.view(torch.uint8), rq.view(torch.uint8)) and torch.equal(out.scale.view(torch.uint8), rs.view(torch.uint8)))
        deq = MX.mx_dequant_bf16(out.data.view(torch.uint8).numpy(), out.scale.view(torch.uint8).numpy())
        r = ref.float().numpy()
        sqnr = float(20 * np.log10(np.linalg.norm(r) / np.linalg.norm(r - deq)))
        # the way back: expert outputs (here: the received rows themselves, in bf16) return to their source ranks in order
        back = a2a_combine_hp_fwd(ref, input_splits, output_splits)
        q.put((rank, same, sqnr, bool(torch.equal(back, x)), per_group.tolist()))
    finally:
        dist.destroy_process_group()


def test_ep_dispatch_and_combine_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(world))
    for rank, same, sqnr, back_ok, per_group in res:
        assert same, f"rank {rank}: quantize-then-exchange differs from exchange-then-quantize"
        assert sqnr > 30.0, (rank, sqnr)  # the reference's bar (test_a2a_dispatch.py:107)
        assert back_ok, f"rank {rank}: combine did not return the rows to their source"
    # rank 0 hosts experts 0, 1: it receives [10, 0] from itself and [17, 25] from rank 1 (and rank 1: [30, 24], [21, 1])
    assert res[0][4] == [10, 0, 17, 25] and res[1][4] == [30, 24, 21, 1]


def _a2a_oracle_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import moe_ref

        ok = True
        for call in range(4):
            # every rank can regenerate every rank's rows and split vector (what the GPU test of the on-device kernel relies on)
            gens = [np.random.default_rng(1000 * call + r) for r in range(world)]
            splits = [g.integers(0, 9, size=world).astype(np.int64) for g in gens]
            if call == 2:
                splits[0][1] = 0
            rows = [g.integers(0, 256, size=(int(s.sum()), 48), dtype=np.uint8) for g, s in zip(gens, splits)]
            want, want_splits = moe_ref.a2a_v(rows, splits, rank)
            # the collective the reference's own test checks its on-device kernel against (test_comms.py: all_to_all_single with the
            # split sizes on the host)
            out_splits = torch.empty(world, dtype=torch.int64)
            dist.all_to_all_single(out_splits, torch.from_numpy(splits[rank]))
            got = torch.empty((int(out_splits.sum()), 48), dtype=torch.uint8)
            dist.all_to_all_single(got, torch.from_numpy(rows[rank]), out_splits.tolist(), splits[rank].tolist())
            ok = ok and np.array_equal(out_splits.numpy(), want_splits) and np.array_equal(got.numpy(), want)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_a2a_v_oracle_equals_all_to_all_single_world2():
    """oracle/moe_ref.py::a2a_v -- the checker of the on-device all-to-all-v kernel (tests/test_ondevice_a2a_gpu.py) -- pinned against
    torch.distributed.all_to_all_single with host-side splits: the equivalence the reference's kernel is defined by
    (torchao/prototype/moe_training/kernels/mxfp8/comms.py:25-60, 405-460)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_a2a_oracle_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(world)) == [(0, True), (1, True)]
