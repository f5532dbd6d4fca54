"""GPU, two processes sharing cuda:0: the on-device all-to-all-v of MXFP8 token rows (csrc/a2a_kernels.hip, ao_amd.prototype.ep.OnDeviceAllToAllV;
reference torchao/prototype/moe_training/kernels/mxfp8/comms.py:25-460).

A real world of two on the only hardware available: each process owns its staging buffers (rows, scale rows, split vector, flags),
the IPC handles are exchanged over gloo, and every rank's kernel pulls its rows out of the peer's memory exactly as it would across
xGMI.  Checked byte for byte against oracle/moe_ref.py::a2a_v on inputs both processes can regenerate from seeds: every scale-row
size class (16-, 4-, 1-byte units), empty splits, repeated calls (the barriers' epochs), a hipGraph replay with fresh inputs, and
the reference's API shape (cast -> exchange -> dequantize -> slice by output_splits.sum())."""
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


def _inputs(rank, call, world, dim, max_rows):
    """(rows uint8 [T, dim], scales uint8 [T, dim / 32], splits int64 [world]) of `rank` at call `call`: regenerable anywhere."""
    g = np.random.default_rng(7000 * call + 13 * rank + dim)
    splits = g.integers(0, max_rows // (2 * world), size=world).astype(np.int64)
    if call % 3 == 1:
        splits[g.integers(0, world)] = 0  # a rank that receives nothing from this one
    t = int(splits.sum())
    return g.integers(0, 256, size=(t, dim), dtype=np.uint8), g.integers(0, 256, size=(t, dim // 32), dtype=np.uint8), splits


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    per_gpu = os.environ.get("AO_TEST_ONE_GPU_PER_RANK") == "1"  # tests/test_multigpu_gpu.py: one process per GPU over RCCL
    torch.cuda.set_device(rank if per_gpu else 0)
    torch.set_num_threads(2)
    dist.init_process_group("nccl" if per_gpu else "gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    try:
        from ao_amd.prototype import ep
        from oracle import moe_ref

        bad, call = [], 0
        max_rows = 512
        for dim in (512, 256, 96):  # scale rows of 16, 8 and 3 bytes
            ex = ep.OnDeviceAllToAllV(max_rows, dim)
            out.setdefault("ok", []).append((ex.ok, ex.why))
            if not ex.ok:
                continue
            for _ in range(3):
                call += 1
                ins = [_inputs(r, call, world, dim, max_rows) for r in range(world)]
                rows, scales, splits = ins[rank]
                o, os_, osp = ex(torch.from_numpy(rows).cuda(), torch.from_numpy(scales).cuda(), torch.from_numpy(splits).cuda())
                want_rows, want_splits = moe_ref.a2a_v([i[0] for i in ins], [i[2] for i in ins], rank)
                want_scales, _ = moe_ref.a2a_v([i[1] for i in ins], [i[2] for i in ins], rank)
                n = int(want_splits.sum())
                if not (np.array_equal(osp.cpu().numpy(), want_splits) and np.array_equal(o[:n].cpu().numpy(), want_rows)
                        and np.array_equal(os_[:n].cpu().numpy(), want_scales)):
                    bad.append((dim, call))
            out.setdefault("status", []).append(ex.status())
            if dim == 256:  # one captured exchange replayed on fresh inputs (epochs advance on the device)
                stream = torch.cuda.Stream()
                t_cap = 64
                rows_d = torch.zeros((t_cap, dim), dtype=torch.uint8, device="cuda")
                sc_d = torch.zeros((t_cap, dim // 32), dtype=torch.uint8, device="cuda")
                sp_d = torch.zeros(world, dtype=torch.int64, device="cuda")
                graph_bad = []
                with torch.cuda.stream(stream):
                    ex(rows_d, sc_d, sp_d)  # warm-up outside capture (a call on every rank)
                    stream.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream):
                        res = ex(rows_d, sc_d, sp_d)
                    for _ in range(3):
                        call += 1
                        gen = [np.random.default_rng(99 * call + r) for r in range(world)]
                        sp_all = [gg.integers(0, t_cap // world + 1, size=world).astype(np.int64) for gg in gen]
                        sp_all = [np.minimum(s, t_cap // world) for s in sp_all]
                        rows_all = [gg.integers(0, 256, size=(t_cap, dim), dtype=np.uint8) for gg in gen]  # only the first sum(splits) rows count
                        sc_all = [gg.integers(0, 256, size=(t_cap, dim // 32), dtype=np.uint8) for gg in gen]
                        rows_d.copy_(torch.from_numpy(rows_all[rank]))
                        sc_d.copy_(torch.from_numpy(sc_all[rank]))
                        sp_d.copy_(torch.from_numpy(sp_all[rank]))
                        stream.synchronize()
                        dist.barrier()  # (host-side: both ranks have staged before either replays -- the kernel's own barrier then does the rest)
                        g.replay()
                        stream.synchronize()
                        want_rows, want_splits = moe_ref.a2a_v(rows_all, sp_all, rank)
                        n = int(want_splits.sum())
                        if not (np.array_equal(res[2].cpu().numpy(), want_splits) and np.array_equal(res[0][:n].cpu().numpy(), want_rows)):
                            graph_bad.append(call)
                out["graph_bad"] = graph_bad
        out["bad"] = bad
        # the reference's API: bf16 tokens in, dequantized tokens of the peers out
        dim, tokens = 128, 40
        gens = [torch.Generator().manual_seed(500 + r) for r in range(world)]
        xs = [torch.randn(tokens, dim, generator=gg).to(torch.bfloat16) for gg in gens]
        sps = [torch.tensor([3 + ((r + j) % 4) for j in range(world)], dtype=torch.int64) for r in range(world)]
        xs = [x[: int(sp.sum())] for x, sp in zip(xs, sps)]
        got, got_splits = ep.mxfp8_on_device_all_to_all_v(xs[rank].cuda(), sps[rank].cuda(), 256)
        from ao_amd import ops
        from ao_amd.prototype.mx import ScaleCalculationMode, mx_dequantize

        deq = []
        for r in range(world):
            qd, qs = ops.mxfp8_quantize(xs[r].cuda(), ScaleCalculationMode.FLOOR)
            deq.append(mx_dequantize(qs, qd, torch.bfloat16).float().cpu().numpy())
        want, want_splits = moe_ref.a2a_v(deq, [s.numpy() for s in sps], rank)
        out["api"] = bool(np.array_equal(got.float().cpu().numpy(), want) and np.array_equal(got_splits.cpu().numpy(), want_splits))
        # a split vector whose prefix reaches past the sender's staged rows must be clamped and reported, never read
        ex = ep.OnDeviceAllToAllV(64, 128)
        assert ex.ok, ex.why
        rows_d = torch.zeros((8, 128), dtype=torch.uint8, device="cuda")
        sc_d = torch.zeros((8, 4), dtype=torch.uint8, device="cuda")
        lie = torch.full((world,), 40, dtype=torch.int64, device="cuda")  # 40 rows "for every rank" out of 8 staged (capacity 64)
        _, _, osp = ex(rows_d, sc_d, lie)
        torch.cuda.synchronize()
        st = ex.status()
        out["overreach"] = (st, osp.cpu().tolist())
        try:
            ex.check()
            out["overreach_raised"] = False
        except RuntimeError:
            out["overreach_raised"] = True
    except Exception as e:  # noqa: BLE001
        import traceback

        out["error"] = repr(e) + traceback.format_exc()[-1500:]
    finally:
        q.put(out)
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_on_device_all_to_all_v_processes_share_one_gpu(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
    for o in outs:
        assert "error" not in o, o
        assert all(ok for ok, _ in o["ok"]), f"set-up failed: {o['ok']}"
        assert all(s == 0 for s in o["status"]), f"a wait timed out or rows overflowed: {o['status']}"
        assert o["bad"] == [] and o["graph_bad"] == [] and o["api"], o
        # the rows "for rank r" start at row 40 r of every sender: ranks >= 1 would read past the 64 staged rows -> bit 2 (value 4);
        # 40 rows from each of `world` peers overflow the 64-row output from the second peer on -> bit 1 (value 2)
        st, osp = o["overreach"]
        assert (st & 4 if o["rank"] >= 1 else st & 2) and o["overreach_raised"], o["overreach"]
        assert all(0 <= v <= 40 for v in osp) and sum(osp) <= 64, o["overreach"]
