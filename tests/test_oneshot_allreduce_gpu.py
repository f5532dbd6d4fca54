"""GPU, two processes sharing cuda:0: the hand-written one-shot all-reduce (csrc/allreduce_kernels.hip) over IPC-mapped buffers.

A real world of two on the only hardware available (RCCL refuses two ranks on one device): each process allocates its staging / flag
buffers, the IPC handles are exchanged over gloo, and every rank's kernel stages, signals, waits and reduces exactly as it would
across xGMI -- the peer's memory just happens to live on the same GPU.  Checked bit for bit against the rank-ordered sum computed
on the host, for fp32 / bf16 / int32, decode sizes up to the 1 MiB slot, repeated calls (parity double-buffering), a hipGraph replay
(device-side epochs), and the RowParallelLinear exact protocol running its accumulator all-reduce through it."""
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


def _vec(rank, call, n, dtype):
    g = torch.Generator().manual_seed(1000 * call + rank)
    if dtype == torch.int32:
        return torch.randint(-(2 ** 20), 2 ** 20, (n,), generator=g, dtype=torch.int32)
    return torch.randn(n, generator=g).to(dtype)


def _expected(world, call, n, dtype):
    vs = [_vec(r, call, n, dtype) for r in range(world)]
    if dtype == torch.int32:
        return sum(vs[1:], vs[0].clone())
    acc = torch.zeros(n, dtype=torch.float32)
    for v in vs:  # rank order, fp32 accumulation, one rounding at the end: what the kernel does
        acc = acc + v.float()
    return acc.to(dtype)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    try:
        from ao_amd import parallel

        ar = parallel.OneShotAllReduce(backend="hip", max_bytes=1 << 20)
        out["ok"], out["why"] = ar.ok, ar.why
        if ar.ok:
            bad = []
            call = 0
            for dtype in (torch.float32, torch.bfloat16, torch.int32):
                for n in (4096, 8192 + 64, 1 << 18 if dtype != torch.bfloat16 else 1 << 19):
                    for _ in range(3):  # same shape three times in a row: parity 0, 1, 0
                        call += 1
                        t = _vec(rank, call, n, dtype).cuda()
                        assert ar.fits(t)
                        ar(t)
                        if not torch.equal(t.cpu(), _expected(world, call, n, dtype)):
                            bad.append((str(dtype), n, call))
            out["bad"] = bad
            # hipGraph: one captured all-reduce replayed on fresh inputs (epochs advance on the device)
            buf = torch.zeros(8192, dtype=torch.float32, device="cuda")
            stream = torch.cuda.Stream()
            graph_bad = []
            with torch.cuda.stream(stream):
                call += 1
                buf.copy_(_vec(rank, call, 8192, torch.float32))
                ar(buf)  # warm-up call outside capture (counts as a call on every rank)
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    ar(buf)
                for _ in range(4):
                    call += 1
                    buf.copy_(_vec(rank, call, 8192, torch.float32))
                    g.replay()
                    stream.synchronize()
                    if not torch.equal(buf.cpu(), _expected(world, call, 8192, torch.float32)):
                        graph_bad.append(call)
            out["graph_bad"] = graph_bad
            # unsupported shapes fall through to the group's all_reduce (gloo here)
            odd = torch.ones(5, device="cuda")
            ar(odd)
            out["fallback"] = bool(torch.equal(odd.cpu(), torch.full((5,), float(world))))
            out["timed_out"] = ar.timed_out()
    except Exception as e:  # noqa: BLE001
        import traceback

        out["error"] = repr(e) + traceback.format_exc()[-1500:]
    finally:
        q.put(out)
        dist.destroy_process_group()


def test_oneshot_allreduce_two_processes_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    for o in outs:
        assert "error" not in o, o
        assert o["ok"], f"one-shot set-up failed: {o['why']}"
        assert not o["timed_out"], "a rank waited for its peer beyond the spin bound (the two processes' kernels did not run concurrently)"
        assert o["bad"] == [] and o["graph_bad"] == [] and o["fallback"], o
