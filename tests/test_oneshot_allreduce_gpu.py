"""GPU, 2 / 4 / 8 processes sharing cuda:0: the hand-written one-shot all-reduce (csrc/allreduce_kernels.hip) over IPC-mapped buffers.

A real world of up to eight (the kernel's kMaxWorld) on the only hardware available (RCCL refuses two ranks on one device): each process allocates its staging / flag
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


def _expected(world, call, n, dtype, op="sum"):
    vs = [_vec(r, call, n, dtype) for r in range(world)]
    if op == "max":
        acc = vs[0].clone()
        for v in vs[1:]:
            acc = torch.maximum(acc, v)
        return acc
    if dtype == torch.int32:
        return sum(vs[1:], vs[0].clone())
    acc = torch.zeros(n, dtype=torch.float32)
    for v in vs:  # rank order, fp32 accumulation, one rounding at the end: what the kernel does
        acc = acc + v.float()
    return acc.to(dtype)


def _worker(rank, world, port, q, coarse=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    per_gpu = os.environ.get("AO_TEST_ONE_GPU_PER_RANK") == "1"  # tests/test_multigpu_gpu.py: one process per GPU over RCCL
    torch.cuda.set_device(rank if per_gpu else 0)
    torch.set_num_threads(2)  # up to eight of these share the host: the checker's CPU tensors must not fan out over every core each
    dist.init_process_group("nccl" if per_gpu else "gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    try:
        from ao_amd import parallel

        ar = parallel.OneShotAllReduce(backend="hip", max_bytes=1 << 20, force_coarse_grained=coarse)
        out["ok"], out["why"], out["memory"] = ar.ok, ar.why, ar.memory
        if ar.ok:
            bad = []
            call = 0
            for dtype in (torch.float32, torch.bfloat16, torch.int32):
                for n in (4096, 8192 + 64, 1 << 18 if dtype != torch.bfloat16 else 1 << 19):
                    for _ in range(3 if world <= 2 else 2):  # same shape in a row: parity 0, 1(, 0) -- eight processes time-share one GPU, 80 s at three
                        call += 1
                        t = _vec(rank, call, n, dtype).cuda()
                        assert ar.fits(t)
                        ar(t)
                        if not torch.equal(t.cpu(), _expected(world, call, n, dtype)):
                            bad.append((str(dtype), n, call))
            # MAX (the row-amax exchange), incl. a length that is not a multiple of 16 bytes (padded through the scratch buffer) and a
            # non-contiguous view (rank-invariant `fits`: copied, never a different collective on one rank)
            for n in (5, 128, 2048 + 3):
                call += 1
                t = _vec(rank, call, n, torch.float32).cuda()
                ar.max_(t)
                if not torch.equal(t.cpu(), _expected(world, call, n, torch.float32, "max")):
                    bad.append(("max", n, call))
            call += 1
            wide = torch.zeros(64, 2, device="cuda")
            wide[:, 0] = _vec(rank, call, 64, torch.float32).cuda()
            col = wide[:, 0]
            assert not col.is_contiguous() and ar.fits(col)
            ar(col)
            if not torch.equal(wide[:, 0].cpu(), _expected(world, call, 64, torch.float32)) or float(wide[:, 1].abs().sum()) != 0.0:
                bad.append(("strided", 64, call))
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
            # unsupported dtypes / sizes fall through to the group's all_reduce (gloo here)
            odd = torch.ones(5, device="cuda", dtype=torch.float64)
            assert not ar.fits(odd)
            ar(odd)
            out["fallback"] = bool(torch.equal(odd.cpu(), torch.full((5,), float(world), dtype=torch.float64)))
            out["timed_out"] = ar.timed_out()
            ar.check()
    except Exception as e:  # noqa: BLE001
        import traceback

        out["error"] = repr(e) + traceback.format_exc()[-1500:]
    finally:
        q.put(out)
        dist.destroy_process_group()


def _run(world, target, args=(), timeout=600):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(120)
    return outs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_oneshot_allreduce_processes_share_one_gpu(world):
    for o in _run(world, _worker):
        assert "error" not in o, o
        assert o["ok"], f"one-shot set-up failed: {o['why']}"
        assert not o["timed_out"], "a rank waited for its peer beyond the timeout (the processes' kernels did not run concurrently)"
        assert o["bad"] == [] and o["graph_bad"] == [] and o["fallback"], o
    # every rank reports the same allocation mode; on this stack it is the C-ABI one (recorded for DESIGN.md section 6)
    assert len({o["memory"] for o in _run(2, _worker)}) == 1


def test_oneshot_allreduce_coarse_grained_fallback_path():
    """The round-3 allocation (torch allocator + CUDA-IPC storage sharing) stays reachable and correct: it is what every rank falls back
    to together when hipExtMallocWithFlags memory cannot be exported."""
    for o in _run(2, _worker, (True,)):
        assert "error" not in o and o["ok"], o
        assert o["memory"].startswith("coarse-grained fallback"), o["memory"]
        assert o["bad"] == [] and o["graph_bad"] == [] and o["fallback"] and not o["timed_out"], o


def _late_worker(rank, world, port, q):
    """Rank 1 never calls: rank 0's kernel must give up after the (shortened) timeout, poison its output and report it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    per_gpu = os.environ.get("AO_TEST_ONE_GPU_PER_RANK") == "1"  # tests/test_multigpu_gpu.py: one process per GPU over RCCL
    torch.cuda.set_device(rank if per_gpu else 0)
    dist.init_process_group("nccl" if per_gpu else "gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    try:
        from ao_amd import _lib, parallel

        ar = parallel.OneShotAllReduce(backend="hip", max_bytes=1 << 16)
        assert ar.ok, ar.why
        _lib.check(_lib.lib().ao_collective_set_timeout_ms(200))
        if rank == 0:
            f = torch.ones(1024, device="cuda")
            i = torch.ones(1024, device="cuda", dtype=torch.int32)
            ar(f)
            ar(i)
            torch.cuda.synchronize()
            out["poisoned"] = bool(torch.isnan(f).all()) and bool((i == -(2 ** 31)).all())
            out["timed_out"] = ar.timed_out()
            try:
                ar.check()
                out["raised"] = False
            except RuntimeError as e:
                out["raised"] = "did not arrive" in str(e)
        dist.barrier()
    except Exception as e:  # noqa: BLE001
        import traceback

        out["error"] = repr(e) + traceback.format_exc()[-1500:]
    finally:
        q.put(out)
        dist.destroy_process_group()


def test_oneshot_allreduce_late_peer_poisons_and_raises():
    """a peer that does not arrive must never yield a plausible wrong sum."""
    outs = {o["rank"]: o for o in _run(2, _late_worker)}
    assert "error" not in outs[0] and "error" not in outs[1], outs
    assert outs[0]["poisoned"] and outs[0]["timed_out"] and outs[0]["raised"], outs[0]
