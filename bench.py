#!/usr/bin/env python3
"""bench.py -- linear-layer tokens/s of the int4 weight-only Llama-3-8B decode path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one token (bs=1, seq=1) through every linear of Llama-3-8B,
Int4WeightOnlyConfig(group_size=128), tile-packed weights (BASELINE.json
configs[1]).  The 218.1 M weights of a layer are laid out the way the serving
stack the reference targets (vLLM) instantiates the model:
    qkv_proj 6144x4096, o_proj 4096x4096, gate_up_proj 28672x4096, down_proj 4096x14336
(merged column-parallel projections are ONE nn.Linear there, so quantize_() sees
one weight and F.linear issues one op).  --unmerged runs gate_proj and up_proj
as two 14336x4096 linears (the HF module layout); same bytes, one more launch
per layer; its tokens/s is also reported in config.unmerged_tokens_per_s.
Every layer owns distinct weights (3.7 GB resident in HBM, far beyond the
256 MiB Infinity Cache), inputs are synthetic and already in HBM, one stream,
launches replayed from a hipGraph.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel (int4_mm_kernel): algorithmic bytes per launch /
                  average kernel duration measured with HIP extension events
  cpu_baseline -- oracle/lowbit_ref.c ("port" of the reference CPU dequant path)
                  timed on the host cores on a bounded sample
N > 1: one process per GPU, each rank decodes its own token stream (the path
partitions by sequence; no data-path collective) -> weak scaling.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (same guide); the dequantised int4 path multiplies in bf16

LLAMA3_8B_MERGED = [  # (name, N, K): vLLM's Llama modules
    ("qkv_proj", 6144, 4096),
    ("o_proj", 4096, 4096),
    ("gate_up_proj", 28672, 4096),
    ("down_proj", 4096, 14336),
]
LLAMA3_8B_UNMERGED = [  # HF module layout (qkv still merged, as SURVEY.md 8d)
    ("qkv", 6144, 4096),
    ("o", 4096, 4096),
    ("gate", 14336, 4096),
    ("up", 14336, 4096),
    ("down", 4096, 14336),
]
N_LAYERS = 32
GROUP = 128


def algorithmic_bytes(m, n, k, g):
    """packed weights + scales/zeros + activation + output."""
    return n * k // 2 + (k // g) * n * 4 + m * k * 2 + m * n * 2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="tokens per step (bs); BASELINE headline is 1")
    ap.add_argument("--layers", type=int, default=N_LAYERS)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--unmerged", action="store_true", help="gate_proj and up_proj as two linears (HF layout)")
    ap.add_argument("--no-second-layout", action="store_true", help="skip the short run of the other module layout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wpb", type=int, default=0, help="tuning: waves per workgroup override")
    ap.add_argument("--mode", type=int, default=0, help="tuning: kernel ablation / depth variant (profiling only)")
    return ap.parse_args()


class Int4Linears:
    """All packed weights of the synthetic model + a raw C-ABI launch list."""

    def __init__(self, device, batch, layers, shapes):
        from ao_amd import _lib, ops

        self.lib = _lib.lib()
        self.check = _lib.check
        self.batch = batch
        self.launches = []  # (x_ptr, q_ptr, sz_ptr, y_ptr, M, N, K, name)
        self.keep = []
        gen = torch.Generator(device=device).manual_seed(0)
        self.shapes = shapes
        for layer in range(layers):
            for name, n, k in shapes:
                # random-init weights of the real shape, quantized by the product kernel
                w = torch.randn(n, k, device=device, dtype=torch.bfloat16, generator=gen) * 0.02
                qdata, sz = ops.int4_quantize_tinygemm(w, GROUP)
                del w
                x = torch.randn(batch, k, device=device, dtype=torch.bfloat16, generator=gen)
                y = torch.empty(batch, n, device=device, dtype=torch.bfloat16)
                self.keep.append((qdata, sz, x, y))
                self.launches.append((x.data_ptr(), qdata.data_ptr(), sz.data_ptr(), y.data_ptr(), batch, n, k, name))
        torch.cuda.synchronize()

    def step(self, stream_ptr):
        f = self.lib.ao_int4_weight_int4pack_mm
        for (xp, qp, sp, yp, m, n, k, _) in self.launches:
            rc = f(xp, qp, sp, yp, m, n, k, GROUP, stream_ptr)
            if rc != 0:
                self.check(rc)

    def bytes_per_step(self):
        return sum(algorithmic_bytes(m, n, k, GROUP) for (_, _, _, _, m, n, k, _) in self.launches)


def profile_kernels(model, stream_ptr):
    """Per-launch kernel durations (ms) of one eager step via HIP extension events."""
    lib = model.lib
    n = len(model.launches)
    model.check(lib.ao_prof_enable(n))
    model.step(stream_ptr)
    buf = (ctypes.c_float * n)()
    cnt = ctypes.c_int(0)
    model.check(lib.ao_prof_collect(buf, n, ctypes.byref(cnt)))
    return np.array(buf[: cnt.value], dtype=np.float64)


def cpu_baseline(batch):
    """Time the C port of the reference CPU dequant->matmul path on ONE layer."""
    from oracle import c_ref

    rng = np.random.default_rng(0)
    threads = c_ref.num_threads()
    t_layer = 0.0
    reps = 0
    budget_s = 12.0
    t_begin = time.perf_counter()
    per_linear = {}
    while True:
        t_layer_once = 0.0
        for name, n, k in LLAMA3_8B_MERGED:
            qdata = rng.integers(-(2**31), 2**31 - 1, size=(n // 8, k // 128, 32, 4), dtype=np.int64).astype(np.int32)
            sz = np.empty((k // GROUP, n, 2), dtype=np.uint16)
            sz[..., 0] = 0x3B00 + rng.integers(0, 64, size=sz.shape[:2])  # scale ~ 2e-3 (bf16 bits)
            sz[..., 1] = 0x3A00 + rng.integers(0, 64, size=sz.shape[:2])  # zero  ~ 5e-4
            x = (rng.standard_normal((batch, k)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
            t0 = time.perf_counter()
            c_ref.int4_linear(x, qdata, sz, n, k, GROUP)
            dt = time.perf_counter() - t0
            per_linear[name] = dt
            t_layer_once += dt
        t_layer += t_layer_once
        reps += 1
        if time.perf_counter() - t_begin > budget_s or reps >= 20:
            break
    t_layer /= reps
    tok_s = batch / (t_layer * N_LAYERS)
    return {
        "value": tok_s,
        "unit": "tokens/s",
        "cores": threads,
        "kind": "port",
        "sample": f"1 of {N_LAYERS} layers (4 merged linears, 218.1M int4 weights, bs={batch}), mean of {reps} reps, x{N_LAYERS} extrapolated; "
        f"oracle/lowbit_ref.c (gcc -O3 -fopenmp, {threads} threads, host has {os.cpu_count()} cpus)",
        "ms_per_layer": t_layer * 1e3,
    }


def build_and_time(args, device, shapes, steps, warmup, dist, world):
    """Build the synthetic model for `shapes`, capture one step into a hipGraph, time `steps` replays.
    Returns (model, stream, elapsed_s, graph_used)."""
    model = Int4Linears(device, args.batch, args.layers, shapes)
    stream = torch.cuda.Stream(device=device)
    sp = stream.cuda_stream
    graph = None
    with torch.cuda.stream(stream):
        model.step(sp)  # first touch (also allocates the library's split-K workspace outside capture)
        stream.synchronize()
        if not args.no_graph:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    model.step(torch.cuda.current_stream().cuda_stream)
            except Exception as e:  # noqa: BLE001
                print(f"warning: hipGraph capture failed ({e!r}); launching eagerly", file=sys.stderr)
                graph = None

    def run_step():
        if graph is not None:
            graph.replay()
        else:
            model.step(sp)

    def sync_all():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    with torch.cuda.stream(stream):
        for _ in range(warmup):
            run_step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        sync_all()
    return model, stream, elapsed, graph is not None


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary
    (profiles/int4_pmc_r01.json, written by scripts/pmc_summary.py from a separate counter run of
    this same command); None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "int4_pmc_r01.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    return d, path


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    from ao_amd import _lib

    lib = _lib.lib()
    if args.wpb or args.mode:
        lib.ao_int4_set_tuning(args.wpb, args.mode)

    shapes = LLAMA3_8B_UNMERGED if args.unmerged else LLAMA3_8B_MERGED
    model, stream, elapsed, graphed = build_and_time(args, device, shapes, args.steps, args.warmup, dist, world)
    sp = stream.cuda_stream
    ms_per_step = elapsed * 1e3 / args.steps
    tokens_per_s = args.batch * world * args.steps / elapsed

    # live per-kernel timing (eager pass, HIP extension events on `stream`)
    prof = None
    if rank == 0:
        with torch.cuda.stream(stream):
            durs = [profile_kernels(model, sp) for _ in range(3)]
        prof = np.mean(np.stack(durs), axis=0)  # ms per launch, launch order

    # the other module layout, for the record (rank 0 of a 1-GPU run only: it doubles resident weights)
    other_tok_s = None
    if rank == 0 and world == 1 and not args.no_second_layout:
        del_model = model  # keep the first model alive until its numbers are computed below
        other_shapes = LLAMA3_8B_MERGED if args.unmerged else LLAMA3_8B_UNMERGED
        steps2 = min(args.steps, 20)
        m2, _, e2, _ = build_and_time(args, device, other_shapes, steps2, min(args.warmup, 3), None, 1)
        other_tok_s = args.batch * steps2 / e2
        del m2
        torch.cuda.empty_cache()

    if rank == 0:
        bytes_step = model.bytes_per_step()
        n_launch = len(model.launches)
        per_shape, kernels = {}, {}
        for name, n, k in shapes:
            idx = [i for i, l in enumerate(model.launches) if l[7] == name]
            b = algorithmic_bytes(args.batch, n, k, GROUP)
            ms = float(prof[idx].mean())
            kern = lib.ao_int4_mm_kernel_name(args.batch, n, k, GROUP).decode()
            per_shape[name] = {"N": n, "K": k, "bytes": b, "us": ms * 1e3, "GBps": b / (ms * 1e-3) / 1e9, "kernel": kern}
            kk = kernels.setdefault(kern, {"launches_per_step": 0, "bytes_per_step": 0, "ms_per_step": 0.0})
            kk["launches_per_step"] += len(idx)
            kk["bytes_per_step"] += b * len(idx)
            kk["ms_per_step"] += float(prof[idx].sum())
        for kk in kernels.values():
            kk["avg_kernel_us"] = kk["ms_per_step"] * 1e3 / kk["launches_per_step"]
            kk["algorithmic_bytes_per_launch"] = kk["bytes_per_step"] / kk["launches_per_step"]
            kk["GBps"] = kk["bytes_per_step"] / (kk["ms_per_step"] * 1e-3) / 1e9
        dom = max(kernels, key=lambda k_: kernels[k_]["ms_per_step"])
        achieved = kernels[dom]["GBps"]
        # batched (bs >= 64) int4-wo is past the HBM ridge: weights are dequantised to bf16 and multiplied on the
        # bf16 MFMA path, so the bounding roofline is the dense bf16 MFMA peak (2.5 PFLOP/s)
        mfma_bound = args.batch >= 64
        flops_step = sum(2.0 * m * n * k for (_, _, _, _, m, n, k, _) in model.launches)
        pmc, pmc_path = pmc_traffic()
        traffic = None
        if pmc is not None and dom in pmc.get("kernels", {}):
            traffic = pmc["kernels"][dom].get("hbm_bytes_per_launch")
        roofline_tok_s = HBM_PEAK_GBS * 1e9 / bytes_step * args.batch
        out = {
            "metric": "linear-layer tokens/sec, Llama-3-8B int4-wo (tinygemm g128), bs=%d" % args.batch,
            "value": tokens_per_s,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 x int4 (dequant bf16, fp32 accumulate)",
            "data": "synthetic (random-init weights of the Llama-3-8B linear shapes, quantized on device)",
            "config": {
                "workload": "Int4WeightOnlyConfig(group_size=128) Llama-3-8B linear shapes, bs=%d seq=1, %d layers x {%s}"
                % (args.batch, args.layers, ", ".join("%s %dx%d" % s_ for s_ in shapes)),
                "module_layout": "HF (gate_proj, up_proj separate)" if args.unmerged else "vLLM (merged qkv_proj, gate_up_proj)",
                "launch": "hipGraph replay" if graphed else "eager",
                "parallelism": "dp%d (one token stream per GPU, no collective)" % world,
                "bytes_per_token": bytes_step,
                "hbm_roofline_tokens_per_s": roofline_tok_s,
                "frac_of_hbm_roofline_end_to_end": (tokens_per_s / world) / roofline_tok_s,
                ("merged_tokens_per_s" if args.unmerged else "unmerged_tokens_per_s"): other_tok_s,
            },
            "roofline": {
                "kernel": dom,
                "bound": "mfma" if mfma_bound else "hbm",
                "achieved": (flops_step / (float(prof.sum()) * 1e-3) / 1e12) if mfma_bound else achieved,
                "peak": MFMA_BF16_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": ((flops_step / (float(prof.sum()) * 1e-3) / 1e12) / MFMA_BF16_PEAK_TFLOPS) if mfma_bound else achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (os.path.relpath(pmc_path, ROOT) if traffic is not None else None),
                "avg_kernel_us": kernels[dom]["avg_kernel_us"],
                "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"],
                "launches_per_step": n_launch,
                "sum_kernel_ms_per_step": float(prof.sum()),
                "all_launches_GBps": (bytes_step / (float(prof.sum()) * 1e-3)) / 1e9,
                "kernels": kernels,
                "per_shape": per_shape,
            },
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
