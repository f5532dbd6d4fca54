#!/usr/bin/env python3
"""A/B of decode-kernel builds inside ONE process (bench.py's model, hipGraph replay, cold weights).

    python tools/int4_modes.py [--modes 0,90,91,92] [--wpbs 0,4] [--layout merged|five] [--rounds 3]

Tuning modes are compiled into the library for profiling (ao_int4_set_tuning is thread-local); the product dispatch is mode 0.
The wrong-result ablation builds (90, 61S - 64S, 941 - 945) exist only in the laboratory library: build it with
`python -m ao_amd.build --lab` and run this tool with AO_MI355_LIB=tools/bin/_C_mi355_lab.so.
Prints one JSON line per (wpb, mode): tokens/s (median and best of the interleaved rounds), per-shape event-timed kernel
durations, and the largest norm-relative difference of any linear's output against mode 0.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="0,90,91,92")
    ap.add_argument("--wpbs", default="0")
    ap.add_argument("--layout", default="merged")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    shapes = bench.LLAMA3_8B_MERGED if args.layout == "merged" else bench.LLAMA3_8B_UNMERGED
    model = bench.Int4Linears(dev, bench.N_LAYERS, shapes)
    lib = model.lib
    B = args.batch
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    cfgs = [(int(w), int(m)) for w in args.wpbs.split(",") for m in args.modes.split(",")]
    graphs, ref, maxrel = {}, None, {}
    with torch.cuda.stream(stream):
        for cfg in cfgs:
            lib.ao_int4_set_tuning(*cfg)
            model.step(B, sp)
            stream.synchronize()
            ys = [k[1].float().clone() for k in model.io[B][0][: len(shapes)]]
            if ref is None:
                ref = ys
            maxrel[cfg] = max(float((a - b).norm() / b.norm()) for a, b in zip(ys, ref))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                model.step(B, torch.cuda.current_stream().cuda_stream)
            graphs[cfg] = g
        times = {cfg: [] for cfg in cfgs}
        for _ in range(args.rounds):
            for cfg in cfgs:
                g = graphs[cfg]
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(args.steps):
                    g.replay()
                e1.record(stream)
                e1.synchronize()
                times[cfg].append(e0.elapsed_time(e1) / args.steps)
        for cfg in cfgs:
            lib.ao_int4_set_tuning(*cfg)
            durs = np.mean(np.stack([bench.event_profile(lib, model.check, lambda: model.step(B, sp), len(model.weights)) for _ in range(2)]), axis=0)
            per_shape = {}
            for name, n, k in shapes:
                idx = [i for i, w in enumerate(model.weights) if w[4] == name]
                per_shape[name] = round(float(durs[idx].mean()) * 1e3, 2)
            t = sorted(times[cfg])
            print(json.dumps({
                "wpb": cfg[0], "mode": cfg[1], "layout": args.layout, "batch": args.batch,
                "ms_per_step_median": t[len(t) // 2], "ms_per_step_best": t[0],
                "tokens_per_s_median": args.batch * 1e3 / t[len(t) // 2], "tokens_per_s_best": args.batch * 1e3 / t[0],
                "event_us": per_shape, "sum_event_ms": float(durs.sum()), "max_rel_vs_first": maxrel[cfg],
            }), flush=True)
        lib.ao_int4_set_tuning(0, 0)


if __name__ == "__main__":
    main()
