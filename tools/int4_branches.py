#!/usr/bin/env python3
"""A/B: the five-shape decode token as ONE chain of 160 graph nodes vs gate || up as two parallel branches of the graph.

    python tools/int4_branches.py [--rounds 5] [--steps 20]

In a transformer layer gate_proj and up_proj read the same activation and do not depend on each other; qkv -> o -> {gate, up} -> down
is the dependency chain.  Captured on one stream the graph serialises all five (a kernel boundary between gate and up); captured with a
fork (side stream waits on the main stream after o, main waits on the side stream before down) the two projections are independent
nodes and the hardware may run them concurrently.  Same kernels, same bytes, same launches per token (160).
Prints one JSON line per variant: median / best tokens/s over alternating replays.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--mode", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.Int4Linears(dev, bench.N_LAYERS, bench.LLAMA3_8B_UNMERGED)
    if args.mode:
        model.lib.ao_int4_set_tuning(0, args.mode)
    main_s, side_s = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    graphs = {}
    for variant in ("chain", "fork"):
        with torch.cuda.stream(main_s):
            model.step(1, main_s.cuda_stream, side_stream=side_s if variant == "fork" else None)
            main_s.synchronize()
            side_s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_s):
                model.step(1, torch.cuda.current_stream().cuda_stream, side_stream=side_s if variant == "fork" else None)
            graphs[variant] = g
    times = {v: [] for v in graphs}
    with torch.cuda.stream(main_s):
        for _ in range(args.rounds):
            for v, g in graphs.items():
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main_s)
                for _ in range(args.steps):
                    g.replay()
                e1.record(main_s)
                e1.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.steps)
    for v, t in times.items():
        t = sorted(t)
        print(json.dumps({"variant": v, "mode": args.mode, "layout": "five", "ms_per_token_median": t[len(t) // 2], "ms_per_token_best": t[0],
                          "tokens_per_s_median": 1e3 / t[len(t) // 2], "tokens_per_s_best": 1e3 / t[0], "rounds": args.rounds}), flush=True)


if __name__ == "__main__":
    main()
