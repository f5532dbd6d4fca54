#!/usr/bin/env python3
"""Table of a tools/midm_sweep.py output: per (kind, shape, M) ours / core in microseconds, the ratio and the kernel; wins counted."""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
Ms = sorted({r["M"] for r in rows if "M" in r})
for kind in sorted({r["kind"] for r in rows if "kind" in r}):
    core, ours, shapes = {}, {}, []
    for r in rows:
        if r.get("kind") != kind or "form" not in r:
            continue
        if r["shape"] not in shapes:
            shapes.append(r["shape"])
        k = (r["shape"], r["M"])
        if r["form"] == "core":
            core[k] = r.get("us")
        else:
            ours[k] = (r.get("us"), r.get("kernel"))
    print(kind)
    print("%-13s" % "shape", *["%22d" % m for m in Ms])
    win = tot = 0
    for s in shapes:
        out = []
        for m in Ms:
            c, o = core.get((s, m)), ours.get((s, m))
            if c and o and o[0]:
                out.append("%6.1f/%6.1f %4.2f %-4s" % (o[0], c, c / o[0], (o[1] or "").replace("gemm8_", "")[:4]))
                tot += 1
                win += c / o[0] >= 1.0
            else:
                out.append("%22s" % "-")
        print("%-13s" % s, *out)
    print("ours >= core in", win, "of", tot, "cells")
