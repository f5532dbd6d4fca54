"""Profiling aid: s_memtime stamps (shader clock) of the grouped MXFP8 weight-streaming kernel under different loads.

    python tools/mx_rb_trace.py N K sizes [alt-library.so | -] [gemm8 variant]     e.g.  14336 4096 32,0,0,0,32,64,0,0
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib  # noqa: E402

if len(sys.argv) > 4 and sys.argv[4] not in ("", "-"):
    _lib.LIB_PATH = os.path.abspath(sys.argv[4])
from ao_amd import ops  # noqa: E402

lib = _lib.lib()
if len(sys.argv) > 5:
    lib.ao_gemm8_set_variant(int(sys.argv[5]))  # e.g. 120: weights through the LDS ring
for kv in filter(None, os.environ.get("AO_GEMM8_TUNE", "").split(",")):  # e.g. AO_GEMM8_TUNE=9=101,10=3 (equal shares, the round-3 meeting)
    key, val = kv.split("=")
    _lib.check(lib.ao_gemm8_set_tuning(int(key), int(val)))
n, k = int(sys.argv[1]), int(sys.argv[2])
sizes = [int(s) for s in sys.argv[3].split(",")]
E, rows = len(sizes), sum(sizes)
dev = "cuda"
a = torch.randn(rows, k, device=dev, dtype=torch.bfloat16)
w = torch.randn(E, n, k, device=dev, dtype=torch.bfloat16) * 0.02
wq, ws = ops.mxfp8_quantize(w)
del w
aq, a_s = ops.mxfp8_quantize(a)
offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=dev)
for _ in range(3):
    ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
gr, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs)
    side.synchronize()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(20):
            ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs)
torch.cuda.synchronize()
gr.replay()
torch.cuda.synchronize()
e0.record()
gr.replay()
e1.record()
torch.cuda.synchronize()
us_graph = e0.elapsed_time(e1) * 50
trace = torch.zeros(16384 * 16, dtype=torch.int64, device=dev)
lib.ao_int4_set_trace(ctypes.c_void_p(trace.data_ptr()))
ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs)
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(0))
t = trace.cpu().view(-1, 16).numpy().astype(np.int64)
t = t[t[:, 0] != 0]
act = sum(1 for s in sizes if s)
d = np.diff(t[:, 2:10], axis=1)
d = d[(t[:, 2:10] != 0).all(axis=1)]
steps = k // 128
if (t[:, 15] != 0).any():  # the stream-K kernel reports the steps of each share
    steps = float(t[:, 15].mean())
    sev = (t[:, 9] != 0) & (t[:, 2] != 0)
    print(f"  stream-K: {steps:.1f} steps per share; first 7 steps {((t[sev, 9] - t[sev, 2]) / 7).mean():.1f} ticks per step; meet + stores {int((t[:, 12] - t[:, 10]).mean())}")
full = trace.cpu().view(-1, 16).numpy().astype(np.int64)
if os.environ.get("AO_TRACE_DUMP"):  # raw stamps, row = workgroup id (blockIdx.x)
    np.save(os.environ["AO_TRACE_DUMP"], full[: max(1, int(np.nonzero(full[:, 0])[0].max()) + 1)])
if (t[:, 3] != 0).all() and (t[:, 4] != 0).all() and (t[:, 15] != 0).any():
    # where the launch's tail comes from: a workgroup's exit time against its id (share position = slab), its XCD (id % 8), its loop rate
    wid = np.nonzero(full[:, 0])[0]
    ex = (t[:, 4] - t[:, 3].min()) / 100.0
    loop = (t[:, 10] - t[:, 2]) / np.maximum(t[:, 15], 1)
    thirds = np.array_split(np.arange(len(wid)), 6)
    print("  exit us by sixth of the grid (share order): " + " ".join(f"{ex[i].mean():.1f}" for i in thirds) + " | loop ticks per step by sixth: " + " ".join(f"{loop[i].mean():.0f}" for i in thirds))
    print("  exit us by XCD (id % 8): " + " ".join(f"{ex[wid % 8 == x].mean():.1f}" for x in range(8)) + " | loop ticks per step by XCD: " + " ".join(f"{loop[wid % 8 == x].mean():.0f}" for x in range(8)))
    print(f"  loop ticks per step p10 {np.percentile(loop, 10):.0f} p50 {np.percentile(loop, 50):.0f} p90 {np.percentile(loop, 90):.0f} max {loop.max():.0f}; tail ticks p10 {np.percentile(t[:, 12] - t[:, 10], 10):.0f} p50 {np.percentile(t[:, 12] - t[:, 10], 50):.0f} p90 {np.percentile(t[:, 12] - t[:, 10], 90):.0f} max {(t[:, 12] - t[:, 10]).max()}")
if (t[:, 3] != 0).all() and (t[:, 4] != 0).all():  # 100 MHz stamps (s_memrealtime) at entry / exit: wall time of the launch and the shader clock inside it
    span_rt = (t[:, 4].max() - t[:, 3].min()) / 100.0
    mhz = ((t[:, 12] - t[:, 0]) / np.maximum(t[:, 4] - t[:, 3], 1) * 100.0)
    print(f"  realtime: launch span {span_rt:.1f} us (first entry -> last exit), workgroup life {((t[:, 4] - t[:, 3]) / 100.0).mean():.1f} us mean / {((t[:, 4] - t[:, 3]) / 100.0).max():.1f} max; "
          f"shader clock while resident {mhz.mean():.0f} MHz (p10 {np.percentile(mhz, 10):.0f}, p90 {np.percentile(mhz, 90):.0f}); entry spread {((t[:, 3] - t[:, 3].min()) / 100.0).max():.1f} us, "
          f"exit p10 {np.percentile(t[:, 4] - t[:, 3].min(), 10) / 100.0:.1f} p50 {np.percentile(t[:, 4] - t[:, 3].min(), 50) / 100.0:.1f} p90 {np.percentile(t[:, 4] - t[:, 3].min(), 90) / 100.0:.1f} us")
print(f"N={n} K={k} sizes={sizes} ({act} active): {us:.1f} us eager back-to-back, {us_graph:.1f} us in a hipGraph; {len(t)} workgroups traced; "
      f"ticks (s_memtime: shader cycles): prime {int((t[:, 1] - t[:, 0]).mean())} (group found at {int((t[:, 13] - t[:, 0]).mean())}, addresses at {int((t[:, 14] - t[:, 0]).mean())}), first data {int((t[:, 2] - t[:, 1]).mean())}, "
      f"per step {d.mean():.1f} (first 7 steps), whole loop {int((t[:, 10] - t[:, 2]).mean())} = {(t[:, 10] - t[:, 2]).mean() / steps:.1f} per step, "
      f"tail {int((t[:, 12] - t[:, 10]).mean())}; launch span {int(t[:, 12].max() - t[:, 0].min())}; "
      f"entry spread p50 {int(np.percentile(t[:, 0] - t[:, 0].min(), 50))} p90 {int(np.percentile(t[:, 0] - t[:, 0].min(), 90))} max {int((t[:, 0] - t[:, 0].min()).max())}; "
      f"exit p10 {int(np.percentile(t[:, 12] - t[:, 0].min(), 10))} p50 {int(np.percentile(t[:, 12] - t[:, 0].min(), 50))} max {int((t[:, 12] - t[:, 0].min()).max())}")
