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
print(f"N={n} K={k} sizes={sizes} ({act} active): {us:.1f} us eager back-to-back, {us_graph:.1f} us in a hipGraph; {len(t)} workgroups traced; "
      f"ticks (s_memtime: shader cycles): prime {int((t[:, 1] - t[:, 0]).mean())} (group found at {int((t[:, 13] - t[:, 0]).mean())}, addresses at {int((t[:, 14] - t[:, 0]).mean())}), first data {int((t[:, 2] - t[:, 1]).mean())}, "
      f"per step {d.mean():.1f} (first 7 steps), whole loop {int((t[:, 10] - t[:, 2]).mean())} = {(t[:, 10] - t[:, 2]).mean() / steps:.1f} per step, "
      f"tail {int((t[:, 12] - t[:, 10]).mean())}; launch span {int(t[:, 12].max() - t[:, 0].min())}; "
      f"entry spread p50 {int(np.percentile(t[:, 0] - t[:, 0].min(), 50))} p90 {int(np.percentile(t[:, 0] - t[:, 0].min(), 90))} max {int((t[:, 0] - t[:, 0].min()).max())}; "
      f"exit p10 {int(np.percentile(t[:, 12] - t[:, 0].min(), 10))} p50 {int(np.percentile(t[:, 12] - t[:, 0].min(), 50))} max {int((t[:, 12] - t[:, 0].min()).max())}")
