#!/usr/bin/env python3
"""Time-boxed random shape sweep through the DEFAULT dispatch of every matmul entry point, at sizes the unit fuzz (tests/test_fuzz_gpu.py)
does not reach: ragged M up to 5000 (the phase-interleaved / persistent 8-bit GEMMs, the 128 x 256 int4 tiles), N and K off the
benchmark's grid.  References are the library's exact building blocks (each pinned to the oracle in tests/): int4 -> ao_int4_dequantize +
fp32 matmul; int8 -> the tiled GEMM (variant 100), bit-equal; fp8 -> fp32 matmul of the codes x scales, 1e-3; MX -> the per-tile kernel
(variant 113).  Prints one JSON line per failure and a summary line; exit code 1 on any failure.

    python tools/fuzz_long.py --seconds 300 --seed 1
"""
import argparse, json, os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops

DEV = "cuda"


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--kinds", default="int4,int8,fp8,mx,dyn,mxdyn,fp8g,f3")
    ap.add_argument("--hog", type=int, default=0, help="1: a second stream copies 512 MiB buffers back and forth the whole time -- memory latencies "
                    "stretch and move, which is what a hand-counted wait that is one request too lenient needs to show itself")
    args = ap.parse_args()
    lib = _lib.lib()
    rng = np.random.default_rng(args.seed)
    kinds = args.kinds.split(",")
    t0 = time.time()
    n_run = {k: 0 for k in kinds}
    fails = []
    names = {}
    hog_stream = hog_evt = hog_a = hog_b = None
    if args.hog:
        hog_stream = torch.cuda.Stream()
        hog_a = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
        hog_b = torch.empty_like(hog_a)
    while time.time() - t0 < args.seconds:
        if args.hog and (hog_evt is None or hog_evt.query()):
            with torch.cuda.stream(hog_stream):
                for _ in range(6):
                    hog_b.copy_(hog_a, non_blocking=True)
                    hog_a.copy_(hog_b, non_blocking=True)
                hog_evt = torch.cuda.Event()
                hog_evt.record(hog_stream)
        kind = kinds[int(rng.integers(len(kinds)))]
        # M: decode, mid, ragged large
        m = int(rng.choice([int(rng.integers(1, 18)), int(rng.integers(17, 300)), int(rng.integers(300, 5000)), int(rng.choice([512, 1024, 2048, 4096]))]))
        n = int(rng.choice([int(rng.integers(1, 64)) * 16, int(rng.integers(8, 128)) * 64, int(rng.choice([1280, 4096, 6144, 7168, 8192, 14336]))]))
        k = int(rng.choice([int(rng.integers(1, 32)) * 128, int(rng.integers(2, 16)) * 512, int(rng.choice([1024, 3584, 4096, 8192, 14336]))]))
        if m * n * k > 2.5e11:
            continue
        gen = torch.Generator(device=DEV).manual_seed(int(rng.integers(1 << 30)))
        try:
            if kind == "int4":
                g = int(rng.choice([g for g in (32, 64, 128, 256) if k % g == 0]))
                w = (torch.randn(n, k, device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
                x = torch.randn(m, k, device=DEV, generator=gen).to(torch.bfloat16)
                qdata, sz = ops.int4_quantize_tinygemm(w, g)
                y = ops.weight_int4pack_mm(x, qdata, g, sz)
                ref = (x.float() @ ops.int4_dequantize(qdata, sz, g).float().t()).to(torch.bfloat16)
                r = rel(y, ref)
                ok = r <= 1e-3 and torch.equal(ops.weight_int4pack_mm(x, qdata, g, sz), y)
                info = {"g": g, "rel": r}
            elif kind in ("int8", "fp8"):
                x = torch.randn(m, k, device=DEV, generator=gen).to(torch.bfloat16)
                w = (torch.randn(n, k, device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
                b = torch.randn(n, device=DEV, generator=gen).to(torch.bfloat16) if rng.integers(2) else None
                if kind == "int8":
                    wq, ws = ops.int8_quantize_rowwise(w)
                    xq, xs = ops.int8_quantize_rowwise(x)
                    y = ops.int8_scaled_mm(xq, xs, wq, ws, b)
                    try:
                        lib.ao_gemm8_set_variant(100)
                        gref = ops.int8_scaled_mm(xq, xs, wq, ws, b)
                    finally:
                        lib.ao_gemm8_set_variant(0)
                    ok = torch.equal(y, gref) and torch.equal(ops.int8_scaled_mm(xq, xs, wq, ws, b), y)
                    info = {"bias": b is not None, "neq": int((y != gref).sum())}
                else:
                    wq, ws = ops.fp8_quantize_rowwise(w)
                    xq, xs = ops.fp8_quantize_rowwise(x)
                    y = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), b)
                    ref = (xq.float() @ wq.float().t()) * xs.reshape(-1, 1).float() * ws.reshape(1, -1).float()
                    if b is not None:
                        ref = ref + b.float()
                    r = rel(y, ref.to(y.dtype))
                    again = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), b)
                    # outputs of a few dozen elements: ONE bf16 rounding flip (the kernel's fp32 sum lands 1e-6 on the other side of a
                    # rounding midpoint) is 3.9e-3 of that element and can push the norm ratio past 1e-3 -- a property of the statistic,
                    # seen 2-5 times per 100 k cases at M <= 5, N <= 256.  There the bar is per element: within half a bf16 ulp of the
                    # fp32 reference + 2e-5 of slack for the summation order.
                    if m * n < 4096 and r > 1e-3:
                        err = (y.float() - ref).abs()
                        half_ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 8)  # half a bf16 ulp of the reference value
                        # + the summation order, priced like the MX parity bar: 2^-16 of sum |a| |b| (sums that cancel carry the error of their terms)
                        mag = (xq.float().abs() @ wq.float().abs().t()) * xs.reshape(-1, 1).float() * ws.reshape(1, -1).float()
                        bound = half_ulp + mag * 2.0 ** -16 + 1e-30
                        close = bool((err <= bound).all())
                    else:
                        close = r <= 1e-3
                    ok = close and torch.equal(again, y)
                    info = {"bias": b is not None, "rel": r, "reproducible": bool(torch.equal(again, y))}
            elif kind == "dyn":
                # the fused cast + linear entry points (one launch at M <= 16) against cast, then scaled mm: same bits
                x = torch.randn(m, k, device=DEV, generator=gen).to(torch.bfloat16)
                w = (torch.randn(n, k, device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
                b = torch.randn(n, device=DEV, generator=gen).to(torch.bfloat16) if rng.integers(2) else None
                wq8, ws8 = ops.int8_quantize_rowwise(w)
                wqf, wsf = ops.fp8_quantize_rowwise(w)
                xq8, xs8 = ops.int8_quantize_rowwise(x)
                xqf, xsf = ops.fp8_quantize_rowwise(x)
                y8, yf = ops.int8_linear(x, wq8, ws8, b), ops.fp8_linear(x, wqf, wsf, b)
                r8, rf = ops.int8_scaled_mm(xq8, xs8, wq8, ws8, b), ops.fp8_scaled_mm(xqf, wqf.t(), xsf, wsf.t(), b)
                ok8 = torch.equal(y8, r8)
                okf = torch.equal(yf, rf)  # (the fused form runs the same kernel with the cast in its prologue: same bits, as tests/test_fuzz_gpu.py asserts)
                rep = torch.equal(ops.int8_linear(x, wq8, ws8, b), y8) and torch.equal(ops.fp8_linear(x, wqf, wsf, b), yf)
                ok = ok8 and okf and rep
                info = {"bias": b is not None, "int8_equal": bool(ok8), "fp8_rel": rel(yf, rf), "reproducible": bool(rep)}
            elif kind == "fp8g":
                # Float8Tensor's _grouped_mm (rowwise): against fp32 matmuls of the codes x scales, group by group; twice for the bits
                e = int(rng.choice([1, 2, 3, 8]))
                sizes = [int(s) for s in rng.choice([0, 1, 5, 16, 31, 33, 64, 70, 140, 300], size=e)]
                if sum(sizes) == 0:
                    sizes[0] = 3
                n, k = min(n, 4096), min(k, 4096)
                m = sum(sizes)
                a = torch.randn(m, k, device=DEV, generator=gen).to(torch.bfloat16)
                w = (torch.randn(e, n, k, device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
                offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
                aq, a_s = ops.fp8_quantize_rowwise(a)
                wq, w_s = ops.fp8_quantize_rowwise(w.reshape(e * n, k))
                wq, w_s = wq.reshape(e, n, k), w_s.reshape(e, n)
                y = ops.fp8_grouped_mm(aq, a_s, wq, w_s, offs)
                ref, mag = torch.zeros(m, n, device=DEV), torch.zeros(m, n, device=DEV)
                lo = 0
                for i, sz in enumerate(sizes):
                    if sz:
                        sc = a_s.reshape(-1, 1)[lo:lo + sz].float() * w_s[i].reshape(1, -1).float()
                        ref[lo:lo + sz] = (aq[lo:lo + sz].float() @ wq[i].float().t()) * sc
                        mag[lo:lo + sz] = (aq[lo:lo + sz].float().abs() @ wq[i].float().abs().t()) * sc
                    lo += sz
                r = rel(y, ref.to(y.dtype))
                if m * n < 4096 and r > 1e-3:
                    half_ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 8)
                    close = bool(((y.float() - ref).abs() <= half_ulp + mag * 2.0 ** -16 + 1e-30).all())
                else:
                    close = r <= 1e-3
                again = ops.fp8_grouped_mm(aq, a_s, wq, w_s, offs)
                ok = close and torch.equal(again, y)
                info = {"sizes": sizes, "rel": r, "reproducible": bool(torch.equal(again, y))}
            elif kind == "f3":
                # fp8 activations x PLAIN int4 weights: the fused-cast call against cast + matmul (same bits where the fused form fits), twice
                from ao_amd.quantization.int4_plain_tensor import Int4Tensor
                g = int(rng.choice([g for g in (32, 64, 128, 256) if k % g == 0]))
                n = min(n, 8192)
                m = min(m, 600)
                w = (torch.randn(n, k, device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
                x = torch.randn(m, k, device=DEV, generator=gen).to(torch.bfloat16)
                wt = Int4Tensor.from_hp(w, [1, g], activation_dtype=torch.float8_e4m3fn)
                qdata_tp, sz = wt.tile_packed()
                xq, xs = ops.fp8_quantize_rowwise(x)
                two = ops.fp8_int4_linear(xq, xs, qdata_tp, sz, g)
                one = ops.fp8_int4_act_linear(x, qdata_tp, sz, g, fused=True)  # (the one-launch form wherever the kernel takes the shape)
                ok = torch.equal(one, two) and torch.equal(ops.fp8_int4_linear(xq, xs, qdata_tp, sz, g), two) and torch.equal(ops.fp8_int4_act_linear(x, qdata_tp, sz, g, fused=True), one)
                info = {"g": g, "fused_equal": bool(torch.equal(one, two))}
            elif kind == "mxdyn":
                e = int(rng.choice([1, 2, 8]))
                sizes = [int(s) for s in rng.choice([0, 0, 1, 5, 16, 31, 33, 48], size=e)]
                if sum(sizes) == 0:
                    sizes[0] = 3
                n = min(n, 4096)
                k = max(512, (min(k, 4096) // 512) * 512)
                mtot = sum(sizes)
                m = mtot
                a = torch.randn(mtot, k, device=DEV, generator=gen).to(torch.bfloat16)
                w1 = (torch.randn(e, n, k, device=DEV, generator=gen) * 0.1).to(torch.bfloat16)
                w3 = (torch.randn(e, n, k, device=DEV, generator=gen) * 0.1).to(torch.bfloat16)
                offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
                mode = "rceil" if rng.integers(2) else "floor"
                aq, a_s = ops.mxfp8_quantize(a, mode)
                w1q, w1s = ops.mxfp8_quantize(w1, "rceil")
                w3q, w3s = ops.mxfp8_quantize(w3, "rceil")
                y1 = ops.mxfp8_grouped_mm(aq, a_s, w1q, w1s, offs)
                y3 = ops.mxfp8_grouped_mm(aq, a_s, w3q, w3s, offs)
                ok, info = True, {"sizes": sizes, "mode": mode}
                if ops.mxfp8_grouped_mm_dyn_fits(mtot, n, k, e):
                    d1 = ops.mxfp8_grouped_mm_dyn(a, w1q, w1s, offs, mode)
                    ok = ok and torch.equal(d1[:mtot], y1[:mtot])
                    info["dyn_equal"] = bool(torch.equal(d1[:mtot], y1[:mtot]))
                if ops.mxfp8_grouped_mm_pair_fits(mtot, n, k, e):
                    p1, p3 = ops.mxfp8_grouped_mm_pair(a, w1q, w1s, w3q, w3s, offs, mode)
                    # (the pair launch cuts the tiles into other shares than the single launches: a cut tile's pieces are added in k order, so
                    # single elements may round the other way -- at most a bf16 ulp, a handful of elements; anything more is a bug:
                    # profiles/mx_pair_race_r06.jsonl)
                    # (an output that is tiny next to its products -- cancellation -- moves by many of ITS ulps when the fp32 order changes: such
                    # elements are judged against the products' magnitude, sum |a||b| x 2^-18, like the other kinds' tiny outputs; the hog run of
                    # gpurun_out/s32 flagged one such element, 16 of its own ulps, in 128 k cases)
                    eq = True
                    for pp, yy, ww in ((p1[:mtot], y1[:mtot], w1), (p3[:mtot], y3[:mtot], w3)):
                        ulp = torch.exp2(torch.floor(torch.log2(yy.float().abs().clamp_min(1e-30))) - 7)
                        d = (pp.float() - yy.float()).abs()
                        info["pair_differ"] = max(info.get("pair_differ", 0), int((pp != yy).sum()))
                        info["pair_max_ulps"] = max(info.get("pair_max_ulps", 0.0), float((d / ulp).max()))
                        within = bool((d <= 2 * ulp).all())
                        if not within:
                            mag, lo = torch.zeros_like(d), 0
                            for i, sz in enumerate(sizes):
                                if sz:
                                    mag[lo:lo + sz] = a[lo:lo + sz].float().abs() @ ww[i].float().abs().t()
                                lo += sz
                            within = bool((d <= 2 * ulp + mag * 2.0 ** -18).all())
                            info["pair_tiny_outputs"] = int((d > 2 * ulp).sum())
                        eq = eq and within and int((pp != yy).sum()) <= max(8, pp.numel() // 2000)
                    again1, again3 = ops.mxfp8_grouped_mm_pair(a, w1q, w1s, w3q, w3s, offs, mode)
                    eq = eq and torch.equal(again1[:mtot], p1[:mtot]) and torch.equal(again3[:mtot], p3[:mtot])
                    ok = ok and eq
                    info["pair_ok"] = bool(eq)
            else:
                e = int(rng.choice([1, 2, 3, 8, 16]))
                sizes = [int(s) for s in rng.choice([0, 0, 1, 5, 16, 31, 33, 64, 70, 140, 300], size=e)]
                if sum(sizes) == 0:
                    sizes[0] = 3
                n = min(n, 4096)
                k = min(k, 4096)
                mtot = sum(sizes)
                a = torch.randn(mtot, k, device=DEV, generator=gen).to(torch.bfloat16)
                w = (torch.randn(e, n, k, device=DEV, generator=gen) * 0.1).to(torch.bfloat16)
                offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
                aq, a_s = ops.mxfp8_quantize(a, "rceil")
                wq, w_s = ops.mxfp8_quantize(w, "rceil")
                y = ops.mxfp8_grouped_mm(aq, a_s, wq, w_s, offs)
                try:
                    lib.ao_gemm8_set_variant(113)
                    y_old = ops.mxfp8_grouped_mm(aq, a_s, wq, w_s, offs)
                finally:
                    lib.ao_gemm8_set_variant(0)
                r = rel(y, y_old)
                again = ops.mxfp8_grouped_mm(aq, a_s, wq, w_s, offs)
                ok = r <= 1e-3 and torch.equal(again, y)
                info = {"sizes": sizes, "rel": r, "reproducible": bool(torch.equal(again, y))}
                m = mtot
            torch.cuda.synchronize()
        except Exception as ex:  # a refusal is a finding too: say which shape
            ok, info = False, {"exception": repr(ex)[:300]}
            lib.ao_gemm8_set_variant(0)
        n_run[kind] += 1
        if not ok:
            row = {"fail": kind, "M": m, "N": n, "K": k, **info}
            fails.append(row)
            print(json.dumps(row), flush=True)
    print(json.dumps({"summary": "fuzz_long", "hog": args.hog, "seed": args.seed, "seconds": round(time.time() - t0, 1), "cases": n_run, "failures": len(fails)}), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
