#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs into profiles/int4_pmc_r01.json.

    python scripts/pmc_summary.py <dir with *counter_collection.csv> [more dirs] -o profiles/int4_pmc_r01.json

Per kernel (demangled name matched by substring) the mean per-dispatch FETCH_SIZE / WRITE_SIZE.
Units and gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB-ish "kilobytes" (x1024), and on gfx950 FETCH_SIZE counts a wide
coalesced streaming read (16 B per lane, plain or LDS-DMA) at exactly half its bytes -> x2.
WRITE_SIZE is uncalibrated (reported raw x1024).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

KERNELS = ["int4_mm_kernel", "int4_mm_rb_kernel", "int4_mm_w32_kernel", "int4_quantize_kernel", "gemm8_p8_kernel", "gemm8_p8p_kernel", "gemm8_p8h_kernel", "gemm8_dma_kernel", "rb8_kernel", "mx_stream_kernel",
           "stream8_kernel", "dyn8_kernel", "fp8_int4_mm_kernel", "quant_rowwise_reg_kernel", "mxfp8_quant_rowwise_kernel"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--source", default="rocprofv3 --pmc (separate passes per counter), bench.py --steps 2")
    args = ap.parse_args()
    acc = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
    files = []
    for d in args.dirs:
        files += glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                k = next((k for k in KERNELS if k + "<" in name or k + "(" in name), None)
                if k is None:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {"source": args.source, "files": [os.path.basename(f) for f in files],
           "corrections": {"FETCH_SIZE": "x1024 (KB) x2 (gfx950 wide-read undercount)", "WRITE_SIZE": "x1024 (KB), uncalibrated"},
           "kernels": {}}
    for k, ctrs in acc.items():
        e = {"dispatches": max(len(v) for v in ctrs.values())}
        fetch = ctrs.get("FETCH_SIZE")
        write = ctrs.get("WRITE_SIZE")
        if fetch:
            e["FETCH_SIZE_raw_mean"] = sum(fetch) / len(fetch)
            e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE_raw_mean"] * 1024 * 2
        if write:
            e["WRITE_SIZE_raw_mean"] = sum(write) / len(write)
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_raw_mean"] * 1024
        if fetch or write:
            e["hbm_bytes_per_launch"] = e.get("hbm_read_bytes_per_launch", 0.0) + e.get("hbm_write_bytes_per_launch", 0.0)
        # every other counter: mean per dispatch, as reported (SQ_* are summed over the chip's SQs; SQ_WAVE_CYCLES, SQ_WAIT_*,
        # SQ_ACTIVE_INST_* count quad-cycles, SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES count cycles -- MI355X_MICROARCH.md)
        for cname, vals in sorted(ctrs.items()):
            if cname not in ("FETCH_SIZE", "WRITE_SIZE"):
                e[cname + "_mean"] = sum(vals) / len(vals)
        out["kernels"][k] = e
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
