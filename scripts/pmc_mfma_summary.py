#!/usr/bin/env python3
"""MFMA-busy summary of a rocprofv3 --pmc run over the 8-bit GEMMs: per (kernel, grid) the mean of every counter over its dispatches and
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024) (the guide's MFMA-utilisation formula for 256 CUs x 4 SIMDs).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
        --output-format csv -d DIR -o x -- <command>
    python scripts/pmc_mfma_summary.py DIR -o profiles/<name>.json --source "<command>"
"""
import argparse
import collections
import csv
import glob
import json


def label(n):
    if "gemm8_p8h_kernel" in n:
        return "gemm8_p8h_kernel<" + ("int8" if "<0," in n or "<1," in n else "fp8") + ">"
    if "gemm8_p8p_kernel" in n:
        return "gemm8_p8p_kernel<" + ("int8" if "<0>" in n or "<1>" in n else "fp8") + ">"
    if "gemm8_p8_kernel" in n:
        return "gemm8_p8_kernel<" + ("int8" if "<0>" in n or "<1>" in n else "fp8") + ">"
    if "gemm8_dma_kernel" in n:
        return "gemm8_dma_kernel"
    if "rb8_kernel" in n:
        return "rb8_kernel"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--source", default="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...")
    args = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args.dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f, newline="")):
                k = label(row["Kernel_Name"])
                if k is not None:
                    acc[(k, row.get("Grid_Size", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for (k, grid), c in sorted(acc.items()):
        e = {cn: sum(v) / len(v) for cn, v in c.items()}
        e["dispatches"] = max(len(v) for v in c.values())
        if e.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["MfmaUtil"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["SQ_BUSY_CYCLES"] / 32 * 1024)
        out[k + " grid=" + grid] = e
    if out:
        json.dump({"source": args.source, "kernels": out}, open(args.out, "w"), indent=1)
    for k, e in out.items():
        print(k, "MfmaUtil", round(e.get("MfmaUtil", 0), 3), "dispatches", e["dispatches"])


if __name__ == "__main__":
    main()
