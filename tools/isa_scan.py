#!/usr/bin/env python3
"""Static scan of the built library's gfx950 code (no GPU needed): for every kernel, the innermost loop that holds its MFMAs -- length,
instruction mix, waits -- and the op sequence of that loop in one line (M big MFMA, m 4x4x4 MFMA, r ds_read, w ds_write, D LDS-DMA,
L global / buffer load, S store, | barrier, . s_waitcnt, n s_nop, v other VALU).  Two round-3 performance bugs were visible only here:
the fp8 GEMM's MFMAs sunk out of their phases, and the wide-buffer-store data-register hazard.

    python tools/isa_scan.py [filter-regex] > profiles/isa_scan_<round>.txt
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels():
    d = tempfile.mkdtemp()
    shutil.copy(os.path.join(ROOT, "ao_amd", "_C_mi355.so"), os.path.join(d, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, capture_output=True, check=True)
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f], cwd=d, capture_output=True, text=True, check=True).stdout
        for part in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", asm):
            m = re.match(r"[0-9a-f]+ <([^>]+)>:", part)
            if not m:
                continue
            ins, addr = [], []
            for l in part.split("\n")[1:]:
                mm = re.match(r"\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
                if mm:
                    ins.append(mm.group(1))
                    addr.append(int(mm.group(2), 16))
            yield m.group(1), ins, addr
    shutil.rmtree(d, ignore_errors=True)


def code(op):
    o = op.split()[0]
    if o.startswith("v_mfma"):
        return "m" if "4x4x4" in o else "M"
    if o == "s_barrier":
        return "|"
    if o.startswith("ds_read"):
        return "r"
    if o.startswith("ds_write"):
        return "w"
    if "load_lds" in o:
        return "D"
    if o.startswith(("global_load", "buffer_load", "flat_load")):
        return "L"
    if o.startswith(("global_store", "buffer_store", "flat_store")):
        return "S"
    if o.startswith("s_waitcnt"):
        return "."
    if o == "s_nop":
        return "n"
    if o.startswith("v_"):
        return "v"
    return ""


def main():
    flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    demangle = shutil.which("c++filt")
    for name, ins, addr in kernels():
        shown = subprocess.run([demangle, name], capture_output=True, text=True).stdout.strip() if demangle else name
        shown = re.sub(r"ao::\(anonymous namespace\)::", "", shown).split("(")[0].replace("void ", "")
        if flt and not flt.search(shown):
            continue
        loops = []
        for i, op in enumerate(ins):
            mm = re.match(r"s_c?branch\w*\s+(-?\d+)", op)
            if mm:
                off = int(mm.group(1))
                off = off - 65536 if off >= 32768 else off
                tgt = addr[i] + 4 + 4 * off
                if tgt <= addr[i] and tgt in addr:
                    loops.append((addr.index(tgt), i))
        best = None
        for lo, hi in loops:
            n = sum(1 for o in ins[lo:hi + 1] if o.startswith("v_mfma"))
            if n and (best is None or n > best[0] or (n == best[0] and hi - lo < best[2] - best[1])):
                best = (n, lo, hi)
        if best is None:
            continue
        _, lo, hi = best
        seg = ins[lo:hi + 1]
        seq = "".join(code(o) for o in seg)
        drains = sum(1 for o in seg if re.match(r"s_waitcnt.*vmcnt\(0\)", o))
        print(f"{shown}\n    loop {len(seg)} instructions: {seq.count('M')} MFMA + {seq.count('m')} small, {seq.count('v')} VALU, {seq.count('r')} ds_read, "
              f"{seq.count('D')} LDS-DMA, {seq.count('L')} loads, {seq.count('|')} barriers, {seq.count('.')} waits ({drains} x vmcnt(0)), {seq.count('n')} s_nop\n    {seq}")


if __name__ == "__main__":
    main()
