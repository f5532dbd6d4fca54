"""Disassembly guards (no GPU: hipcc cross-compiles): properties of the generated code that no numerics test can see.

gemm8_p8_kernel is built on a phase structure -- per 128-byte K tile four phases, each `[LDS reads / LDS-DMA issue | barrier | MFMAs |
barrier]`, the two wave rows one barrier apart so that one multiplies while the other loads.  In round 3 the instruction selector was
found to have moved all of a K tile's fp8 MFMAs (builtins: pure value computations) to the end of the loop body: same results, 21 %
slower.  The fp8 MFMAs are volatile asm since; this test keeps both flavours honest."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _loop_ops(asm: str, mangled_prefix: str):
    m = re.search(r"\n(" + re.escape(mangled_prefix) + r"[^\n:]*):[^\n]*\n", asm)
    assert m, f"{mangled_prefix} not found in the disassembly"
    body = asm[m.start(): asm.find(".Lfunc_end", m.start())].split("\n")
    labels = {mm.group(1): i for i, l in enumerate(body) if (mm := re.match(r"(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    def ops(lo, hi):
        seq = []
        for l in body[lo:hi]:
            op = l.split(";")[0].split()
            if not op:
                continue
            op = op[0]
            seq.append("M" if op.startswith("v_mfma") else "|" if op == "s_barrier" else "r" if op.startswith("ds_read") else
                       "D" if op.startswith("global_load_lds") else "")
        return "".join(seq)

    # the K loop is the innermost loop that holds the MFMAs: the shortest backward branch with the most of them
    cands = [ops(lo, hi) for lo, hi in loops]
    most = max(c.count("M") for c in cands)
    return min((c for c in cands if c.count("M") == most), key=len)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_gemm8_p8_mfmas_stay_inside_their_phases(tmp_path):
    src = os.path.join(ROOT, "ao_amd", "csrc", "gemm8_p8_kernels.hip")
    out = tmp_path / "p8.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-fno-vectorize",
           "-Wno-unused-result", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "ao_amd", "csrc"), src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    for epi, per_phase in ((0, 16), (1, 16), (2, 8), (3, 8)):  # int8: 16 x 16x16x64 per phase; fp8: 8 x 16x16x128
        seq = _loop_ops(asm, f"_ZN2ao12_GLOBAL__N_115gemm8_p8_kernelILi{epi}EEEvNS0_6P8ArgsE")
        assert seq.count("M") == 4 * per_phase, (epi, seq)
        groups = [len(g) for g in re.findall(r"M+", seq)]
        assert groups == [per_phase] * 4, f"EPI {epi}: the K tile's MFMAs are not four phases of {per_phase}: {seq}"
        # every MFMA group sits between two barriers (the seams), the fragment reads come in two groups, the DMAs in four pairs
        assert len(re.findall(r"\|M+\|", seq)) + (1 if seq.startswith("M") or seq.endswith("M") else 0) >= 3, seq
        assert seq.count("D") >= 6, (epi, seq)  # (8 per K tile; a rotated loop leaves the last pair outside the backward branch's span)
