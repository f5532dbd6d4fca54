"""Build the gfx950 C-ABI library in-tree:  ao_amd/_C_mi355.so

    python -m ao_amd.build            # rebuild if sources are newer
    python -m ao_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but
travels with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "_C_mi355.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-fno-strict-aliasing",
    # the loop vectorizer "vectorises" wave-uniform scalar loops across SGPRs (8x unrolled unit
    # search, 480 SGPR spills in the streaming GEMV); nothing here wants it
    "-fno-vectorize",
    "-Wno-unused-result",
]


def sources():
    return sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")
    )


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")
    ] + [os.path.join(os.path.dirname(HERE), "include", "ao_mi355.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip under ao_amd/csrc into one shared library."""
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError(f"link failed:\n{out.stdout.decode()}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
