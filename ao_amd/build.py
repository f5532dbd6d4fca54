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
OPS_LIB = os.path.join(HERE, "_C_mi355_ops.so")  # C++ dispatcher registrations (csrc_torch/binding.cpp), links against LIB
OPS_SRC = os.path.join(HERE, "csrc_torch", "binding.cpp")
OPS_SRC_STABLE = os.path.join(HERE, "csrc_torch", "binding_stable.cpp")  # torchao:: ops through the stable ABI (no ATen / c10 headers)
CXX = os.environ.get("CXX", "g++")
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


def build_ops(force=False, verbose=False):
    """Compile the host-only dispatcher binding against the installed torch (no device code: plain g++)."""
    if not force and os.path.exists(OPS_LIB) and os.path.getmtime(OPS_LIB) >= max(os.path.getmtime(OPS_SRC), os.path.getmtime(OPS_SRC_STABLE), os.path.getmtime(LIB)):
        return OPS_LIB
    import torch

    t = os.path.dirname(torch.__file__)
    cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{t}/include", f"-I{t}/include/torch/csrc/api/include", "-I/opt/rocm/include", OPS_SRC, OPS_SRC_STABLE, "-o", OPS_LIB,
           f"-L{t}/lib", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip", f"-L{HERE}", "-l:_C_mi355.so",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{t}/lib"]
    if verbose:
        print(" ".join(cmd))
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError(f"building {OPS_LIB} failed:\n{out.stdout.decode()}")
    return OPS_LIB


def _prune_objects(bdir, keep):
    """Objects of sources that no longer exist must not linger (they never ship -- ao_amd/build/ is gpurun-ignored -- but a
    stale object is a stale object)."""
    for f in os.listdir(bdir):
        p = os.path.join(bdir, f)
        if f.endswith(".o") and p not in keep:
            os.remove(p)


def build(force=False, verbose=False, lab=False):
    """Compile every .hip under ao_amd/csrc into one shared library, then the dispatcher binding.  lab=True: the laboratory
    library instead (tools/bin/_C_mi355_lab.so, -DAO_LAB: the product plus the wrong-result ablation builds of the profiling
    tools; select it with AO_MI355_LIB=tools/bin/_C_mi355_lab.so) -- never what ao_amd loads by default."""
    if lab:
        return _build_lab(verbose)
    if not force and not _stale():
        build_ops(force=False, verbose=verbose)
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
    _prune_objects(bdir, set(objs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError(f"link failed:\n{out.stdout.decode()}")
    build_ops(force=True, verbose=verbose)
    return LIB


def _build_lab(verbose=False):
    out = os.path.join(os.path.dirname(HERE), "tools", "bin", "_C_mi355_lab.so")
    bdir = os.path.join(HERE, "build", "lab")
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-DAO_LAB=1", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        o, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{o.decode()}")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv))
