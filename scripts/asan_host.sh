#!/bin/bash
# Host-side sanitizers (the GPU pool has no device sanitizers): every .hip compiled with -Xarch_host -fsanitize=address,undefined, the
# library loaded through AO_MI355_LIB under the clang ASan runtime, and the host-only tests + a grid of dispatch queries (zero, odd and
# 2^31 - 1 sized shapes) run against it.  No GPU needed.   bash scripts/asan_host.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
B=${TMPDIR:-/tmp}/ao_asan
mkdir -p $B && cd $B
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-strict-aliasing -fno-vectorize -Wno-unused-result -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer"
for f in $R/ao_amd/csrc/*.hip; do /opt/rocm/bin/hipcc $F -c $f -o $(basename $f).o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $B/_C_mi355_asan.so *.hip.o
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $R
export AO_MI355_LIB=$B/_C_mi355_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
python -m pytest tests/test_host_dispatch.py tests/test_abi.py tests/test_host_api.py tests/test_streamk_partition.py -q -m "not gpu" 2>&1 | tail -3
python - <<'PY'
import ctypes, itertools
from ao_amd import _lib
lib = _lib.lib()
lib.ao_gemm8_kernel_name.restype = lib.ao_int4_mm_kernel_name.restype = ctypes.c_char_p
I = ctypes.c_int64
names = set()
for M, N, K in itertools.product([0, 1, 3, 16, 17, 64, 65, 128, 129, 256, 1000, 2048, 16384, 262144, (1 << 31) - 1], [16, 48, 1280, 4096, 14336, 1 << 20], [128, 1024, 4096, 14336, 1 << 20]):
    for i8 in (0, 1):
        names.add(lib.ao_gemm8_kernel_name(i8, I(M), I(N), I(K)))
        bn, sp = ctypes.c_int(), ctypes.c_int()
        lib.ao_gemm8_plan(i8, I(M), I(N), I(K), ctypes.byref(bn), ctypes.byref(sp))
        lib.ao_gemm8_plan_rows(i8, I(M), I(N), I(K), ctypes.byref(bn))
    for g in (32, 128):
        names.add(lib.ao_int4_mm_kernel_name(I(M), I(N), I(K), g))
    for E in (1, 8, 64, 65):
        lib.ao_mxfp8_grouped_mm_dyn_fits(I(M), I(N), I(K), I(E))
        lib.ao_mxfp8_grouped_mm_pair_fits(I(M), I(N), I(K), I(E))
for rows, groups in itertools.product([0, 1, 127, 128, (1 << 31) - 1], [0, 1, 32, 1 << 20]):
    lib.ao_mx_blocked_rows(I(rows), I(groups))
    lib.ao_mx_block_rearrange_2d_m_groups(None, None, None, I(rows), I(16), I(groups), None)  # null pointers: refused before any launch
    lib.ao_mx_to_blocked(None, None, I(rows), I(0), None)
print("dispatch queries under ASan / UBSan:", len(names), "kernel names, no report")
PY
