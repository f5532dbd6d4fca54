#!/bin/bash
# int4 ring-wait fix check + per-step trace of the grouped MX kernel with / without the producer wave under different loads
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s10
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_int4_gpu.py -m gpu -q --timeout 600 2>&1 | tail -4
for lib in "" tools/bin/_C_mi355_noprod.so; do
  echo "== library: ${lib:-product (producer wave)}"
  for sizes in 32,0,0,0,0,0,0,0 32,32,0,0,0,0,0,0 32,0,0,0,32,64,0,0 32,0,32,16,16,0,32,0 16,16,16,16,16,16,16,16 32,32,32,32,32,32,32,32; do
    timeout 300 python tools/mx_rb_trace.py 14336 4096 $sizes $lib 2>/dev/null
    timeout 300 python tools/mx_rb_trace.py 4096 14336 $sizes $lib 2>/dev/null
  done
done 2>&1 | tee $O/mx_trace.txt
