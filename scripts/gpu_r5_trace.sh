#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trace; mkdir -p $O; cd $R
{
for t in "" "3=1" "1=32" "1=32,3=1" "1=64,2=8" "1=32,2=4"; do timeout 120 python tools/fp8_rb_trace.py 128 1280 8192 101 "$t"; done
for t in "" "3=1" "1=32" "1=32,3=1"; do timeout 120 python tools/fp8_rb_trace.py 128 8192 1024 101 "$t"; done
for t in "" "1=64"; do timeout 120 python tools/fp8_rb_trace.py 128 8192 3584 101 "$t"; done
for t in "" ; do timeout 120 python tools/fp8_rb_trace.py 1024 8192 1024 101 "$t"; done
} 2>&1 | grep -v amdgpu.ids | tee $O/trace.txt
