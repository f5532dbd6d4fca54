#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/abl; mkdir -p $O; cd $R
{
for a in 0 1 3 4 8 12 7 11 15; do timeout 120 python tools/fp8_rb_trace.py 128 1280 8192 101 "1=32,5=$a"; done
for a in 0 1 3 4 8 12 15; do timeout 120 python tools/fp8_rb_trace.py 128 7168 8192 101 "5=$a"; done
} 2>&1 | grep -v amdgpu.ids > $O/trace.txt
grep -E "^M=|per workgroup|mean ticks" $O/trace.txt
