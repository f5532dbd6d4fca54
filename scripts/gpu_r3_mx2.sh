#!/bin/bash
mkdir -p gpurun_out
for v in 118 7118 119 7119; do
  echo "== gemm8 variant $v"
  for sh in "14336 4096" "4096 14336"; do
    for sz in 32,0,0,0,32,64,0,0 16,16,16,16,16,16,16,16; do
      timeout 120 python tools/mx_rb_trace.py $sh $sz - $v 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|per step {d" | sed 's/; launch span.*//'
    done
  done
done > gpurun_out/mx_rb_trace7.txt
cat gpurun_out/mx_rb_trace7.txt
