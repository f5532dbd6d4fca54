#!/bin/bash
# round 3: timing probes of the MX stream kernel (thousands digit: 2 = no activation DMAs, 5 = activation DMAs read one line; wrong numbers)
mkdir -p gpurun_out
for v in 118 2118 5118 0 2000 114 2114; do
  echo "== gemm8 variant $v"
  for sh in "14336 4096"; do
    for sz in 32,0,0,0,32,64,0,0 16,16,16,16,16,16,16,16; do
      timeout 120 python tools/mx_rb_trace.py $sh $sz - $v 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|per step {d" | sed 's/; launch span.*//'
    done
  done
done > gpurun_out/mx_rb_trace5.txt
cat gpurun_out/mx_rb_trace5.txt
