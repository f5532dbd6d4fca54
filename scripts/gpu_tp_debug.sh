#!/bin/bash
# why does the TP step's hipGraph capture fail?  HIP runtime error log of the world-1 TP run, twice
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  AO_BENCH_DEBUG=1 timeout 300 python bench.py --force-tp --configs tp --no-second-layout --no-cpu-baseline --steps 5 > gpurun_out/tp_dbg_$i.json 2> gpurun_out/tp_dbg_$i.err
  echo "== run $i rc=$?"
  grep -n "hipError\|capture\|Capture\|\[bench" gpurun_out/tp_dbg_$i.err | grep -v "^.*frame #" | tail -2 | cut -c1-200; grep -c "hipGraph replay" gpurun_out/tp_dbg_$i.json
done
