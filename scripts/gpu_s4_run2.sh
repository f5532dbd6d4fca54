#!/bin/bash
# sweep tile width (wpb = n-tiles per workgroup) x split-K parts at bs=128
mkdir -p gpurun_out
for cfg in "8 508" "8 504" "8 502" "4 508" "4 504" "4 502" "4 501" "2 504" "2 502" "2 501"; do
  set -- $cfg
  timeout 120 python bench.py --batch 128 --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout --wpb $1 --mode $2 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
print("tnw $1 mode $2", round(d["value"]), {k:round(v["us"],1) for k,v in d["roofline"]["per_shape"].items()})
PY
done
