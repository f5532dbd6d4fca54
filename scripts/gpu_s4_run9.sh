#!/bin/bash
mkdir -p gpurun_out
for cfg in "97 0" "97 4" "97 8" "97 16" "96 0" "95 0" "96 8"; do
set -- $cfg
timeout 300 python bench.py --batch 1 --steps 30 --warmup 3 --no-cpu-baseline --no-second-layout --mode $1 --wpb $2 > gpurun_out/sw.json 2> gpurun_out/sw.err
python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
print("mode $1 wpb $2 tok/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],4), {k:round(v["us"],1) for k,v in d["roofline"]["per_shape"].items()})
PY
done
