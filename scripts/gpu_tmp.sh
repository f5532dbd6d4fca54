#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_int4_gpu.py tests/test_subclass_gpu.py -x -q -m gpu 2>&1 | tail -3
for mode in 94 0 94 0; do
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-second-layout --mode $mode > gpurun_out/sw.json 2> gpurun_out/sw.err
python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
print("mode $mode tok/s", round(d["value"],1), {k:round(v["us"],1) for k,v in d["roofline"]["per_shape"].items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
PY
done
