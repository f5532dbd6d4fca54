#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_int4_gpu.py -x -q -m gpu -k "register_b" 2>&1 | tail -3
timeout 120 python tools/rb_trace.py 4096 4096 1 2>&1 | grep -v amdgpu.ids | tail -3
for cfg in "8 601" "8 608" "8 604" "4 604" "8 668"; do
  set -- $cfg
  timeout 120 python bench.py --batch 128 --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout --wpb $1 --mode $2 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
    print("wpb $1 mode $2", round(d["value"]), {k:round(v["us"],1) for k,v in d["roofline"]["per_shape"].items()})
except Exception as e:
    print("wpb $1 mode $2 FAILED", e); print(open("gpurun_out/sw.err").read()[-600:])
PY
done
