#!/bin/bash
# register-B batched kernel: parity, then timing vs the LDS-B tiled kernel at bs=128
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_int4_gpu.py -x -q -m gpu -k "register_b or tiled" 2>&1 | tail -8
for cfg in "0 0" "8 601" "8 602" "8 604" "8 608" "4 601" "4 602" "4 604" "4 608"; do
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
