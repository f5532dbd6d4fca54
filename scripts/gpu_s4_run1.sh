#!/bin/bash
# split-K tiled kernel: parity + bs=128 bench
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_int4_gpu.py -x -q -m gpu 2>&1 | tail -5
for mode in 0 501; do
  echo "== bs128 mode $mode"
  timeout 300 python bench.py --batch 128 --steps 20 --warmup 3 --no-cpu-baseline --no-second-layout --mode $mode > gpurun_out/bench_bs128_m$mode.json 2> gpurun_out/bench_bs128_m$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_bs128_m$mode.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:round(v["us"],1) for k,v in d["roofline"]["per_shape"].items()})
PY
done
