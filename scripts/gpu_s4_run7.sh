#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_int4_gpu.py tests/test_subclass_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --batch 128 --steps 20 --warmup 3 --no-cpu-baseline --no-second-layout > gpurun_out/bench_bs128.json 2> gpurun_out/bench_bs128.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_bs128.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:round(v["us"],1) for k,v in d["roofline"]["per_shape"].items()}, d["roofline"]["frac"])
PY
for m in 32 512 2048; do
timeout 300 python bench.py --batch $m --steps 5 --warmup 2 --layers 4 --no-cpu-baseline --no-second-layout > gpurun_out/sw.json 2> gpurun_out/sw.err
python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
print("M=$m", round(d["value"]), {k:round(v["us"],1) for k,v in d["roofline"]["per_shape"].items()}, d["roofline"]["frac"])
PY
done
