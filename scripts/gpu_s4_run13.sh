#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_subclass_gpu.py -x -q -m gpu 2>&1 | tail -3
for m in 1 16 128; do
timeout 600 python tools/bench_8bit.py --m $m --iters 10 --which int8 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_8bit_int8.jsonl
python - <<PY
import json
for l in open("gpurun_out/bench_8bit_int8.jsonl"):
    l=l.strip()
    if not l.startswith("{"): print(l[:200]); continue
    d=json.loads(l)
    b = d["N"]*d["K"]
    print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ("shape","M","us","TOPs","us_with_act_quant")}, "weights GB/s", round(b/d["us"]/1e3))
PY
done
