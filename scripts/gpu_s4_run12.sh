#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_subclass_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/bench_8bit.py --m 2048 --iters 10 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_8bit_r01.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/bench_8bit_r01.jsonl"):
    l=l.strip()
    if not l.startswith("{"): print(l[:200]); continue
    d=json.loads(l)
    if d.get("kernel")=="fp8_scaled_mm": print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ("shape","M","us","GBps","TFLOPs")})
PY
