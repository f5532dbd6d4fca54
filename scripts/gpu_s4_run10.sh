#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/bench_8bit.py --m 2048 --iters 10 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_8bit_m2048.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/bench_8bit_m2048.jsonl"):
    l=l.strip()
    if not l.startswith("{"): print(l[:200]); continue
    d=json.loads(l)
    print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k not in ("note",)})
PY
