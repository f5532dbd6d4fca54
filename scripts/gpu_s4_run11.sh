#!/bin/bash
mkdir -p gpurun_out
for v in 101 102; do
echo "variant $v"
timeout 600 python tools/bench_8bit.py --m 128 --iters 10 --which fp8 --gemm-variant $v 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_8bit_fp8.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/bench_8bit_fp8.jsonl"):
    l=l.strip()
    if not l.startswith("{"): print(l[:200]); continue
    d=json.loads(l)
    if d["M"]==128: print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ("shape","M","us","GBps")})
PY
done
