#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/bench_8bit.py --which fp8,quant --m 2048 --iters 20 2>&1 | grep -v amdgpu > gpurun_out/bench_8bit_m2048.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/bench_8bit_m2048.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    if d['kernel']=='fp8_scaled_mm' and d['M'] in (128,2048): print(d['shape'].ljust(14), d['M'], d['N'], d['K'], round(d['us'],1),'us', round(d['TFLOPs']),'TF', round(d['GBps']),'GB/s')
    if 'quantize' in d['kernel']: print(d['kernel'], d['M'], d['K'], round(d['us'],1), 'us', round(d['GBps']), 'GB/s')
PY
