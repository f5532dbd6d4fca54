#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest 8bit ==" ; timeout 600 python -m pytest tests/test_8bit_gpu.py -m gpu -q --timeout 200 2>&1 | tail -6
for v in 16; do
echo "== 8bit bench M=8192 variant $v ==" ; timeout 600 python tools/bench_8bit.py --which int8,fp8 --m 8192 --gemm-variant $v 2>&1 | grep -v amdgpu.ids | tee $O/bench_8bit_m8192_v$v.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    if d['M'] < 1000: continue
    print(d['kernel'], d.get('shape'), 'M', d['M'], 'us', round(d['us'],1), 'frac_mfma', round(d.get('frac_mfma',0),3), 'T', round(d.get('TOPs', d.get('TFLOPs',0)),0))"
done
