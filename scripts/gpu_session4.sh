#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_8bit_gpu.py -m gpu -q -k "race_screen" --timeout 600 2>&1 | tail -3
for m in 512 1024 2048 4096 8192; do
 for v in 0 2 8 32; do
  timeout 300 python tools/bench_8bit.py --which int8,fp8l --m $m --iters 10 --gemm-variant $v 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if d.get('M')==$m: print('M=$m v=$v', d['kernel'], d.get('shape'), d['N'], d['K'], round(d['us'],1), round(d.get('TOPs', d.get('TFLOPs',0))))
"
 done
done 2>&1 | tee $O/sweep.txt
