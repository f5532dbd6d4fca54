#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_int4_gpu.py tests/test_fuzz_gpu.py -m gpu -q --timeout 600 2>&1 | tail -5
for b in 32 64 128 256 512; do
timeout 600 python tools/int4_modes.py --batch $b --layout five --modes 0 --wpbs 0 --rounds 2 --steps 10 2>$O/m.err | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('b=$b tok/s',round(d['tokens_per_s_best']), d['event_us'])
"
done 2>&1 | tee $O/bs_sweep.txt
