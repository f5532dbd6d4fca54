#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s23
mkdir -p $O
cd $R
timeout 600 python tools/mx_pair_diff.py 2>&1 | grep "^{\|Error\|error" | tee $O/mx_pair_diff.jsonl
timeout 1500 python tools/midm_sweep.py --kinds fp8,int8 --ms 128,256,512,768,1024,2048 --forms default 2>/dev/null | grep "^{" > $O/midm_final.jsonl; wc -l $O/midm_final.jsonl
