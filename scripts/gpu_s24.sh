#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s24
mkdir -p $O
cd $R
timeout 600 python tools/mx_pair_diff.py 2>&1 | grep "^{\|Error\|error" | tee $O/mx_pair_diff.jsonl | cut -c1-900
