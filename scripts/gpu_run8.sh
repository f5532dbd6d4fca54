#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest gpu =="; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -5
echo "== lab =="; timeout 600 tools/int4_lab 0:99 0:0 4:0 8:0 16:0 0:104 0:106 0:110 0:101 0:102 8:101 2>&1 | tee $O/lab8.txt
