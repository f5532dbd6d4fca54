#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
echo "== lab =="; timeout 600 tools/int4_lab 0:99 0:300 0:320 0:321 0:330 0:331 0:312 0:322 0:332 2>&1 | tee $O/lab11.txt
