#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== lab =="; timeout 600 tools/int4_lab 0:99 0:300 0:301 0:304 0:305 0:309 4:301 8:301 16:301 0:311 0:312 0:313 2>&1 | tee $O/lab10.txt
