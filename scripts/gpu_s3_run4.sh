#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest int4 ==" ; timeout 600 python -m pytest tests/test_int4_gpu.py -m gpu -q -x --timeout 120 2>&1 | tail -4
echo "== lab ==" ; timeout 120 tools/int4_lab 0:100 0:0 8:0 2>&1 | tee $O/lab_s3_4.txt
