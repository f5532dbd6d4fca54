#!/bin/bash
# Session-3 GPU pass 2: streaming GEMV parity + lab timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest int4 ==" ; timeout 900 python -m pytest tests/test_int4_gpu.py -m gpu -q -x --timeout 300 2>&1 | tee $O/pytest_int4.log | tail -15
echo "== lab ==" ; timeout 600 tools/int4_lab 0:100 0:0 8:0 0:401 0:402 2>&1 | tee $O/lab_s3_2.txt
echo "== bench ==" ; timeout 600 python bench.py --no-cpu-baseline 2>$O/bench.err | tee $O/bench2.json | cut -c1-2500
tail -3 $O/bench.err
