#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest 8bit+subclass ==" ; timeout 600 python -m pytest tests/test_subclass_gpu.py tests/test_8bit_gpu.py -m gpu -q --timeout 200 2>&1 | tail -8
