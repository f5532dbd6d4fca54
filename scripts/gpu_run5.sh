#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest gpu ==" ; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tee $O/pytest_gpu.log | tail -60
