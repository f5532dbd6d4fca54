#!/bin/bash
# quick GPU check of a subset: bash scripts/gpu_quick.sh "<pytest args>" 
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/quick
mkdir -p $O
cd $R
timeout 1500 python -m pytest $1 -m gpu -q --timeout 900 2>&1 | tee $O/pytest.log | tail -40
