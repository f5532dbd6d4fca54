#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== clock probe =="; timeout 120 tools/clock_probe 2>&1 | tee $O/clock_probe.txt
echo "== pytest gpu =="; timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5
