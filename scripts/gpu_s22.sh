#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s22
mkdir -p $O
cd $R
for s in 11 12; do
  timeout 400 python tools/fuzz_long.py --seconds 200 --seed $s --kinds dyn,mxdyn 2>&1 | grep "^{" | tee -a $O/fuzz_long_dyn.jsonl | cut -c1-300
done
timeout 400 python tools/fuzz_long.py --seconds 200 --seed 13 2>&1 | grep "^{" | tee -a $O/fuzz_long_dyn.jsonl | cut -c1-300
