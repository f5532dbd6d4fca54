#!/bin/bash
# round 6, session 21: time-boxed random shape sweep through the default dispatch (tools/fuzz_long.py), then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s21
mkdir -p $O
cd $R
for s in 1 2 3; do
  timeout 400 python tools/fuzz_long.py --seconds 240 --seed $s 2>&1 | grep "^{" | tee -a $O/fuzz_long.jsonl | cut -c1-300
done
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | grep -a "passed\|failed\|rror" | tail -5 | tee $O/pytest_all.log
