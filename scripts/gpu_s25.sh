#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s25
mkdir -p $O
cd $R
timeout 900 python tools/stress_mx_pair.py --iters 1500 --seed 1 --noise 1 2>&1 | grep "^{\|Error\|error" | tee $O/stress_noise.jsonl | cut -c1-1200 | tail -12
timeout 900 python tools/stress_mx_pair.py --iters 1500 --seed 2 --noise 0 2>&1 | grep "^{\|Error\|error" | tee $O/stress_quiet.jsonl | cut -c1-1200 | tail -12
