#!/bin/bash
# bs = 128 A/B over (wpb knob, mode) pairs:  bash scripts/gpu_r3_c.sh "0,1,2" "840,850"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3c; mkdir -p $O; cd $R
for b in ${3:-128}; do timeout 600 python tools/int4_modes.py --batch $b --layout five --wpbs $1 --modes $2 --rounds 3 --steps 10 2>>$O/err.txt | tee -a $O/modes.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['batch'],'knob',d['wpb'],d['mode'],round(d['tokens_per_s_median']),d['event_us'],'rel %.1e'%d['max_rel_vs_first'])"; done
