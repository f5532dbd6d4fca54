#!/bin/bash
# bs = 1 A/B over modes, both layouts:  bash scripts/gpu_r3_e.sh "0,230"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3e; mkdir -p $O; cd $R
for lay in five merged; do timeout 600 python tools/int4_modes.py --batch 1 --layout $lay --modes $1 --rounds 5 --steps 20 2>>$O/err.txt | tee -a $O/modes.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['layout'],d['mode'],round(d['tokens_per_s_median'],1),round(d['tokens_per_s_best'],1),d['event_us'],'rel %.1e'%d['max_rel_vs_first'])"; done
