#!/bin/bash
# round-3 call B: K-half-wave kernel A/B at bs = 64 / 128 / 256
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
MODES=${1:-0,840,842,844,848,850}
for b in 128 64 256; do timeout 600 python tools/int4_modes.py --batch $b --layout five --modes $MODES --rounds 3 --steps 10 2>>$O/err.txt | tee -a $O/modes.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['batch'],d['mode'],round(d['tokens_per_s_median']),d['event_us'],'rel %.1e'%d['max_rel_vs_first'])"; done
tail -3 $O/err.txt
