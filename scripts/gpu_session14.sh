#!/bin/bash
# int4 bs=128: columns per workgroup (NT n-tiles per wave) x split-K -- is the kernel bound by re-staging x per 64 columns?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s14
mkdir -p $O
cd $R
for b in 128 256; do
for wpb in 4 8; do
  modes="0,811,812,814,818"; [ $wpb = 4 ] && modes="0,811,812,814,818,821,822,824,828"
  timeout 900 python tools/int4_modes.py --batch $b --layout five --modes $modes --wpbs $wpb --rounds 2 --steps 10 2>$O/m.err | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('b=$b wpb', d['wpb'], 'mode', d['mode'], 'tok/s', round(d['tokens_per_s_best']), d['event_us'], 'rel', d['max_rel_vs_first'])
"
done; done 2>&1 | tee $O/nt_sweep.txt
