#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
for b in 128 256; do
timeout 600 python tools/int4_modes.py --batch $b --layout five --modes 0,661,662,664,668 --wpbs 4 --rounds 2 --steps 10 2>$O/m.err | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('b=$b wpb',d['wpb'],'mode',d['mode'],'tok/s',round(d['tokens_per_s_best']), d['event_us'], 'rel', d['max_rel_vs_first'])
"
done 2>&1 | tee $O/bs128_nt2.txt
tail -3 $O/m.err
