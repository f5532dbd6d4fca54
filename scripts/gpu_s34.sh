#!/bin/bash
# round 6: (slab height, tile width, K parts) grid of the weight-streaming 8-bit kernel at 80 <= M <= 512, cold weights -- the data rb8_plan is fitted to
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s34
mkdir -p $O
cd $R
FORMS="default"
for bm in 128 64; do for bn in 32 64 128; do for s in 1 2 3 4 6 8; do FORMS="$FORMS,bm$bm+bn$bn+s$s"; done; done; done
timeout 1200 python tools/midm_sweep.py --ms 80,96,128,160,192,256,320,384,512 --kinds fp8 --families 70b,8b --no-core --forms $FORMS 2>&1 | grep "^{" > $O/grid_fp8.jsonl
wc -l $O/grid_fp8.jsonl
timeout 900 python tools/midm_sweep.py --ms 128,256,512 --kinds int8 --families 70b,8b --no-core --forms $FORMS 2>&1 | grep "^{" > $O/grid_int8.jsonl
wc -l $O/grid_int8.jsonl
