#!/bin/bash
# round 6, verdict item 5 measured: 64-row slabs at M = 128 / 256 (M cut instead of K) and the barrier-halving timing probe
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s33
mkdir -p $O
cd $R
FORMS="default,bm64,bm64+bn32,bm64+bn64,bm64+bn128,bm64+bn32+s1,bm64+bn32+s2,bm64+bn64+s1,bm64+bn64+s2,bm64+bn64+s3,bm64+bn128+s2,bm64+bn128+s3,bm64+bn128+s4,bn32+s1"
timeout 900 python tools/midm_sweep.py --ms 128,256 --kinds fp8,int8 --families 70b,8b --check --forms $FORMS 2>&1 | grep "^{" > $O/midm_bm64.jsonl
wc -l $O/midm_bm64.jsonl
{
for shape in "128 1280 8192" "128 8192 1024" "128 7168 8192" "128 8192 3584"; do
  for abl in 0 16; do
    timeout 120 python tools/fp8_rb_trace.py $shape 101 5=$abl 2>&1 | grep -v "^wg"
  done
done
} > $O/rb_barrier_probe.txt 2>&1
tail -30 $O/rb_barrier_probe.txt
