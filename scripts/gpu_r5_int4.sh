#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/int4; mkdir -p $O; cd $R
echo "== tests =="; timeout 900 python -m pytest tests/test_int4_gpu.py -q -x --timeout 600 -k "register_b_kernel" 2>&1 | tail -5 | tee $O/tests.log
echo "== A/B =="; timeout 600 python tools/int4_w32_ab.py --ms 128,256 > $O/ab.jsonl 2>$O/ab.err; tail -2 $O/ab.err; cat $O/ab.jsonl | cut -c1-200
