#!/bin/bash
# round 6, session 16: 8-row tiles of the decode kernel -- parity in every form, then the A/B (291 never / 292 always), cold weights
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s16
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_8bit_gpu.py -m gpu -q --timeout 600 -x -k "decode_kernel_every_form or fused_dynamic" 2>&1 | tail -5 | tee $O/pytest.log
timeout 900 python tools/bench_dec8.py --kinds fp8,int8 --ms 1,4 --variants 291,292,291,292,0 2>&1 | tee $O/dec8_rows8_ab.jsonl | cut -c1-300
