#!/bin/bash
# round 6, session 17: 64-column wave tiles of the batched int4 kernel (modes 95S / 96S) -- parity, then the A/B at M = 128 ... 2048
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s17
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_int4_gpu.py -m gpu -q --timeout 600 -x -k "register_b_kernel" 2>&1 | tail -5 | tee $O/pytest.log
timeout 1200 python tools/int4_w32_ab.py --ms 128,256,512,2048 --modes 0,950,952,954,960,962,964,0 2>&1 | tee $O/int4_w64_ab.jsonl | cut -c1-250
