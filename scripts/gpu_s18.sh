#!/bin/bash
# round 6, session 18: the 128 x 256 tile as product dispatch -- int4 suites, then product (0) vs never (912), M = 512 ... 4096
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s18
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_int4_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_scale_gpu.py tests/test_int4_plain_hqq_gpu.py -m gpu -q --timeout 900 -x 2>&1 | tail -4 | tee $O/pytest.log
timeout 900 python tools/int4_w32_ab.py --ms 512,1024,2048,4096 --modes 0,912,0,912 2>&1 | tee $O/int4_w64_product_ab.jsonl | cut -c1-200
