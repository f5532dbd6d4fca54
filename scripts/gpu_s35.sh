#!/bin/bash
# round 6: the refitted rb8_plan (64-row slabs above 64 rows) on the box: small-M grid, the sweep next to hipBLASLt, the 8-bit suites, a fuzz pass
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s35
mkdir -p $O
cd $R
FORMS="default"
for bn in 32 64 128; do for s in 1 2 3 4 6 8; do FORMS="$FORMS,bm64+bn$bn+s$s"; done; done
timeout 600 python tools/midm_sweep.py --ms 40,48,64 --kinds fp8,int8 --families 70b,8b --no-core --forms $FORMS 2>&1 | grep "^{" > $O/grid_small.jsonl
wc -l $O/grid_small.jsonl
timeout 900 python tools/midm_sweep.py --ms 80,96,128,160,192,256,384,512,768,1024 --kinds fp8,int8 --families 70b,8b --check --forms default,bm128,bm64 2>&1 | grep "^{" > $O/midm_default.jsonl
wc -l $O/midm_default.jsonl
timeout 1500 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_variants_gpu.py tests/test_fuzz_gpu.py tests/test_subclass_gpu.py -m gpu -q --timeout 900 -x 2>&1 | tail -5 | tee $O/pytest.log
timeout 400 python tools/fuzz_long.py --seconds 300 --seed 71 2>&1 | grep "^{" | tee $O/fuzz.jsonl | cut -c1-400 | tail -4
