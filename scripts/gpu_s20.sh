#!/bin/bash
# round 6, session 20: p8p epilogue by column pairs (fewer live scale registers) -- 8-bit suites, persistent A/B, then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s20
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py -m gpu -q --timeout 900 -x -k "gemm8 or p8 or persistent or int8 or fp8" 2>&1 | tail -4 | tee $O/pytest_8bit.log
timeout 900 python tools/p8_persist_ab.py 2>&1 | tee $O/p8_persist_ab.jsonl | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -5 | tee $O/pytest_all.log
