#!/bin/bash
# round 6, session 14: the persistent 256 x 256 GEMM -- parity against the per-tile kernel, then the A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s14
mkdir -p $O
cd $R
echo "== parity =="
timeout 900 python -m pytest tests/test_8bit_gpu.py -m gpu -q --timeout 600 -x -k "persistent_form or phase_interleaved" 2>&1 | tail -15 | tee $O/pytest.log
echo "== A/B =="
timeout 900 python tools/p8_persist_ab.py int8,fp8 2>&1 | tee $O/p8_persist_ab.jsonl | cut -c1-400
