#!/bin/bash
# round 3: 4-step scale fetches in the per-tile MX kernel too -- parity tests, then prefill-size groups (128 / 256 rows per expert) and decode groups by variant
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py -x -q -m gpu -k "mx or MX or grouped" > gpurun_out/mx_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/mx_tests.log
tail -5 gpurun_out/mx_tests.log
grep -q "rc=0" gpurun_out/mx_tests.log || exit 1
for m in 1024 2048 128; do
for v in 0 129 113; do
  timeout 300 python tools/bench_8bit.py --which mx --iters 20 --gemm-variant $v --m $m > gpurun_out/mx_m${m}_v$v.jsonl 2>&1
  echo "== rows $m (all 8 experts) variant $v"; grep -o '"shape": "[^"]*"[^}]*"us": [0-9.]*' gpurun_out/mx_m${m}_v$v.jsonl | sed 's/"E".*"us"/us/'
done; done
