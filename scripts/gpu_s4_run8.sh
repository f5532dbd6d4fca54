#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_moe_pad.py -x -q -m gpu 2>&1 | tail -5
timeout 120 python tools/bench_moe_pad.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_moe_pad.json
