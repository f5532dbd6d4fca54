#!/bin/bash
# MX / fp8 grouped kernel after the device-side slab enumeration: parity + timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py tests/test_moe_pad.py -m gpu -q --timeout 600 -k "grouped or mx or moe or mixtral" 2>&1 | tail -8
timeout 600 python bench.py --no-second-layout --configs mx --steps 10 --no-cpu-baseline > $O/bench_mx.json 2>$O/bench_mx.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s8/bench_mx.json').read().strip().splitlines()[-1])
c=d['configs']['mxfp8_mixtral_bs64']; print('mx config', c['value'], c['ms_per_step'], c['roofline']['achieved'], c['roofline']['frac'])
P
timeout 600 python tools/bench_8bit.py --which mx --iters 20 2>&1 | tee $O/bench8_mx_ragged.jsonl | cut -c1-200
timeout 600 python tools/bench_8bit.py --which mx --m 128 --iters 20 2>&1 | tee $O/bench8_mx_128.jsonl | cut -c1-200
timeout 600 python tools/bench_8bit.py --which mx --m 1024 --iters 20 2>&1 | tee $O/bench8_mx_1024.jsonl | cut -c1-200
