#!/bin/bash
# round 3: stream-K MX decode kernel -- parity tests, then config 5 with the product rule (0), always stream-K (119), never (113)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py -x -q -m gpu -k "mx or MX or grouped" > gpurun_out/mx_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/mx_tests.log
tail -5 gpurun_out/mx_tests.log
grep -q "rc=0" gpurun_out/mx_tests.log || exit 1
for v in 0 119 113 0; do
  timeout 600 python bench.py --configs mx --no-tp-graph --no-stack-baseline --no-subclass-graph --no-cpu-baseline --steps 20 --warmup 5 --gemm-variant $v > gpurun_out/bench_mx_v$v.json 2> gpurun_out/bench_mx_v$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/bench_mx_v{v}.json').read().strip().splitlines()[-1])
    c = d.get('configs', {}).get('mxfp8_mixtral_bs64', {})
    print('variant', v, 'config 5:', c.get('value'), 'tok/s', c.get('ms_per_step'), 'ms/step frac', c.get('roofline', {}).get('frac'))
except Exception as e:
    print('variant', v, 'failed', e); print(open(f'gpurun_out/bench_mx_v{v}.err').read()[-1500:])
PY
done
for m in 128 256; do
for v in 0 113; do
  timeout 300 python tools/bench_8bit.py --which mx --iters 20 --gemm-variant $v --m $m > gpurun_out/mx_m${m}_v$v.jsonl 2>&1
  echo "== rows $m variant $v (all 8 experts)"; grep -o '"shape": "[^"]*"[^}]*"us": [0-9.]*' gpurun_out/mx_m${m}_v$v.jsonl | sed 's/"E".*"us"/us/'
done; done
