#!/bin/bash
# round 3: stream-K MX kernel, activation pieces fetched only where the group has rows -- parity tests, traces, config 5
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py -x -q -m gpu -k "mx or MX or grouped" > gpurun_out/mx_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/mx_tests.log
tail -5 gpurun_out/mx_tests.log
grep -q "rc=0" gpurun_out/mx_tests.log || exit 1
for v in 119 118; do
  echo "== gemm8 variant $v"
  for sh in "14336 4096" "4096 14336"; do
    for sz in 32,0,0,0,32,64,0,0 16,16,16,16,16,16,16,16; do
      timeout 120 python tools/mx_rb_trace.py $sh $sz - $v 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|per step {d" | sed 's/; launch span.*//'
    done
  done
done > gpurun_out/mx_rb_trace9.txt
grep "^==\|^N=" gpurun_out/mx_rb_trace9.txt | cut -c1-150
for v in 0 118; do
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
