#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s8
mkdir -p $O
cd $R
echo "== mx parity =="
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py -m gpu -q --timeout 900 -k "mx" 2>&1 | tail -6
timeout 300 python tools/mx_debug.py 2>&1 | grep -c ": ok"; timeout 300 python tools/mx_debug.py 2>&1 | grep "bad elements\|Error\|error" | head -5
for w in "14336 4096 32,0,0,0,32,64,0,0" "4096 14336 32,0,0,0,32,64,0,0"; do
  timeout 300 python tools/mx_rb_trace.py $w 2>&1 | grep -v "ret = \|RuntimeWarning\|per step {d" | tail -6 | cut -c1-300
done
timeout 600 python bench.py --configs mx --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['mxfp8_mixtral_bs64']; print('multinomial tok/s %.0f frac %.3f ms %.3f' % (c['value'], c['roofline']['frac'], c['ms_per_step']), '| uniform16 tok/s %.0f frac %.3f ms %.3f' % (c['uniform16']['value'], c['uniform16']['roofline']['frac'], c['uniform16']['ms_per_step']), c.get('two_launch'))"
