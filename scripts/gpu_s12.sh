#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== mx parity =="
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py -m gpu -q --timeout 900 -k "mx" 2>&1 | tail -5
timeout 600 python bench.py --configs mx --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['mxfp8_mixtral_bs64']; print('multinomial tok/s %.0f frac %.3f ms %.3f' % (c['value'], c['roofline']['frac'], c['ms_per_step']), '| uniform16 tok/s %.0f frac %.3f ms %.3f' % (c['uniform16']['value'], c['uniform16']['roofline']['frac'], c['uniform16']['ms_per_step'])); print('one per product', c.get('one_launch_per_product')); print('two launch', c.get('two_launch'))"
