#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== mx debug =="
timeout 300 python tools/mx_debug.py 2>&1 | grep -c ": ok"; timeout 300 python tools/mx_debug.py 2>&1 | grep "bad elements\|Error\|error" | head
echo "== mx parity =="
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py -m gpu -q --timeout 900 -k "mx" 2>&1 | tail -12
echo "== bench mx: 8-wave, 16-wave, 16-wave + fused cast =="
for v in 119 0; do
  timeout 600 python bench.py --configs mx --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph --steps 3 --warmup 1 --gemm-variant $v 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['mxfp8_mixtral_bs64']; print('variant $v', 'multinomial tok/s %.0f frac %.3f ms %.3f' % (c['value'], c['roofline']['frac'], c['ms_per_step']), '| uniform16 tok/s %.0f frac %.3f ms %.3f' % (c['uniform16']['value'], c['uniform16']['roofline']['frac'], c['uniform16']['ms_per_step']), {k: v for k, v in c.items() if k.startswith('fused')})"
done
