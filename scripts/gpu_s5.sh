#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== mx debug =="
timeout 300 python tools/mx_debug.py 2>&1 | grep -c ": ok"; timeout 300 python tools/mx_debug.py 2>&1 | grep "bad elements" | head
echo "== mx traces =="
for cfg in "9=3|-" "|-" "|116" "9=3|116"; do
  tune=${cfg%%|*}; var=${cfg##*|}
  echo "---- tune [$tune] variant [$var]"
  for w in "14336 4096 32,0,0,0,32,64,0,0" "4096 14336 32,0,0,0,32,64,0,0" "14336 4096 32,32,32,32,32,32,32,32" "4096 14336 32,32,32,32,32,32,32,32"; do
    if [ "$var" = "-" ]; then AO_GEMM8_TUNE=$tune timeout 300 python tools/mx_rb_trace.py $w 2>&1 | grep -v "ret = \|RuntimeWarning\|per step {d\|exit us by XCD\|loop ticks per step p10" | tail -4 | cut -c1-400
    else AO_GEMM8_TUNE=$tune timeout 300 python tools/mx_rb_trace.py $w - $var 2>&1 | grep -v "ret = \|RuntimeWarning\|per step {d\|exit us by XCD\|loop ticks per step p10" | tail -4 | cut -c1-400; fi
  done
done | tee $O/mx_trace.txt
echo "== bench mx, product vs 116 =="
for v in 0 116; do
  timeout 600 python bench.py --configs mx --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph --steps 3 --warmup 1 --gemm-variant $v 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['mxfp8_mixtral_bs64']; print('variant $v', 'multinomial tok/s %.0f frac %.3f ms %.3f' % (c['value'], c['roofline']['frac'], c['ms_per_step']), '| uniform16 tok/s %.0f frac %.3f ms %.3f' % (c['uniform16']['value'], c['uniform16']['roofline']['frac'], c['uniform16']['ms_per_step']))"
done
