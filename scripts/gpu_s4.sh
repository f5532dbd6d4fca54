#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s4
mkdir -p $O
cd $R
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== mx debug =="
timeout 300 python tools/mx_debug.py 2>&1 | grep -c ": ok"; timeout 300 python tools/mx_debug.py 2>&1 | grep "bad elements" | head
echo "== mx parity =="
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_dispatcher_gpu.py -m gpu -q --timeout 600 -k "mx or fp16_activation" 2>&1 | tail -4
echo "== mx traces =="
for cfg in "9=101,10=3|-" "9=101|-" "|-" "|116" "10=3|116"; do
  tune=${cfg%%|*}; var=${cfg##*|}
  echo "---- tune [$tune] variant [$var]"
  for w in "14336 4096 32,0,0,0,32,64,0,0" "4096 14336 32,0,0,0,32,64,0,0" "14336 4096 32,32,32,32,32,32,32,32" "4096 14336 32,32,32,32,32,32,32,32"; do
    if [ "$var" = "-" ]; then AO_GEMM8_TUNE=$tune timeout 300 python tools/mx_rb_trace.py $w 2>&1 | grep -v "ret = \|RuntimeWarning\|per step {d" | tail -6 | cut -c1-420
    else AO_GEMM8_TUNE=$tune timeout 300 python tools/mx_rb_trace.py $w - $var 2>&1 | grep -v "ret = \|RuntimeWarning\|per step {d" | tail -6 | cut -c1-420; fi
  done
done | tee $O/mx_trace.txt
