#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s3
mkdir -p $O
cd $R
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== mx parity =="
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_dispatcher_gpu.py -m gpu -q --timeout 600 -k "mx or fp16_activation" 2>&1 | tail -6
echo "== mx traces: round-3 protocol + equal shares | new protocol + equal | new + 5:4 | new + 6:5 | new + 4:3 =="
for tune in "9=101,10=3" "9=101" "" "9=605" "9=403"; do
  echo "---- tune [$tune]"
  for w in "14336 4096 32,0,0,0,32,64,0,0" "4096 14336 32,0,0,0,32,64,0,0" "14336 4096 32,32,32,32,32,32,32,32" "4096 14336 32,32,32,32,32,32,32,32"; do
    AO_GEMM8_TUNE=$tune timeout 300 python tools/mx_rb_trace.py $w 2>&1 | grep -v "ret = \|RuntimeWarning\|per step {d" | tail -6 | cut -c1-420
  done
done | tee $O/mx_trace.txt
