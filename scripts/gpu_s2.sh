#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== new parity cases =="
timeout 900 python -m pytest tests/test_baseline_scale_gpu.py tests/test_dispatcher_gpu.py -m gpu -q --timeout 600 -k "bs128 or benched_chunk or fp16_activation" 2>&1 | tail -8
echo "== mx traces =="
i=0
for w in "14336 4096 32,0,0,0,32,64,0,0" "4096 14336 32,0,0,0,32,64,0,0" "14336 4096 32,32,32,32,32,32,32,32" "4096 14336 32,32,32,32,32,32,32,32" "14336 4096 32,0,0,0,32,32,0,0"; do
  AO_TRACE_DUMP=$O/trace_$i.npy timeout 300 python tools/mx_rb_trace.py $w 2>&1 | grep -v "ret = \|RuntimeWarning" | tail -7
  i=$((i+1))
done | tee $O/mx_trace.txt
echo "== rest of gpu suite =="
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5
