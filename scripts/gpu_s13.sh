#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s13
mkdir -p $O
cd $R
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== headline A/B: pruned int4 kernel source vs the committed one (same other objects), alternating =="
for i in 1 2 3; do
  for lib in "" tools/bin/_C_mi355_old_int4.so; do
    AO_MI355_LIB=$lib timeout 300 python bench.py --no-configs --no-second-layout --no-cpu-baseline --no-stack-baseline --no-subclass-graph --steps 50 --warmup 5 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib [$lib]', 'tok/s %.1f ms %.4f event_frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
  done
done | tee $O/headline_ab.txt
echo "== full gpu suite =="
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8
