#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
summ() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('tok/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'sumk', round(r['sum_kernel_ms_per_step'],3), 'frac', round(r['frac'],3), {k: (round(v['us'],2), round(v['GBps'])) for k, v in r['per_shape'].items()})"; }
echo "== mx diag =="; timeout 300 python scripts/mx_diag.py 2>&1 | tail -8
for cfg in "0 0" "0 1" "0 34" "0 20" "4 1" "16 1"; do
  set -- $cfg
  echo "== bench wpb=$1 mode=$2 ==" ; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wpb $1 --mode $2 2>/dev/null | summ
done
