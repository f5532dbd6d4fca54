#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== dq variants ==" ; timeout 300 python scripts/check_dq_variants.py 2>&1 | grep -v Warn | tee $O/dq_variants.txt
summ() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('tok/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'sumk', round(r['sum_kernel_ms_per_step'],3), 'frac', round(r['frac'],3), {k: (round(v['us'],2), round(v['GBps'])) for k, v in r['per_shape'].items()})"; }
for cfg in "0 22" "0 24" "0 34" "0 1"; do
  set -- $cfg
  echo "== bench wpb=$1 mode=$2 ==" ; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wpb $1 --mode $2 2>/dev/null | summ
done
