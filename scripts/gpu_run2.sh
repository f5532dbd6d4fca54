#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== ubench ==" ; timeout 300 ./tools/ubench_valu 2>&1 | tee $O/ubench_valu.txt
summ() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('tok/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'sumk', round(r['sum_kernel_ms_per_step'],3), 'frac', round(r['frac'],3), {k: (round(v['us'],2), round(v['GBps'])) for k, v in r['per_shape'].items()})"; }
for cfg in "0 0" "0 1" "0 2" "0 12" "0 18" "4 1" "16 1" "4 18" "16 12"; do
  set -- $cfg
  echo "== bench wpb=$1 mode=$2 ==" ; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wpb $1 --mode $2 2>/dev/null | summ
done
echo "== rocprof ==" 
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01 -o int4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof.log 2>&1
ls $O/prof_r01 | head; f=$(find $O/prof_r01 -name "*kernel_stats.csv" | head -1); echo $f; head -8 "$f"
