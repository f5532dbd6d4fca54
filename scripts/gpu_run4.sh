#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/pmc
cd $R
summ() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('tok/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'sumk', round(r['sum_kernel_ms_per_step'],3), 'frac', round(r['frac'],3), {k: (round(v['us'],2), round(v['GBps'])) for k, v in r['per_shape'].items()})"; }
for cfg in "0 2" "0 32" "0 22"; do
  set -- $cfg
  echo "== bench wpb=$1 mode=$2 ==" ; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wpb $1 --mode $2 2>/dev/null | summ
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/pmc/counters_list.txt 2>&1
grep -c . $O/pmc/counters_list.txt
for mode in 0 22; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/m${mode}_$tag -o p -- python $R/bench.py --layers 2 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --mode $mode > $O/pmc/m${mode}_$tag.log 2>&1
    echo "mode $mode set $tag rc=$?"
  done
done
find $O/pmc -name "*.csv" | head -20
