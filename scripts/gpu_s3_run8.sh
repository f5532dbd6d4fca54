#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest int4 ==" ; timeout 600 python -m pytest tests/test_int4_gpu.py -m gpu -q -x --timeout 120 2>&1 | tail -4
echo "== lab ==" ; timeout 120 tools/int4_lab 0:100 0:0 8:0 0:403 2>&1 | tee $O/lab_s3_8.txt
echo "== bench ==" ; timeout 300 python bench.py --no-cpu-baseline 2>$O/bench.err | tee $O/bench8.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('tok/s', round(d['value'],1), 'unmerged', round(d['config']['unmerged_tokens_per_s'],1), 'ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), {k: (round(v['us'],2), round(v['GBps'])) for k, v in r['per_shape'].items()})"
tail -3 $O/bench.err
