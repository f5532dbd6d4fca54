#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest int4 ==" ; timeout 900 python -m pytest tests/test_int4_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -6
for mode in 0 99; do
echo "== bench bs=128 mode $mode ==" ; timeout 300 python bench.py --batch 128 --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout --mode $mode 2>$O/bench.err | tee $O/bench_bs128_m$mode.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('tok/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), {k: (round(v['us'],2), v['kernel']) for k, v in r['per_shape'].items()})"
tail -2 $O/bench.err
done
