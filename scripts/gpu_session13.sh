#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s13
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py tests/test_int4_gpu.py -m gpu -q --timeout 600 2>&1 | tail -3
timeout 600 python bench.py --no-second-layout --configs fp8,mx --steps 10 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python - <<'P'
import json,sys
d=json.loads(open('gpurun_out/s13/bench.json').read().strip().splitlines()[-1])
c=d['configs']['mxfp8_mixtral_bs64']; print('mx config', round(c['value']), c['ms_per_step'], round(c['roofline']['achieved']), round(c['roofline']['frac'],3))
c=d['configs']['fp8_tp8_shards']; print('fp8 shards', {k:(round(v['tokens_per_s']), round(v['frac'],3)) for k,v in c['by_M'].items()})
P
for sizes in 32,0,0,0,32,64,0,0 32,0,32,16,16,0,32,0 16,16,16,16,16,16,16,16 128,128,128,128,128,128,128,128; do
  timeout 300 python tools/mx_rb_trace.py 14336 4096 $sizes 2>/dev/null | cut -c1-330
  timeout 300 python tools/mx_rb_trace.py 4096 14336 $sizes 2>/dev/null | cut -c1-330
done 2>&1 | tee $O/mx_trace.txt
