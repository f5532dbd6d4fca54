#!/bin/bash
# rb8_kernel with register-ring weights: parity of every kind, then per-step trace vs ring depth / the LDS-ring form
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s11
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py tests/test_variants_gpu.py -m gpu -q --timeout 600 2>&1 | tail -8
for cfg in "- 0" "tools/bin/_C_mi355_ring4.so 0" "tools/bin/_C_mi355_ring6.so 0" "- 120"; do
  echo "== library / variant: $cfg"
  for sizes in 32,0,0,0,0,0,0,0 32,0,0,0,32,64,0,0 32,0,32,16,16,0,32,0 16,16,16,16,16,16,16,16 128,128,128,128,128,128,128,128; do
    timeout 300 python tools/mx_rb_trace.py 14336 4096 $sizes $cfg 2>/dev/null
    timeout 300 python tools/mx_rb_trace.py 4096 14336 $sizes $cfg 2>/dev/null
  done
done 2>&1 | tee $O/mx_trace.txt
timeout 600 python bench.py --no-second-layout --configs fp8,mx --steps 10 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s11/bench.json').read().strip().splitlines()[-1])
c=d['configs']['mxfp8_mixtral_bs64']; print('mx config', round(c['value']), c['ms_per_step'], round(c['roofline']['achieved']), round(c['roofline']['frac'],3))
c=d['configs']['fp8_tp8_shards']; print('fp8 shards', {k:(round(v['tokens_per_s']), round(v['frac'],3)) for k,v in c['by_M'].items()})
P
timeout 600 python tools/bench_8bit.py --which fp8 --m 128 --iters 20 2>&1 | tee $O/bench8_fp8.jsonl | cut -c1-230
