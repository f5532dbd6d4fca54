#!/bin/bash
# rb8_kernel with the activation-producer wave: parity of every kind + timing of the configs it serves
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s9
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py tests/test_variants_gpu.py -m gpu -q --timeout 600 -x 2>&1 | tail -8
timeout 600 python bench.py --no-second-layout --configs fp8,mx --steps 10 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s9/bench.json').read().strip().splitlines()[-1])
c=d['configs']['mxfp8_mixtral_bs64']; print('mx config', round(c['value']), c['ms_per_step'], round(c['roofline']['achieved']), round(c['roofline']['frac'],3))
c=d['configs']['fp8_tp8_shards']; print('fp8 shards', {k:(round(v['tokens_per_s']), round(v['frac'],3)) for k,v in c['by_M'].items()})
P
timeout 600 python tools/bench_8bit.py --which mx --iters 20 2>&1 | tee $O/bench8_mx_ragged.jsonl | cut -c1-190
timeout 600 python tools/bench_8bit.py --which mx --m 128 --iters 20 2>&1 | tee $O/bench8_mx_128.jsonl | cut -c1-190
timeout 600 python tools/bench_8bit.py --which mx --m 1024 --iters 20 2>&1 | tee $O/bench8_mx_1024.jsonl | cut -c1-190
timeout 600 python tools/bench_8bit.py --which fp8 --m 128 --iters 20 2>&1 | tee $O/bench8_fp8.jsonl | cut -c1-230
timeout 600 python tools/bench_8bit.py --which int8 --m 64 --iters 20 2>&1 | tee $O/bench8_int8_m64.jsonl | cut -c1-230
