#!/bin/bash
# round 3: gemm8_p8<fp8> with its MFMAs pinned inside the phases (inline asm) -- parity tests, per-shape numbers, config 4
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_variants_gpu.py tests/test_dispatcher_gpu.py -x -q -m gpu -k "fp8 or float8 or gemm8 or scaled" > gpurun_out/fp8_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/fp8_tests.log
tail -4 gpurun_out/fp8_tests.log
grep -q "rc=0" gpurun_out/fp8_tests.log || exit 1
timeout 600 python tools/bench_8bit.py --which fp8 --m 2048 --iters 20 2>&1 | grep -v amdgpu > gpurun_out/bench_8bit_m2048.jsonl
timeout 600 python tools/bench_8bit.py --which fp8l --m 8192 --iters 10 2>&1 | grep -v amdgpu > gpurun_out/bench_8bit_fp8l_m8192.jsonl
python - <<'PY'
import json
for f in ('gpurun_out/bench_8bit_m2048.jsonl','gpurun_out/bench_8bit_fp8l_m8192.jsonl'):
    for l in open(f):
        try: d=json.loads(l)
        except Exception: continue
        if d['kernel']=='fp8_scaled_mm' and d['M'] in (2048,8192): print(d['shape'].ljust(14), d['M'], d['N'], d['K'], round(d['us'],1),'us', round(d['TFLOPs']),'TF')
PY
timeout 600 python bench.py --configs fp8 --no-tp-graph --no-stack-baseline --no-subclass-graph --no-cpu-baseline --no-second-layout --steps 10 --warmup 2 > gpurun_out/bench_fp8.json 2> gpurun_out/bench_fp8.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_fp8.json').read().strip().splitlines()[-1])
c=d['configs']['fp8_tp8_shards']; print({m:(round(v['tokens_per_s']), round(v['frac'],3)) for m,v in c['by_M'].items()})
PY
