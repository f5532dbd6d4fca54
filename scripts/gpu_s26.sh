#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s26
mkdir -p $O
cd $R
echo "== pre-fix library: the new test must FAIL, the stress loop must show the 4-row garbage =="
AO_MI355_LIB=tools/bin/_C_mi355_prefix.so timeout 600 python -m pytest tests/test_8bit_gpu.py -m gpu -q --timeout 500 -x -k "race_screen and pair" 2>&1 | grep -a "passed\|failed\|assert\|Error" | head -8 | tee $O/prefix_test.log
AO_MI355_LIB=tools/bin/_C_mi355_prefix.so timeout 600 python tools/stress_mx_pair.py --iters 800 --seed 5 --noise 0 2>&1 | grep "^{" > $O/stress_prefix.jsonl; tail -1 $O/stress_prefix.jsonl
echo "== fixed library =="
timeout 600 python -m pytest tests/test_8bit_gpu.py -m gpu -q --timeout 500 -x -k "race_screen and pair" 2>&1 | grep -a "passed\|failed\|assert\|Error" | head -8 | tee $O/fixed_test.log
timeout 900 python tools/stress_mx_pair.py --iters 3000 --seed 5 --noise 0 2>&1 | grep "^{" > $O/stress_fixed.jsonl; tail -1 $O/stress_fixed.jsonl
timeout 900 python tools/stress_mx_pair.py --iters 2000 --seed 6 --noise 1 --n 14336 2>&1 | grep "^{" > $O/stress_fixed_14336.jsonl; tail -1 $O/stress_fixed_14336.jsonl
python - <<'PY'
import json
for f in ("stress_prefix", "stress_fixed", "stress_fixed_14336"):
    big = small = 0
    for l in open(f"gpurun_out/s26/{f}.jsonl"):
        d = json.loads(l)
        for v in d.get("mismatch", {}).values():
            if v["max_ulps_vs_single"] > 2: big += 1
            else: small += 1
    print(f, "outputs with garbage:", big, "outputs with a 1-ulp element:", small)
PY
timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_subclass_gpu.py tests/test_fuzz_gpu.py -m gpu -q --timeout 600 -k "mx" 2>&1 | grep -a "passed\|failed" | tail -2
timeout 400 python tools/fuzz_long.py --seconds 240 --seed 21 --kinds mxdyn,mx,fp8,dyn 2>&1 | grep "^{" | tee $O/fuzz_after_fix.jsonl | cut -c1-300 | tail -5
timeout 600 python bench.py --configs mx --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['mxfp8_mixtral_bs64']; print('multinomial tok/s %.0f frac %.3f' % (c['value'], c['roofline']['frac']), '| uniform16 tok/s %.0f frac %.3f' % (c['uniform16']['value'], c['uniform16']['roofline']['frac']))"
