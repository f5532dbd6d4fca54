#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s32
mkdir -p $O
cd $R
timeout 700 python tools/fuzz_long.py --seconds 600 --seed 61 2>&1 | grep "^{" | tee -a $O/fuzz.jsonl | cut -c1-400 | tail -6
timeout 700 python tools/fuzz_long.py --seconds 600 --seed 62 --hog 1 2>&1 | grep "^{" | tee -a $O/fuzz.jsonl | cut -c1-400 | tail -6
timeout 700 python tools/stress_mx_pair.py --iters 4000 --seed 9 --noise 1 --hog 1 2>&1 | grep "^{" > $O/stress.jsonl; tail -1 $O/stress.jsonl
python - <<'PY'
import json
big = small = 0
for l in open("gpurun_out/s32/stress.jsonl"):
    d = json.loads(l)
    for v in d.get("mismatch", {}).values():
        if v["max_ulps_vs_single"] > 2: big += 1
        else: small += 1
print("stress: outputs with garbage:", big, "with a 1-ulp element:", small)
PY
