#!/bin/bash
# round 5, call 1: same-XCD meeting + LDS epilogue + priming order of rb8 -- parity, then the mid-M sweep next to hipBLASLt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c1
mkdir -p $O
cd $R
timeout 120 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== new tests =="; timeout 600 python -m pytest tests/test_8bit_gpu.py -q -x -k "same_xcd" --timeout 300 2>&1 | tail -15 | tee $O/new_tests.log
echo "== 8bit + baseline + fuzz tests =="; timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_fuzz_gpu.py tests/test_variants_gpu.py -q --timeout 600 2>&1 | tail -8 | tee $O/tests.log
echo "== sweep M=128 =="; timeout 600 python tools/midm_sweep.py --ms 128 --forms default,nolocal,rb+bn32,rb+bn64,rb+bn128,rb+bn64+s4,rb+bn64+s8,rb+bn32+s4,rb+bn32+s8,rb+bn128+s2,rb+bn128+s4,rb+bn128+s8 > $O/sweep_m128.jsonl 2>$O/sweep_m128.err; tail -3 $O/sweep_m128.err
echo "== sweep M=256..1024 =="; timeout 900 python tools/midm_sweep.py --ms 256,512,1024 --forms default,nolocal,rb,rb+bn128+s2,rb+bn128+s4,rb+bn64+s4,tile,p8 > $O/sweep_mid.jsonl 2>$O/sweep_mid.err; tail -3 $O/sweep_mid.err
python - <<'PY'
import json,sys,os
for f in ("sweep_m128.jsonl","sweep_mid.jsonl"):
    p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/c1",f)
    rows=[json.loads(l) for l in open(p) if l.startswith("{") and "form" in l]
    key=lambda r:(r["shape"],r["M"])
    seen=[]
    for r in rows:
        if key(r) not in seen: seen.append(key(r))
    for k in seen:
        rs=[r for r in rows if key(r)==k]
        print(k, " ".join(f"{r['form']}={r.get('us','ERR')}" for r in rs))
PY
