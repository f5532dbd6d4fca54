#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c5; mkdir -p $O; cd $R
echo "== tests =="; timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_fuzz_gpu.py tests/test_variants_gpu.py -q -x --timeout 600 2>&1 | tail -3 | tee $O/tests.log
echo "== sweep fp8 =="; timeout 1200 python tools/midm_sweep.py --ms 128,256,512,768,1024 --forms default,nolocal,v100 > $O/sweep_fp8.jsonl 2>$O/sweep.err; tail -2 $O/sweep.err
echo "== sweep int8 =="; timeout 1200 python tools/midm_sweep.py --kinds int8 --ms 128,512,1024 --forms default,v100 > $O/sweep_int8.jsonl 2>>$O/sweep.err
python - <<'PY'
import json,os
for f in ("sweep_fp8.jsonl","sweep_int8.jsonl"):
    p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/c5",f)
    rows=[json.loads(l) for l in open(p) if l.startswith("{") and "form" in l]
    key=lambda r:(r["shape"],r["M"]); seen=[]
    for r in rows:
        if key(r) not in seen: seen.append(key(r))
    w=0
    for k in seen:
        rs={r["form"]:r.get("us") for r in rows if key(r)==k}
        w+= rs["default"]<=rs["core"]
        print(f, k, rs, f"ratio={rs['core']/rs['default']:.2f}")
    print("wins", w, "of", len(seen))
PY
