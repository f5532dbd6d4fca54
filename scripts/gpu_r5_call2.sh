#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c2; mkdir -p $O; cd $R
echo "== tests =="; timeout 900 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_fuzz_gpu.py -q -x --timeout 600 2>&1 | tail -5 | tee $O/tests.log
{
for t in "" "1=32" "1=32,2=4" "1=64,2=4"; do timeout 120 python tools/fp8_rb_trace.py 128 1280 8192 101 "$t"; done
for t in "" "1=32"; do timeout 120 python tools/fp8_rb_trace.py 128 8192 1024 101 "$t"; done
for t in "" "1=64"; do timeout 120 python tools/fp8_rb_trace.py 128 8192 3584 101 "$t"; done
timeout 120 python tools/fp8_rb_trace.py 128 7168 8192 101 ""
timeout 120 python tools/fp8_rb_trace.py 1024 8192 1024 101 ""
} 2>&1 | grep -v amdgpu.ids > $O/trace.txt
grep -E "^M=|per workgroup|storing|mean ticks" $O/trace.txt
F="default"; for bn in 32 64 128; do for s in 0 1 2 3 4 6 8; do F="$F,rb+bn$bn+s$s"; done; done
echo "== sweep =="; timeout 1200 python tools/midm_sweep.py --ms 128,256,512,1024 --forms $F > $O/sweep.jsonl 2>$O/sweep.err; tail -2 $O/sweep.err
python - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/c2/sweep.jsonl")
rows=[json.loads(l) for l in open(p) if l.startswith("{") and "form" in l]
key=lambda r:(r["shape"],r["M"]); seen=[]
for r in rows:
    if key(r) not in seen: seen.append(key(r))
for k in seen:
    rs=[r for r in rows if key(r)==k and "us" in r]
    core=[r for r in rs if r["form"]=="core"][0]["us"]; d=[r for r in rs if r["form"]=="default"][0]["us"]
    best=min((r for r in rs if r["form"] not in ("core",)), key=lambda r:r["us"])
    top=sorted((r for r in rs if r["form"] not in ("core","default")), key=lambda r:r["us"])[:4]
    print(k, f"core={core} default={d} best={best['form']}={best['us']} ratio={core/best['us']:.2f} |", " ".join(f"{r['form'][3:]}={r['us']}" for r in top))
PY
