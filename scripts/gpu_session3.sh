#!/bin/bash
# p8 GEMM: correctness / race screen, then A/B in bench's int8 + fp8 configs
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_8bit_gpu.py -m gpu -q -k "every_kernel_variant or race_screen" --timeout 600 2>&1 | tee $O/pytest.log | tail -15
for v in 0 32; do
  echo "== bench int8,fp8 gemm-variant $v =="
  timeout 600 python bench.py --configs int8,fp8 --no-second-layout --no-cpu-baseline --steps 5 --gemm-variant $v 2>$O/bench_v$v.err > $O/bench_v$v.json
  python - <<PY
import json
d=json.loads(open("$O/bench_v$v.json").read().strip().splitlines()[-1])
for k,c in d["configs"].items():
    if "error" in c: print(k, c); continue
    print(k, round(c["value"]), c["roofline"]["achieved"], c["roofline"].get("end_to_end_TOPs"), {m:(round(x["tokens_per_s"]),round(x["TFLOPs"])) for m,x in c.get("by_M",{}).items()})
PY
done
