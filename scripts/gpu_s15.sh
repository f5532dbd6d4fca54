#!/bin/bash
# round 6, session 15: the persistent GEMM as the product dispatch -- the 8-bit suites, then the int8 / fp8 config lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s15
mkdir -p $O
cd $R
echo "== 8-bit suites =="
timeout 1500 python -m pytest tests/test_8bit_gpu.py tests/test_baseline_scale_gpu.py tests/test_fuzz_gpu.py tests/test_subclass_gpu.py tests/test_variants_gpu.py tests/test_dispatcher_gpu.py tests/test_isa_structure.py tests/test_parallel_2proc_gpu.py -m gpu -q --timeout 900 -x 2>&1 | tail -8 | tee $O/pytest.log
echo "== bench int8, fp8 =="
timeout 900 python bench.py --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph --configs int8,fp8 > $O/bench_int8_fp8.json 2> $O/bench.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/s15/bench_int8_fp8.json"))
c = d["configs"]
i8 = c["int8_dyn_bs128x2048"]; f8 = c["fp8_tp8_shards"]
print("int8: value", round(i8["value"]), "gemm frac", round(i8["roofline"]["frac"], 4), "gemm ms", i8["roofline"].get("gemm_ms_per_layer"), "cast ms", i8["roofline"].get("act_cast_ms_per_layer"))
print("fp8:", {k: (round(v["tokens_per_s"]), round(v["frac"], 4)) for k, v in f8["by_M"].items()})
PY
