#!/bin/bash
# round 6: the blocked scale layout op on the box, the sweep table next to hipBLASLt with the refitted plan, the fp8 shard config
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s36
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_moe_pad.py tests/test_dispatcher_gpu.py tests/test_abi.py tests/test_host_dispatch.py -q --timeout 600 2>&1 | tail -15 | tee $O/pytest.log
timeout 900 python tools/midm_sweep.py --ms 128,256,512,768,1024,2048 --kinds fp8,int8 --families 70b,8b --forms default 2>&1 | grep "^{" > $O/midm_final.jsonl
python tools/midm_table.py $O/midm_final.jsonl | tail -30
timeout 900 python bench.py --configs fp8 --no-second-layout --no-cpu-baseline --no-stack-baseline --no-subclass-graph --steps 20 2>$O/bench.err | tail -1 > $O/bench_fp8.json
python -c "
import json; d=json.load(open('$O/bench_fp8.json')); print({k:v for k,v in d.items() if k.startswith('fp8_')})"
