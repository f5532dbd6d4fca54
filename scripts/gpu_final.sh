#!/bin/bash
# The short end-of-round pass (scripts/gpu_profile.sh is the long one): rocprofv3 stats + PMC of the headline command FIRST, copied into
# profiles/ on the box, so that the bench line that follows quotes the profile of the same build; then the full bench line, the whole GPU
# suite, smoke, and the two multi-rank code paths a one-GPU box can exercise.  Every step under its own timeout.
#   bash scripts/gpu_final.sh [out-dir under gpurun_out, default final]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=r04
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || { echo "GPU canary failed"; exit 3; }
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HEAD_CMD="python $R/bench.py --warmup 1 --no-cpu-baseline --no-second-layout --no-configs"
echo "== rocprof stats, headline =="
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o int4 -- $HEAD_CMD --steps 10 > $O/rocprof_stats.log 2>&1
f=$(find $O/prof_stats -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/int4_kernel_stats.csv; cp "$f" $R/profiles/int4_kernel_stats_$ROUND.csv; grep "^{" $O/rocprof_stats.log | tail -1 > $O/bench_under_rocprof.json; grep int4_mm_kernel "$f" | cut -c1-60,230-330; else echo "no kernel_stats.csv"; tail -5 $O/rocprof_stats.log; fi
echo "== rocprof pmc, headline (separate passes) =="
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_fetch -o int4 -- $HEAD_CMD --steps 2 > $O/rocprof_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_write -o int4 -- $HEAD_CMD --steps 2 > $O/rocprof_pmc_write.log 2>&1
cd $R
timeout 120 python scripts/pmc_summary.py $O/prof_pmc_fetch $O/prof_pmc_write -o $O/int4_pmc.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --no-configs" > $O/pmc_summary.out 2>&1
python - <<PY
import json
try:
    d = json.load(open("$O/int4_pmc.json"))
    k = d["kernels"]["int4_mm_kernel"]
    assert k["hbm_bytes_per_launch"] > 1e6
    json.dump(d, open("$R/profiles/int4_pmc_$ROUND.json", "w"), indent=1)
    print("pmc ok: bytes per launch", round(k["hbm_bytes_per_launch"]))
except Exception as e:
    print("pmc summary not usable, the committed file stays:", repr(e))
PY
find $O -name "*counter_collection.csv" -size +4M -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
echo "== full bench ==" ; ( time timeout 900 python bench.py ) 2>$O/bench.err > $O/bench.json; tail -4 $O/bench.err; cut -c1-300 $O/bench.json
echo "== pytest gpu ==" ; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tee $O/pytest_gpu.log | tail -4
echo "== smoke ==" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== TP over RCCL, world 1 ==" ; timeout 400 python bench.py --force-tp --configs tp --no-second-layout --no-cpu-baseline --steps 5 2>$O/tp.err > $O/bench_tp.json; python -c "
import json; d=json.loads(open('$O/bench_tp.json').read().strip().splitlines()[-1]); json.dump({'fp8_tp': d['configs']['fp8_tp']}, open('$O/bench_tp_world1.json','w'), indent=1); print({k: round(v['per_gpu_TFLOPs']) for k, v in d['configs']['fp8_tp']['by_M'].items()})"
echo "== N = 2 code path, dry run (two ranks sharing the GPU over gloo; numbers meaningless) =="
AO_BENCH_ONE_SHOT=1 AO_BENCH_SHARE_GPU=1 AO_BENCH_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_dryrun.json 2>$O/n2.err; python -c "
import json; out=open('$O/bench_n2_dryrun.json').read().strip().splitlines(); d=json.loads(out[-1]); print('stdout lines', len(out), 'n_gpus', d['n_gpus'], 'fp8_tp keys', sorted(d['configs']['fp8_tp'].get('by_M', d['configs']['fp8_tp'])))"
echo "== the watchdog: a TP leg that cannot finish in 1 s must still leave ONE line =="
timeout 300 python bench.py --force-tp --configs tp --no-second-layout --no-cpu-baseline --steps 5 --tp-timeout 1 2>/dev/null | python -c "
import sys, json; out=sys.stdin.read().strip().splitlines(); d=json.loads(out[-1]); print('lines', len(out), 'value', round(d['value']), 'fp8_tp', str(d['configs']['fp8_tp'])[:90])"
du -sh $O
