#!/bin/bash
# Final validation: full parity suite, smoke, headline bench (+ torchrun world=1 path), rocprofv3 stats + PMC
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest gpu ==" ; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tee $O/pytest_gpu.log | tail -5
echo "== smoke ==" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench ==" ; timeout 600 python bench.py 2>$O/bench.err | tee $O/bench_r01.json | cut -c1-1500
tail -3 $O/bench.err
echo "== bench via torchrun, world 1 ==" ; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-second-layout 2>$O/bench_tr.err | tee $O/bench_torchrun.json | cut -c1-400
tail -3 $O/bench_tr.err
cd /tmp && export TMPDIR=/tmp
echo "== rocprof stats =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o int4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout > $O/rocprof_stats.log 2>&1
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-300
echo "== rocprof pmc =="
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_fetch -o int4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-layout > $O/rocprof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_write -o int4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-layout > $O/rocprof_pmc_write.log 2>&1
cd $R
python scripts/pmc_summary.py $O/prof_pmc_fetch $O/prof_pmc_write -o $O/int4_pmc_r01.json | grep -A8 "int4_mm" | head -40
find $O/prof_pmc_fetch $O/prof_pmc_write -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
echo "== secondary workloads =="
timeout 600 python tools/bench_8bit.py --m 2048 --iters 10 2>/dev/null | grep "^{" > $O/bench_8bit_r01.jsonl; wc -l $O/bench_8bit_r01.jsonl
timeout 300 python bench.py --batch 128 --steps 20 --warmup 3 --no-cpu-baseline --no-second-layout 2>/dev/null > $O/bench_bs128.json; cut -c1-200 $O/bench_bs128.json
timeout 120 python tools/bench_moe_pad.py 2>/dev/null | grep "^{" > $O/bench_moe_pad.json; cut -c1-200 $O/bench_moe_pad.json
