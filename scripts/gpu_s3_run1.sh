#!/bin/bash
# Session-3 GPU pass 1: parity tests, smoke, bench, rocprof kernel stats.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== rocminfo ==" ; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc
echo "== pytest gpu ==" ; timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tee $O/pytest_gpu.log | tail -15
echo "== smoke ==" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench ==" ; timeout 600 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-4000
tail -5 $O/bench.err
echo "== rocprof =="
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_s3 -o int4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof.log 2>&1
f=$(find $O/prof_s3 -name "*kernel_stats.csv" | head -1); echo $f; head -8 "$f"
