#!/bin/bash
# First GPU pass: probe torch core, parity tests, smoke, bench, wpb sweep, rocprof.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== rocminfo ==" ; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -8
echo "== probe ==" ; timeout 300 python scripts/gpu_probe_torch_core.py 2>&1 | grep -v Warning | tee $O/probe.log | tail -30
echo "== pytest gpu ==" ; timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tee $O/pytest_gpu.log | tail -40
echo "== smoke ==" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== bench ==" ; timeout 600 python bench.py --steps 30 --warmup 5 2>$O/bench.err | tee $O/bench.json | cut -c1-3000
tail -5 $O/bench.err
for w in 4 8 16; do
  echo "== bench wpb=$w ==" ; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wpb $w 2>/dev/null | tee $O/bench_wpb$w.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('tok/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), {k: (round(v['us'],2), round(v['GBps'])) for k, v in r['per_shape'].items()})"
done
echo "== rocprof ==" 
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r01 -o int4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof.log 2>&1
ls -R $O/prof_r01 | head -20
f=$(find $O/prof_r01 -name "*kernel_stats.csv" | head -1); echo $f; head -12 "$f"
