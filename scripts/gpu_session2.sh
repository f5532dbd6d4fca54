#!/bin/bash
# round-2 session 2: new BASELINE-scale parity tests + TP building blocks, full GPU suite, restructured bench.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
echo "== new tests ==" ; timeout 1200 python -m pytest tests/test_baseline_scale_gpu.py -m gpu -q -x --timeout 600 2>&1 | tee $O/pytest_new.log | tail -15
echo "== full gpu suite ==" ; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_baseline_scale_gpu.py 2>&1 | tee $O/pytest_gpu.log | tail -8
echo "== smoke ==" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench ==" ; ( time timeout 900 python bench.py ) 2>$O/bench.err | tee $O/bench.json | cut -c1-3000
tail -5 $O/bench.err
