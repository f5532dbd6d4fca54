#!/bin/bash
# end-of-round check at HEAD: the driver's own sequence (GPU tests, smoke, default bench line)
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tee gpurun_out/final/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python bench.py ) 2> gpurun_out/final/bench.err > gpurun_out/final/bench.json; tail -4 gpurun_out/final/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], {k:(round(v.get('value',0)), round(v.get('roofline',{}).get('frac',0),3)) for k,v in d['configs'].items() if isinstance(v,dict)})
PY
