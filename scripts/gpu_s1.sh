#!/bin/bash
# round 6, session 1: the new BASELINE-shape parity cases, MX stream-K traces with wall-clock stamps, L2 hit counters of the MX kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || exit 3
echo "== new parity cases =="
timeout 900 python -m pytest tests/test_baseline_scale_gpu.py -m gpu -q --timeout 600 -k "bs128 or benched_chunk" 2>&1 | tail -15
echo "== mx traces =="
for w in "14336 4096 32,0,0,0,32,64,0,0" "4096 14336 32,0,0,0,32,64,0,0" "14336 4096 32,32,32,32,32,32,32,32" "4096 14336 32,32,32,32,32,32,32,32"; do
  timeout 300 python tools/mx_rb_trace.py $w 2>&1 | tail -4
done | tee $O/mx_trace.txt
echo "== mx pmc: L2 hits =="
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -o mx -- python $R/tools/mx_rb_trace.py 14336 4096 32,0,0,0,32,64,0,0 > $O/pmc_l2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/pmc_ea -o mx -- python $R/tools/mx_rb_trace.py 14336 4096 32,0,0,0,32,64,0,0 > $O/pmc_ea.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("pmc_l2", "pmc_ea"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/s1/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if "mx_stream_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: (sum(v) / len(v), len(v)) for k, v in acc.items()})
PY
find $O -name "*.csv" -size +2M -delete
echo "== full gpu suite =="
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -5
