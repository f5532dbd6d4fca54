#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== cvt scale probe =="; timeout 60 tools/probe_cvt_scale 2>&1 | tee $O/probe_cvt_scale.txt
echo "== stream probe =="; timeout 300 tools/stream_probe 2>&1 | tee $O/stream_probe.txt
echo "== ubench mix =="; timeout 300 tools/ubench_mix > $O/ubench_mix.txt 2>&1; wc -l $O/ubench_mix.txt
