#!/bin/bash
# round-2 session 1: issue-cost lab, pure-read ceiling, in-situ A/B of decode-kernel builds, HEAD bench + rocprofv3 stats + SQ PMC
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
echo "== dequant lab ==" ; timeout 120 tools/bin/dequant_lab 2>&1 | tee $O/dequant_lab.txt
echo "== layer probe ==" ; timeout 120 tools/bin/layer_probe 2>&1 | tee $O/layer_probe.txt
echo "== int4 modes ==" ; timeout 600 python tools/int4_modes.py --modes 0,90,91,92,89 --wpbs 0,4 --rounds 3 2>$O/modes.err | tee $O/int4_modes.jsonl
tail -3 $O/modes.err
timeout 300 python tools/int4_modes.py --modes 0,91,92 --wpbs 0 --layout five --rounds 3 2>>$O/modes.err | tee $O/int4_modes_five.jsonl
cd /tmp && export TMPDIR=/tmp
echo "== rocprof stats (HEAD, mode 0) =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o int4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout > $O/rocprof_stats.log 2>&1
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-300
echo "== rocprof SQ pmc =="
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/prof_pmc_sq -o int4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-layout > $O/rocprof_pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_pmc_sq2 -o int4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-layout > $O/rocprof_pmc_sq2.log 2>&1
cd $R
python scripts/pmc_summary.py $O/prof_pmc_sq $O/prof_pmc_sq2 -o $O/int4_pmc_sq.json --source "rocprofv3 --pmc SQ_* (two passes), bench.py --steps 2" | grep -A24 '"int4_mm_kernel"' | head -40
find $O -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
du -sh $O
