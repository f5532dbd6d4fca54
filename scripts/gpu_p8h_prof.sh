cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/p8h_prof; mkdir -p $O
CMD="python $R/tools/midm_sweep.py --ms 768,1024 --kinds fp8,int8 --families 70b,8b --forms default --no-core"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p8h -- $CMD > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "^\"?Name|p8h|p8_kernel|rb8_kernel" "$f" | cut -c1-400 > $O/p8h_kernel_stats.csv
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o p8h -- $CMD > $O/pmc.log 2>&1
cd $R; python scripts/pmc_mfma_summary.py $O/pmc -o $O/p8h_pmc_mfma.json --source "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... -- tools/midm_sweep.py --ms 768,1024 --kinds fp8,int8 --forms default --no-core (product dispatch, cold weights)"
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/p8h_kernel_stats.csv | cut -c1-200
