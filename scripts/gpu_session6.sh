#!/bin/bash
# TP path on one GPU (world 1), rocprofv3 stats + PMC at HEAD (headline kernel; p8 GEMM MfmaUtil), full bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
echo "== tp world1 ==" ; timeout 600 python bench.py --force-tp --configs tp --no-second-layout --no-cpu-baseline --steps 5 2>$O/tp.err | tee $O/bench_tp.json | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d.get('configs',{}).get('fp8_tp'), indent=0)[:1500])"
tail -3 $O/tp.err
echo "== torchrun world1 ==" ; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout --no-configs 2>$O/tr.err | cut -c1-300
tail -2 $O/tr.err
cd /tmp && export TMPDIR=/tmp
echo "== rocprof stats headline =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o int4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout --no-configs > $O/rocprof_stats.log 2>&1
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); head -3 "$f" | cut -c1-250
echo "== rocprof pmc fetch/write headline =="
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_fetch -o int4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-layout --no-configs > $O/rocprof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_write -o int4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-layout --no-configs > $O/rocprof_pmc_write.log 2>&1
echo "== rocprof gemm8: stats + mfma pmc =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gemm_stats -o gemm8 -- python $R/tools/bench_8bit.py --which int8,fp8l --m 8192 --iters 5 > $O/rocprof_gemm_stats.log 2>&1
f=$(find $O/prof_gemm_stats -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-200
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/prof_gemm_pmc -o gemm8 -- python $R/tools/bench_8bit.py --which int8,fp8l --m 8192 --iters 3 > $O/rocprof_gemm_pmc.log 2>&1
cd $R
python scripts/pmc_summary.py $O/prof_pmc_fetch $O/prof_pmc_write -o $O/int4_pmc_r02.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --no-configs" | grep -A8 '"int4_mm_kernel"' | head -12
python - <<'PY'
import csv, glob, json, os, collections
O=os.environ.get("O","gpurun_out/s6")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/prof_gemm_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        n=row["Kernel_Name"]
        if "gemm8_p8_kernel" in n: k="gemm8_p8_kernel<"+("int8" if "<0>" in n or "<1>" in n else "fp8")+">"
        elif "gemm8_dma_kernel" in n: k="gemm8_dma_kernel"
        else: continue
        grid=row.get("Grid_Size","")
        acc[(k,grid)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out={}
for (k,grid),c in acc.items():
    e={cn: sum(v)/len(v) for cn,v in c.items()}
    e["dispatches"]=max(len(v) for v in c.values())
    # MfmaUtil = matrix-pipe busy cycles / (SQ busy cycles per SE x 1024 SIMDs / 32 SEs): SQ_BUSY_CYCLES is summed over 32 SEs
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]>0:
        e["MfmaUtil"]=e["SQ_VALU_MFMA_BUSY_CYCLES"]/(e["SQ_BUSY_CYCLES"]/32*1024)
    out[k+" grid="+grid]=e
json.dump({"source":"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... -- tools/bench_8bit.py --which int8,fp8l --m 8192","kernels":out}, open(O+"/gemm8_p8_pmc_mfma.json","w"), indent=1)
for k,e in out.items(): print(k, "MfmaUtil", round(e.get("MfmaUtil",0),3), "dispatches", e["dispatches"])
PY
find $O -name "*counter_collection.csv" -size +6M -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -size +6M -delete 2>/dev/null
echo "== full bench ==" ; ( time timeout 900 python bench.py ) 2>$O/bench.err > $O/bench.json; tail -4 $O/bench.err; cut -c1-400 $O/bench.json
du -sh $O
