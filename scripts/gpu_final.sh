#!/bin/bash
# The end-of-round pass (round 6): rocprofv3 stats + PMC of the headline command FIRST (five-shape layout, then the merged layout's own
# PMC pass), copied into profiles/ on the box, so that the bench line that follows quotes the profile of the same build; the secondary
# configs' PMC traffic and kernel stats; the 8-bit GEMM's MFMA-busy counters; then the full bench line, the whole GPU suite, smoke, and
# the two multi-rank code paths a one-GPU box can exercise.  Every step under its own timeout; counters never together with a trace domain.
#   bash scripts/gpu_final.sh [out-dir under gpurun_out, default final]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=r06
timeout 90 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || { echo "GPU canary failed"; exit 3; }
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HEAD_CMD="python $R/bench.py --warmup 1 --no-cpu-baseline --no-second-layout --no-configs"
echo "== rocprof stats, headline =="
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o int4 -- $HEAD_CMD --steps 10 > $O/rocprof_stats.log 2>&1
f=$(find $O/prof_stats -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/int4_kernel_stats.csv; cp "$f" $R/profiles/int4_kernel_stats_$ROUND.csv; grep "^{" $O/rocprof_stats.log | tail -1 > $O/bench_under_rocprof.json; cp $O/bench_under_rocprof.json $R/profiles/bench_${ROUND}_under_rocprof.json; grep int4_mm_kernel "$f" | cut -c1-60,230-330; else echo "no kernel_stats.csv"; tail -5 $O/rocprof_stats.log; fi
# the same process's kernel trace cut to the line's timed replays: sum of kernel time per step next to the step time (round 6)
( cd $R; timeout 300 python scripts/rocprof_crosscheck.py $O/prof_stats $O/bench_under_rocprof.json -o profiles/int4_rocprof_crosscheck_$ROUND.json | grep -E "ms_per_step|avg_kernel_us|frac_of|sum_kernel_le_span|span_over" )
echo "== rocprof pmc, headline (separate passes), five-shape then merged layout =="
for lay in five merged; do
  EXTRA=""; [ $lay = merged ] && EXTRA="--merged"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_fetch_$lay -o int4 -- $HEAD_CMD --steps 2 $EXTRA > $O/rocprof_pmc_fetch_$lay.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_write_$lay -o int4 -- $HEAD_CMD --steps 2 $EXTRA > $O/rocprof_pmc_write_$lay.log 2>&1
  ( cd $R; timeout 120 python scripts/pmc_summary.py $O/prof_pmc_fetch_$lay $O/prof_pmc_write_$lay -o $O/int4_pmc_$lay.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --no-configs $EXTRA" > $O/pmc_summary_$lay.out 2>&1 )
done
O=$O R=$R ROUND=$ROUND python - <<'PY'
import json, os
O, R, ROUND = os.environ["O"], os.environ["R"], os.environ["ROUND"]
for lay, name in (("five", f"int4_pmc_{ROUND}.json"), ("merged", f"int4_pmc_merged_{ROUND}.json")):
    try:
        d = json.load(open(f"{O}/int4_pmc_{lay}.json"))
        k = d["kernels"]["int4_mm_kernel"]
        assert k["hbm_bytes_per_launch"] > 1e6
        json.dump(d, open(f"{R}/profiles/{name}", "w"), indent=1)
        print("pmc ok:", lay, "bytes per launch", round(k["hbm_bytes_per_launch"]))
    except Exception as e:
        print("pmc summary not usable for", lay, repr(e))
PY
echo "== rocprof pmc + stats, secondary configs =="
CFG_CMD="python $R/bench.py --warmup 1 --steps 2 --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph"
for c in int4_bs128 int8 mx fp8; do
  EXTRA=""; [ $c = fp8 ] && EXTRA="--fp8-layers 4"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_cfg_${c}_fetch -o cfg -- $CFG_CMD --configs $c $EXTRA > $O/rocprof_cfg_${c}_fetch.log 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_cfg_${c}_write -o cfg -- $CFG_CMD --configs $c $EXTRA > $O/rocprof_cfg_${c}_write.log 2>&1
  ( cd $R; python scripts/pmc_summary.py $O/prof_cfg_${c}_fetch $O/prof_cfg_${c}_write -o $O/cfg_${c}_pmc.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --configs $c --steps 2" > /dev/null )
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg_stats -o cfg -- $CFG_CMD --configs int4_bs128,int8,fp8,mx --fp8-layers 8 > $O/rocprof_cfg_stats.log 2>&1
f=$(find $O/prof_cfg_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/configs_kernel_stats.csv && cp "$f" $R/profiles/configs_kernel_stats_$ROUND.csv && head -8 "$f" | cut -c1-160
O=$O R=$R ROUND=$ROUND python - <<'PY'
import csv, glob, json, os
O, R, ROUND = os.environ["O"], os.environ["R"], os.environ["ROUND"]
dom = {"int4_bs128": "int4_mm_rb_kernel", "int8": "gemm8_p8p_kernel", "mx": "mx_stream_kernel", "fp8": "gemm8_p8_kernel"}
out = {"source": "scripts/gpu_final.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes per config; FETCH x1024 x2 (gfx950), WRITE x1024 (uncalibrated); "
                 "mean over the dispatches of the config's dominant kernel (int4_bs128: int4_mm_rb_kernel serves qkv / o / down, int4_mm_w32_kernel gate / up -- both listed)", "configs": {}}
for c, k in dom.items():
    try:
        d = json.load(open(f"{O}/cfg_{c}_pmc.json"))["kernels"]
    except Exception as e:
        out["configs"][c] = {"error": repr(e)}
        continue
    if k in d:
        e = d[k]
        out["configs"][c] = {"kernel": k, "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "hbm_read_bytes_per_launch": e.get("hbm_read_bytes_per_launch"),
                             "hbm_write_bytes_per_launch": e.get("hbm_write_bytes_per_launch"), "dispatches": e.get("dispatches"),
                             "other_kernels": {kk: vv.get("hbm_bytes_per_launch") for kk, vv in d.items() if kk != k}}
try:
    def _vals(d, counter):
        rows = []
        for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                if "mx_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        return [v for _, v in sorted(rows)]
    fe, wr = _vals("prof_cfg_mx_fetch", "FETCH_SIZE"), _vals("prof_cfg_mx_write", "WRITE_SIZE")
    h = len(fe) // 2
    if h and len(wr) == len(fe) and "hbm_bytes_per_launch" in out["configs"].get("mx", {}):
        by = {}
        for name, (fa, wa) in {"multinomial": (fe[:h], wr[:h]), "uniform16": (fe[h:], wr[h:])}.items():
            rd, ww = sum(fa) / len(fa) * 1024 * 2, sum(wa) / len(wa) * 1024
            by[name] = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": ww, "hbm_bytes_per_launch": rd + ww, "dispatches": len(fa)}
        m = out["configs"]["mx"]
        m["note"] = "first half of the kernel's dispatches = the multinomial draw, second half = 16 tokens on every expert; the top-level fields are the multinomial half"
        m["by_workload"] = by
        for k in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch"):
            m[k] = by["multinomial"][k]
except Exception as e:
    out["configs"].setdefault("mx", {})["split_error"] = repr(e)
json.dump(out, open(f"{O}/configs_pmc.json", "w"), indent=1)
ok = sum(1 for v in out["configs"].values() if isinstance(v, dict) and v.get("hbm_bytes_per_launch"))
if ok >= 3:
    json.dump(out, open(f"{R}/profiles/configs_pmc_{ROUND}.json", "w"), indent=1)
print({c: (round(v.get("hbm_bytes_per_launch") or 0) if isinstance(v, dict) else v) for c, v in out["configs"].items()})
PY
echo "== rocprof 8-bit GEMM: MFMA-busy pmc =="
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/prof_gemm_pmc -o gemm8 -- python $R/tools/bench_8bit.py --which int8,fp8l --m 8192 --iters 3 > $O/rocprof_gemm_pmc.log 2>&1
( cd $R; python scripts/pmc_mfma_summary.py $O/prof_gemm_pmc -o profiles/gemm8_p8_pmc_mfma_$ROUND.json --source "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... -- tools/bench_8bit.py --which int8,fp8l --m 8192 (round 6: the persistent 256 x 256 form takes these shapes)" )
find $O -name "*counter_collection.csv" -size +4M -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
cd $R
echo "== full bench ==" ; ( time timeout 1200 python bench.py ) 2>$O/bench.err > $O/bench.json; tail -4 $O/bench.err; cut -c1-300 $O/bench.json; cp $O/bench.json $R/profiles/bench_$ROUND.json
echo "== pytest gpu ==" ; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --durations=15 2>&1 | tee $O/pytest_gpu.log | tail -4; cp $O/pytest_gpu.log $R/profiles/pytest_gpu_$ROUND.log
echo "== smoke ==" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== TP over RCCL, world 1 ==" ; timeout 400 python bench.py --force-tp --configs tp --no-second-layout --no-cpu-baseline --steps 5 2>$O/tp.err > $O/bench_tp.json; python -c "
import json; d=json.loads(open('$O/bench_tp.json').read().strip().splitlines()[-1]); json.dump({'fp8_tp': d['configs']['fp8_tp']}, open('$R/profiles/bench_${ROUND}_tp_world1.json','w'), indent=1); print({k: round(v['per_gpu_TFLOPs']) for k, v in d['configs']['fp8_tp']['by_M'].items()})"
echo "== N = 2 code path, dry run (two ranks sharing the GPU over gloo; numbers meaningless) =="
AO_BENCH_ONE_SHOT=1 AO_BENCH_SHARE_GPU=1 AO_BENCH_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_dryrun.json 2>$O/n2.err; cp $O/bench_n2_dryrun.json $R/profiles/bench_${ROUND}_n2_dryrun_gloo.json; python -c "
import json; out=open('$O/bench_n2_dryrun.json').read().strip().splitlines(); d=json.loads(out[-1]); print('stdout lines', len(out), 'n_gpus', d['n_gpus'], 'fp8_tp keys', sorted(d['configs']['fp8_tp'].get('by_M', d['configs']['fp8_tp'])))"
echo "== the watchdog: a TP leg that cannot finish in 1 s must still leave ONE line =="
timeout 300 python bench.py --force-tp --configs tp --no-second-layout --no-cpu-baseline --steps 5 --tp-timeout 1 2>/dev/null | python -c "
import sys, json; out=sys.stdin.read().strip().splitlines(); d=json.loads(out[-1]); print('lines', len(out), 'value', round(d['value']), 'fp8_tp', str(d['configs']['fp8_tp'])[:90])"
mkdir -p $R/gpurun_out/profiles_out; cp $R/profiles/*_$ROUND* $R/gpurun_out/profiles_out/ 2>/dev/null
du -sh $O
