#!/bin/bash
# Regenerates everything under profiles/ that DESIGN.md / bench.py cite for the current HEAD, on one MI355X:
#   bash scripts/gpu_profile.sh [out-dir under gpurun_out, default prof]
# full parity suite + smoke, the full bench line, the torchrun (world 1) and TP-over-RCCL (world 1) paths, rocprofv3
# --kernel-trace --stats of the headline command, PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_* in separate runs, as the gfx950 guide
# prescribes: never together with a trace domain), the 8-bit GEMM's MFMA-busy counters, and the secondary micro-benchmarks.
# (Round 3 lesson: a box whose first GPU access faults -- "Memory access fault by GPU node-2" 0.3 s after HSA init, before any kernel of
# this repository had been launched -- left a rocprofv3-wrapped python hanging until its timeout and cost 15 GPU-minutes.  Every
# rocprofv3 step below therefore runs under its own short timeout, and the script starts with a 60-second canary.)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 60 python -c "import torch; torch.zeros(1, device='cuda').add_(1).item(); print('canary ok')" || { echo "GPU canary failed: not profiling on this box"; exit 3; }
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
cd $R
echo "== pytest gpu ==" ; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tee $O/pytest_gpu.log | tail -4
echo "== smoke ==" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== full bench ==" ; ( time timeout 900 python bench.py ) 2>$O/bench.err > $O/bench.json; tail -4 $O/bench.err; cut -c1-300 $O/bench.json
echo "== torchrun world 1 ==" ; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-second-layout --no-configs 2>$O/tr.err | cut -c1-200
echo "== TP over RCCL, world 1 ==" ; timeout 600 python bench.py --force-tp --configs tp --no-second-layout --no-cpu-baseline --steps 5 2>$O/tp.err > $O/bench_tp.json; python -c "
import json; d=json.loads(open('$O/bench_tp.json').read().strip().splitlines()[-1]); json.dump({'fp8_tp': d['configs']['fp8_tp']}, open('$O/bench_tp_world1.json','w'), indent=1); print({k: round(v['per_gpu_TFLOPs']) for k, v in d['configs']['fp8_tp']['by_M'].items()})"
echo "== N = 2 code path, dry run: two ranks sharing the GPU over gloo (numbers meaningless; the line, the rank-0 printing and the TP = 2 config must work) =="
AO_BENCH_ONE_SHOT=1 AO_BENCH_SHARE_GPU=1 AO_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_dryrun.json 2>$O/n2.err; python -c "
import json; out=open('$O/bench_n2_dryrun.json').read().strip().splitlines(); d=json.loads(out[-1]); print('stdout lines', len(out), 'n_gpus', d['n_gpus'], 'fp8_tp keys', sorted(d['configs']['fp8_tp'].get('by_M', d['configs']['fp8_tp'])))"
cd /tmp && export TMPDIR=/tmp
HEAD_CMD="python $R/bench.py --warmup 1 --no-cpu-baseline --no-second-layout --no-configs"
echo "== rocprof stats, headline =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o int4 -- $HEAD_CMD --steps 10 > $O/rocprof_stats.log 2>&1
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/int4_kernel_stats.csv; head -3 "$f" | cut -c1-250
echo "== rocprof pmc, headline (separate passes) =="
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_fetch -o int4 -- $HEAD_CMD --steps 2 > $O/rocprof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_write -o int4 -- $HEAD_CMD --steps 2 > $O/rocprof_pmc_write.log 2>&1
[ "${AO_PROFILE_FULL:-0}" = 1 ] && timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/prof_pmc_sq -o int4 -- $HEAD_CMD --steps 2 > $O/rocprof_pmc_sq.log 2>&1
[ "${AO_PROFILE_FULL:-0}" = 1 ] && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_pmc_sq2 -o int4 -- $HEAD_CMD --steps 2 > $O/rocprof_pmc_sq2.log 2>&1
echo "== rocprof pmc + stats, secondary configs (one bench.py process per config; FETCH_SIZE and WRITE_SIZE in separate passes) =="
CFG_CMD="python $R/bench.py --warmup 1 --steps 2 --no-cpu-baseline --no-second-layout --no-stack-baseline --no-subclass-graph"
# (fp8: under counter collection the 80-layer config did not finish in 900 s in round 3; the counters are per launch, so 4 layers do)
for c in int4_bs128 int8 mx fp8; do
  EXTRA=""; [ $c = fp8 ] && EXTRA="--fp8-layers 4"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_cfg_${c}_fetch -o cfg -- $CFG_CMD --configs $c $EXTRA > $O/rocprof_cfg_${c}_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_cfg_${c}_write -o cfg -- $CFG_CMD --configs $c $EXTRA > $O/rocprof_cfg_${c}_write.log 2>&1
  python $R/scripts/pmc_summary.py $O/prof_cfg_${c}_fetch $O/prof_cfg_${c}_write -o $O/cfg_${c}_pmc.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --configs $c --steps 2" > /dev/null
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg_stats -o cfg -- $CFG_CMD --configs int4_bs128,fp8,mx --fp8-layers 8 > $O/rocprof_cfg_stats.log 2>&1
f=$(find $O/prof_cfg_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/configs_kernel_stats.csv && head -8 "$f" | cut -c1-160
O=$O python - <<'PY'
import json, os
O = os.environ["O"]
dom = {"int4_bs128": "int4_mm_rb_kernel", "int8": "gemm8_p8p_kernel", "mx": "mx_stream_kernel", "fp8": "gemm8_p8_kernel"}
out = {"source": "scripts/gpu_profile.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes per config; FETCH x1024 x2 (gfx950), WRITE x1024 (uncalibrated); "
                 "mean over the dispatches of the config's dominant kernel", "configs": {}}
for c, k in dom.items():
    try:
        d = json.load(open(f"{O}/cfg_{c}_pmc.json"))["kernels"]
    except Exception as e:  # noqa: BLE001
        out["configs"][c] = {"error": repr(e)}
        continue
    if k in d:
        e = d[k]
        out["configs"][c] = {"kernel": k, "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "hbm_read_bytes_per_launch": e.get("hbm_read_bytes_per_launch"),
                             "hbm_write_bytes_per_launch": e.get("hbm_write_bytes_per_launch"), "dispatches": e.get("dispatches"),
                             "other_kernels": {kk: vv.get("hbm_bytes_per_launch") for kk, vv in d.items() if kk != k}}
# the MX config times TWO workloads in one process: the first half of mx_stream_kernel's dispatches is the multinomial draw (3 of 8 experts
# hit), the second half 16 tokens on every expert -- split them (the config's roofline line quotes the multinomial half)
try:
    import csv, glob
    def _vals(d, counter):
        rows = []
        for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                if "mx_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        return [v for _, v in sorted(rows)]
    fe, wr = _vals("prof_cfg_mx_fetch", "FETCH_SIZE"), _vals("prof_cfg_mx_write", "WRITE_SIZE")
    h = len(fe) // 2
    if h and len(wr) == len(fe) and "mx" in out["configs"] and "hbm_bytes_per_launch" in out["configs"]["mx"]:
        by = {}
        for name, (fa, wa) in {"multinomial": (fe[:h], wr[:h]), "uniform16": (fe[h:], wr[h:])}.items():
            rd, ww = sum(fa) / len(fa) * 1024 * 2, sum(wa) / len(wa) * 1024
            by[name] = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": ww, "hbm_bytes_per_launch": rd + ww, "dispatches": len(fa)}
        m = out["configs"]["mx"]
        m["note"] = "first half of the kernel's dispatches = the multinomial draw, second half = 16 tokens on every expert; the top-level fields are the multinomial half (all-dispatch mean: %.0f)" % m["hbm_bytes_per_launch"]
        m["by_workload"] = by
        for k in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch"):
            m[k] = by["multinomial"][k]
except Exception as e:  # noqa: BLE001
    out["configs"].setdefault("mx", {})["split_error"] = repr(e)
json.dump(out, open(f"{O}/configs_pmc.json", "w"), indent=1)
print({c: (round(v.get("hbm_bytes_per_launch") or 0) if isinstance(v, dict) else v) for c, v in out["configs"].items()})
PY
cd $R
python scripts/pmc_summary.py $O/prof_pmc_fetch $O/prof_pmc_write -o $O/int4_pmc.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --no-configs" | grep -A8 '"int4_mm_kernel"' | head -12
find $O -name "*counter_collection.csv" -size +6M -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -size +6M -delete 2>/dev/null
# the parts below did not change in round 4 (GEMM MFMA-busy counters, SQ counters of the int4 decode kernel, tile-shape sweeps): AO_PROFILE_FULL=1 re-collects them
[ "${AO_PROFILE_FULL:-0}" = 1 ] || { echo "== round-4 micro-benchmarks =="; for b in 1 16; do timeout 300 python tools/bench_fp8_int4.py --batch $b 2>/dev/null | grep "^{"; done > $O/fp8_int4.jsonl; timeout 300 python tools/bench_dec8.py --ms 1 > $O/dec8.jsonl 2>/dev/null; du -sh $O; exit 0; }
cd /tmp
echo "== rocprof 8-bit GEMM: stats + MFMA-busy pmc =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gemm_stats -o gemm8 -- python $R/tools/bench_8bit.py --which int8,fp8l --m 8192 --iters 5 > $O/rocprof_gemm_stats.log 2>&1
f=$(find $O/prof_gemm_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/gemm8_kernel_stats.csv; head -5 "$f" | cut -c1-200
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/prof_gemm_pmc -o gemm8 -- python $R/tools/bench_8bit.py --which int8,fp8l --m 8192 --iters 3 > $O/rocprof_gemm_pmc.log 2>&1
cd $R
python scripts/pmc_summary.py $O/prof_pmc_fetch $O/prof_pmc_write -o $O/int4_pmc.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --no-configs" | grep -A8 '"int4_mm_kernel"' | head -12
python scripts/pmc_summary.py $O/prof_pmc_sq $O/prof_pmc_sq2 -o $O/int4_pmc_sq.json --source "rocprofv3 --pmc SQ_* (two passes), bench.py --steps 2 --no-configs" | grep -A20 '"int4_mm_kernel"' | head -24
O=$O python - <<'PY'
import csv, glob, json, os, collections
O = os.environ["O"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/prof_gemm_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        n = row["Kernel_Name"]
        if "gemm8_p8_kernel" in n or "gemm8_p8p_kernel" in n:
            k = ("gemm8_p8p_kernel<" if "gemm8_p8p_kernel" in n else "gemm8_p8_kernel<") + ("int8" if "<0>" in n or "<1>" in n else "fp8") + ">"
        elif "gemm8_dma_kernel" in n:
            k = "gemm8_dma_kernel"
        else:
            continue
        acc[(k, row.get("Grid_Size", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for (k, grid), c in acc.items():
    e = {cn: sum(v) / len(v) for cn, v in c.items()}
    e["dispatches"] = max(len(v) for v in c.values())
    # MfmaUtil = matrix-pipe busy cycles / (SQ busy cycles per SE x 1024 SIMDs / 32 SEs): SQ_BUSY_CYCLES is summed over 32 SEs
    if e.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        e["MfmaUtil"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["SQ_BUSY_CYCLES"] / 32 * 1024)
    out[k + " grid=" + grid] = e
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... -- tools/bench_8bit.py --which int8,fp8l --m 8192", "kernels": out},
          open(O + "/gemm8_p8_pmc_mfma.json", "w"), indent=1)
for k, e in out.items():
    print(k, "MfmaUtil", round(e.get("MfmaUtil", 0), 3), "dispatches", e["dispatches"])
PY
find $O -name "*counter_collection.csv" -size +6M -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -size +6M -delete 2>/dev/null
echo "== secondary micro-benchmarks =="
for m in 512 1024 2048 8192; do for v in 0 8 32; do timeout 300 python tools/bench_8bit.py --which int8,fp8l --m $m --iters 10 --gemm-variant $v 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('M=$m v=$v', d['kernel'], d['shape'], d['N'], d['K'], round(d['us'],1), round(d.get('TOPs', d.get('TFLOPs', 0))))"; done; done > $O/gemm8_variants.txt; tail -4 $O/gemm8_variants.txt
timeout 300 python tools/bench_8bit.py --which mx --iters 20 2>/dev/null | grep "^{" > $O/mx_grouped_ragged.jsonl
timeout 300 python tools/bench_8bit.py --which mx --m 128 --iters 20 2>/dev/null | grep "^{" > $O/mx_grouped_16rows.jsonl
timeout 300 python tools/bench_8bit.py --which mx --m 1024 --iters 20 2>/dev/null | grep "^{" > $O/mx_grouped_128rows.jsonl
timeout 300 python tools/bench_8bit.py --which quant,fp8 --m 128 --iters 20 2>/dev/null | grep "^{" > $O/bench_8bit_m128.jsonl
timeout 120 python tools/bench_moe_pad.py 2>/dev/null | grep "^{" > $O/bench_moe_pad.json
for b in 1 16; do timeout 300 python tools/bench_fp8_int4.py --batch $b 2>/dev/null | grep "^{"; done > $O/fp8_int4.jsonl
du -sh $O
