#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/n2dbg
for L in 8 32; do
echo "== B: 2 ranks, no configs, layers $L"; AO_BENCH_SHARE_GPU=1 AO_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2952$((L%10)) bench.py --gpus 2 --steps 5 --warmup 2 --no-configs --layers $L > gpurun_out/n2dbg/b$L.json 2> gpurun_out/n2dbg/b$L.err; echo rc=$?; grep -E "fault|Error|error" gpurun_out/n2dbg/b$L.err | head -3; cut -c1-200 gpurun_out/n2dbg/b$L.json
done
echo "== D: 2 ranks, no configs, 32 layers, eager"; AO_BENCH_SHARE_GPU=1 AO_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29527 bench.py --gpus 2 --steps 5 --warmup 2 --no-configs --no-graph > gpurun_out/n2dbg/d.json 2> gpurun_out/n2dbg/d.err; echo rc=$?; grep -E "fault|Error|error" gpurun_out/n2dbg/d.err | head -3; cut -c1-200 gpurun_out/n2dbg/d.json
