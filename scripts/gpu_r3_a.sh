#!/bin/bash
# round-3 call A: in-situ A/Bs (bs = 128 tile shapes, bs = 1 split-K / multi-tile / graph branches) + int4 parity subset
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3a; mkdir -p $O; cd $R
echo "== int4 parity subset =="; timeout 600 python -m pytest tests/test_int4_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3
echo "== bs128 modes =="; timeout 900 python tools/int4_modes.py --batch 128 --layout five --modes 0,800,810,820,830,802,812,804 --rounds 3 --steps 10 2>$O/bs128.err | tee $O/bs128_modes.jsonl | cut -c1-330
echo "== bs64/256 modes =="; for b in 64 256; do timeout 600 python tools/int4_modes.py --batch $b --layout five --modes 0,800,810 --rounds 3 --steps 10 2>>$O/bs128.err | tee -a $O/bs_other_modes.jsonl | cut -c1-330; done
echo "== bs1 modes =="; timeout 900 python tools/int4_modes.py --batch 1 --layout five --modes 0,202,204,212,214,222,224 --rounds 5 --steps 20 2>$O/bs1.err | tee $O/bs1_modes.jsonl | cut -c1-330
echo "== bs1 merged modes =="; timeout 900 python tools/int4_modes.py --batch 1 --layout merged --modes 0,202,214,224 --rounds 5 --steps 20 2>>$O/bs1.err | tee $O/bs1_modes_merged.jsonl | cut -c1-330
echo "== branches =="; timeout 600 python tools/int4_branches.py --rounds 5 2>$O/br.err | tee $O/branches.jsonl
tail -3 $O/bs128.err $O/bs1.err $O/br.err
