#!/bin/bash
# round 3: bs = 128 int4 with the dequant's t + z add on the VALU instead of the matrix pipe (profiling library), product form (0), producer form (900), K-half waves (840)
mkdir -p gpurun_out
for lib in "" tools/bin/_C_mi355_valu.so; do
  echo "== library: ${lib:-product}"
  AO_MI355_LIB=${lib:+$PWD/$lib} timeout 600 python tools/int4_modes.py --batch 128 --layout five --modes 0,900,840 --rounds 2 --steps 5 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d.get('mode'), round(d.get('tokens_per_s_median') or d.get('tokens_per_s') or 0), 'rel', d.get('max_rel_vs_mode0'), {k:round(v,1) for k,v in (d.get('per_shape_us') or {}).items()})
"
done
