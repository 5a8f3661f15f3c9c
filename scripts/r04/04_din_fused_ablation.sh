#!/bin/bash
# Round 4, GPU call: where k_din_fused's slot loop spends its time -- ablation variants (library built with -DSPRK_DF_XP; bits: 1 no
# MFMAs, 2 no product split, 4 no h32, 8 no PReLU dot, 16 no reduce / sigmoid, 32 no pooling, 64 no row loads), attention-only launches,
# strict order, BASELINE config 3.  Results are garbage by construction (--no-check); the time is the point.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_04
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for xp in ${XPS:-0 1 2 4 8 16 32 64 3 56 60 63 65 126 127}; do
  SPRK_DF_XP=$xp timeout 200 python bench.py --workload din_c3 --steps 60 --warmup 6 $STRICT 2>$O/xp$xp.err | tail -1 > $O/xp$xp.json
  python - $O/xp$xp.json $xp <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('XP=%3s attention-only %.2f us' % (sys.argv[2], l['roofline']['avg_launch_us']))
except Exception as e:
    print('XP=%s FAILED %s' % (sys.argv[2], e))
PY
done
