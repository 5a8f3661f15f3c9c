#!/bin/bash
# Round 4: what the fused launch's TAIL costs (library built with -DSPRK_DF_XP): 128 no folded-row gathers, 256 no fc1, 512 no fc0
# MFMAs (numerics + pooled history), 896 all three; one launch per batch, BASELINE config 3.  Garbage results by construction.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_12
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for xp in 0 128 256 512 896 0; do
  SPRK_DF_XP=$xp timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/xp$xp.err | tail -1 > $O/xp$xp.json
  python - $O/xp$xp.json $xp <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('XP=%3s fused step %.2f us   attention-only %.2f us' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us']))
except Exception as e:
    print('XP=%s FAILED %s' % (sys.argv[2], e))
PY
done
