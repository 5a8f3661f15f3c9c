#!/bin/bash
# Round 5: the stamped timeline alone (scripts/r05/libsparrow_hip_xp.so = this tree built with -DSPRK_DF_XP)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_07}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/strict.err | tail -1 > $O/strict.json
python -c "
import json;l=json.loads(open('$O/strict.json').read());print('product build: step %.2f us, attention-only %.2f us' % (l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us']))"
cp scripts/r05/libsparrow_hip_xp.so sparrowrecsys_amd/libsparrow_hip.so
SPRK_DF_XP=1024 SPRK_DF_TS_FILE=$O/ts.bin timeout 200 python bench.py --workload din_c3 --steps 40 --warmup 8 $STRICT 2>$O/ts.err | tail -1 > $O/ts.json
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
python scripts/r04/din_fused_timeline.py $O/ts.bin $O/ts.json | tee $O/timeline.txt
