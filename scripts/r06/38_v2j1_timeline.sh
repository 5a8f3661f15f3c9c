#!/bin/bash
# Round 6 (the tree of commit 0a3d9be): k_deepfm_v2_joint1's timeline on BASELINE config 2 (Infinity-Cache-resident and HBM-resident tables), -DSPRK_DF_XP build.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_41}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
cp scripts/r06/libsparrow_hip_xp.so sparrowrecsys_amd/libsparrow_hip.so
SPRK_V2J1_TS_FILE=$O/ts_c2.bin timeout 200 python bench.py --steps 60 --warmup 10 --input-batches 32 $STRICT 2>$O/c2.err | tail -1 > $O/c2.json
SPRK_V2J1_TS_FILE=$O/ts_c2_hbm.bin timeout 300 python bench.py --steps 60 --warmup 10 --big-vocab 8388608 --input-batches 32 $STRICT 2>$O/c2_hbm.err | tail -1 > $O/c2_hbm.json
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
for w in c2 c2_hbm; do
  python -c "
import json;l=json.loads(open('$O/$w.json').read());print('$w (stamped build): kernel %.2f us' % l['roofline']['avg_launch_us'])"
  python scripts/r04/v2j1_timeline.py $O/ts_$w.bin | tee $O/timeline_$w.txt
done
