#!/bin/bash
# Round 6: k_din_fused's stamped timeline on the current tree (SPRK_BUILD_DEFINES=-DSPRK_DF_XP builds ONE unit since round 6, so the stamps are found)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_22}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
cp scripts/r06/libsparrow_hip_xp.so sparrowrecsys_amd/libsparrow_hip.so
SPRK_DF_XP=1024 SPRK_DF_TS_FILE=$O/ts.bin timeout 200 python bench.py --workload din_c3 --steps 40 --warmup 8 $STRICT 2>$O/ts.err | tail -1 > $O/ts.json
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
python scripts/r04/din_fused_timeline.py $O/ts.bin $O/ts.json | tee $O/timeline.txt
