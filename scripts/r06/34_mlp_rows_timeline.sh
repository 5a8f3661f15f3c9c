#!/bin/bash
# Round 6: k_mlp_rows' stamped timeline (config 5, -DSPRK_DF_XP build in scripts/r06/libsparrow_hip_xp.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_35}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
cp scripts/r06/libsparrow_hip_xp.so sparrowrecsys_amd/libsparrow_hip.so
for w in widedeep_c5 embedding_mlp_ref; do
  SPRK_MR_TS_FILE=$O/ts_$w.bin timeout 200 python bench.py --workload $w --steps 40 --warmup 8 $STRICT 2>$O/ts_$w.err | tail -1 > $O/ts_$w.json
  python scripts/r06/mlp_rows_timeline.py $O/ts_$w.bin $O/ts_$w.json | tee $O/timeline_$w.txt
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
timeout 200 python bench.py --workload widedeep_c5 --steps 100 --warmup 8 $STRICT 2>/dev/null | tail -1 | python -c "import sys,json;l=json.loads(sys.stdin.read());print('product: strict %.2f us' % l['roofline']['avg_launch_us'])" | tee -a $O/timeline_widedeep_c5.txt
