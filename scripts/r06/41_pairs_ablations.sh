#!/bin/bash
# Round 6: k_deepfm_pairs1's loads, ablated one kind at a time (config 2, -DV1_XP=bits builds, WRONG RESULTS, timing only): 1 one ids load per lane instead of
# six, 2 the three genre fields' rows not requested, 4 no second first-order load, 8 the deep part's own rows not requested, 16 no numerics loads, 31 all
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_45}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
MANY="--cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe --variants 0"
for rep in 1 2; do
  for v in product 1 2 4 8 16 31; do
    if [ $v = product ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_v1xp$v.so sparrowrecsys_amd/libsparrow_hip.so; fi
    a=$(timeout 300 python bench.py --workload deepfm_c2 --steps 400 --warmup 40 $STRICT 2>>$O/err.txt | tail -1 | python -c "import sys,json;l=json.loads(sys.stdin.read());print('strict %.2f us' % l['roofline']['avg_launch_us'])")
    b=$(timeout 300 python bench.py --workload deepfm_c2 --steps 400 --warmup 40 $MANY 2>>$O/err.txt | tail -1 | python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.2f us/step at %s per launch' % (1e3*l['ms_per_step'], l['config'].get('batches_per_launch')))")
    echo "V1_XP=$v: $a | $b" | tee -a $O/timing.txt
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
