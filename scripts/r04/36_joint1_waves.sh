#!/bin/bash
# Round 4: waves per workgroup of k_deepfm_v2_joint1 again (4 / 8 / 16; round 3 measured 7.93 / 7.65 / 8.41 us with the barrier BEHIND the
# gathers) now that the barrier sits in front of them: libraries built with -DV2J1_WAVES=4 / 16 beside the product (8).
# RESULT (profiles/r04/experiments/r04_36/waves.txt): 4: 7.57 us, 8: 7.25 us, 16: 7.19 us (HBM-resident 9.08 / 8.77 / 8.78) -- sixteen no longer
# loses, and does not win enough to change the default.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_36}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in 8 16 4 8 16 4; do
  if [ $lib = 8 ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r04/libsparrow_hip_w$lib.so sparrowrecsys_amd/libsparrow_hip.so; fi
  a=$(timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | tail -1 | python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.2f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))")
  b=$(timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>/dev/null | tail -1 | python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.2f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))")
  echo "waves=$lib: config 2 $a | HBM-resident $b" | tee -a $O/waves.txt
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
