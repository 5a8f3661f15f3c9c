#!/bin/bash
# Round 4: k_deepfm_v2_joint1's HOIST form (weight fragments read in front of the gathers) picked at finalize for tables larger than the
# Infinity Cache, against the previous commit's library: the bit-identity test over both forms, then config 2 with cache-resident
# tables (must not change: the default form) and with HBM-resident ones (the HOIST form), config 4.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_37}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "joint1 or v2_joint" > $O/pytest.log 2>&1
tail -1 $O/pytest.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in new head new head; do
  if [ $lib = head ]; then cp scripts/r04/libsparrow_hip_head.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; fi
  a=$(timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | tail -1 | tee $O/c2_$lib.json | python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.2f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))")
  b=$(timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>/dev/null | tail -1 | tee $O/c2_hbm_$lib.json | python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.2f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))")
  echo "$lib: config 2 $a | HBM-resident $b"
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
