#!/bin/bash
# Round 5, call 12: the joint kernels' final cross-row sum as two permlane swaps (rows4_sum) instead of two ds_bpermute round trips
# (libsparrow_hip_swap.so, one-unit build of the tree) against the product; parity (bit-identity between the joint kernels) first.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_12}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.3f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"; }
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
cp scripts/r05/libsparrow_hip_prio.so sparrowrecsys_amd/libsparrow_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py -m gpu -x -q -k "v2 or joint" > $O/pytest_v2.log 2>&1; grep -E "passed|failed" $O/pytest_v2.log | tail -1
for rep in 1 2 3; do
for lib in product prio; do
  if [ $lib = product ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$lib.so sparrowrecsys_amd/libsparrow_hip.so; fi
  a=$(timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  b=$(timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>/dev/null | tail -1 | get)
  v=$(timeout 300 python bench.py --steps 64 --warmup 64 --cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe --variants 0 2>/dev/null | tail -1 | python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.4g samples/s' % l['value'])")
  echo "$lib: config 2 $a | HBM-resident $b | 64 batches per launch $v" | tee -a $O/prio.txt
done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
