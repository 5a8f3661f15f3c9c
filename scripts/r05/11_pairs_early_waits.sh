#!/bin/bash
# Round 5, call 11: k_deepfm_pairs1 with (a) deep0's bias + numerics MFMAs in front of the pair dots and (b) one path through its gather (the
# table form folded at compile time, row loads unconditional): the waitcnt pass no longer puts a vmcnt(0) in front of the whole scoring stage
# ("load x12 wait(8) mfma x2 wait(1) mfma x18 wait(0) mfma x24" instead of "wait(0) mfma x44").  Against round 4's library; parity first.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_11}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py tests/test_gpu_stated_sizes.py tests/test_gpu_sharded_table.py -m gpu -x -q -k "pair or deepfm or sharded or config4" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());print('%.3f us frac %.3f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']))"; }
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for rep in 1 2; do
for lib in r04 product; do
  if [ $lib = product ]; then cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$lib.so sparrowrecsys_amd/libsparrow_hip.so; fi
  a=$(timeout 200 python bench.py --workload deepfm_c2 --steps 400 --warmup 40 $STRICT 2>/dev/null | tail -1 | get)
  b=$(timeout 200 python bench.py --workload deepfm_ref --steps 400 --warmup 40 $STRICT 2>/dev/null | tail -1 | get)
  c=$(timeout 300 python bench.py --workload deepfm_c4 --steps 200 --warmup 20 $STRICT 2>/dev/null | tail -1 | get)
  echo "$lib: c2_pairs $a | deepfm_ref $b | c4_pairs $c" | tee -a $O/pairs.txt
done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
