#!/bin/bash
# Round 4: split-f16 weight fragments in LANE order inside their 1-KB pieces (conflict-free ds_read_b128) against the sample-major
# order of rounds 2-4 (scripts/r04/libsparrow_hip_head.so = the previous commit's library): every kernel that reads them, strict order,
# one batch per launch.  Then the whole GPU suite on the new tree.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_23}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%-28s step %.2f us   dominant kernel %.2f us   frac %.3f' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us'], l['roofline']['frac']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in new head new head; do
  if [ $lib = head ]; then cp scripts/r04/libsparrow_hip_head.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; fi
  for w in widedeep_c5 embedding_mlp_ref deepfm_c2 din_ref; do
    timeout 200 python bench.py --workload $w --steps 100 --warmup 10 $STRICT 2>$O/${w}_$lib.err | tail -1 > $O/${w}_$lib.json
    show $O/${w}_$lib.json "$w $lib"
  done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
