#!/bin/bash
# Round 4: k_din_tail with the first task's loads requested in dependency order and nothing waiting for the image (ids, operands,
# image pieces, rows; wait for all but the rows; barrier) against the previous form (scripts/r04/libsparrow_hip_head.so: image and ids,
# barrier, then operands and rows).  DIN.py / DIEN.py literal shapes (the two-launch path), strict order, one batch per launch.
# RESULT (profiles/r04/experiments/r04_22): SLOWER -- the tail launch 18.6 us against 16.3 us per 65 536 samples; not kept (the
# reordered k_din_tail.h is not in the tree; this script documents the experiment and needs both libraries built by hand).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_22}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%s: step %.2f us   first stage %.2f us   value %.4g' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us'], l['value']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "din or DIN or dien or DIEN" > $O/pytest.log 2>&1
tail -1 $O/pytest.log
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in new head new head; do
  if [ $lib = head ]; then cp scripts/r04/libsparrow_hip_head.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; fi
  for w in din_ref dien_ref; do
    timeout 200 python bench.py --workload $w --steps 120 --warmup 12 $STRICT 2>$O/${w}_$lib.err | tail -1 > $O/${w}_$lib.json
    show $O/${w}_$lib.json "$w $lib"
  done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
SPRK_DIN_FUSED=0 timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/c3_two.err | tail -1 > $O/c3_two.json
show $O/c3_two.json "din_c3 two launches (new)"
