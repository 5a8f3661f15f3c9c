#!/bin/bash
# Round 4: k_din_fused, one batch per launch: the attention's A fragments through LDS (one 8-KB copy per workgroup instead of one per
# wave), the tail's raw rows behind the image's pieces, the numerics as hidden loads -- against the previous commit's library
# (scripts/r04/libsparrow_hip_head.so).  DIN tests first.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_28}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "din or DIN" > $O/pytest_din.log 2>&1
tail -1 $O/pytest_din.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%-34s step %.2f us   attention-only %.2f us   value %.4g' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us'], l['value']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in new head new head; do
  if [ $lib = head ]; then cp scripts/r04/libsparrow_hip_head.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; fi
  timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/strict_$lib.err | tail -1 > $O/strict_$lib.json
  show $O/strict_$lib.json "strict $lib"
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
MBF="--cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe"
timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/mb.err | tail -1 > $O/mb.json
show $O/mb.json "16 batches per launch (new)"
