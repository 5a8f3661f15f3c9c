#!/bin/bash
# Round 4: k_deepfm_v2_joint1 A/B against the previous commit's library (scripts/r04/libsparrow_hip_head.so, built by hand from `git
# archive HEAD`).  Run 1 (profiles/r04/experiments/r04_33): the workgroup's barrier IN FRONT of the row requests (the waves arrive there
# within 0.5 us of each other; behind the requests they waited 1.4 us for the slowest issuer): config 2 7.56 -> 7.26 us, HBM-resident
# 9.15 -> 8.9 us -- kept.  Run 2 (r04_34): on top of it, the weight fragments' LDS reads in front of the gathers: HBM-resident 8.9 ->
# 8.68 us but config 2 7.26 -> 7.38 us (and 127 of 128 VGPRs) -- NOT kept.  Parity tests of the DeepFM_v2 kernels first.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_33}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v2 or joint or deepfm" > $O/pytest_v2.log 2>&1
tail -1 $O/pytest_v2.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%-26s kernel %.2f us   frac %.3f   value %.4g' % (sys.argv[2], l['roofline']['avg_launch_us'], l['roofline']['frac'], l['value']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in new head new head; do
  if [ $lib = head ]; then cp scripts/r04/libsparrow_hip_head.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; fi
  timeout 200 python bench.py --steps 400 --warmup 40 --input-batches 32 $STRICT 2>$O/c2_$lib.err | tail -1 > $O/c2_$lib.json
  show $O/c2_$lib.json "c2 $lib"
  timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>$O/c2_hbm_$lib.err | tail -1 > $O/c2_hbm_$lib.json
  show $O/c2_hbm_$lib.json "c2 HBM-resident $lib"
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
timeout 300 python bench.py --workload deepfm_v2_c4 --steps 200 --warmup 20 $STRICT 2>$O/c4.err | tail -1 > $O/c4_new.json
show $O/c4_new.json "c4 v2 new"
