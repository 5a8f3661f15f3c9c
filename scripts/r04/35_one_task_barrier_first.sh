#!/bin/bash
# Round 4: the other one-task-per-wave kernels (k_deepfm_pairs1, k_rows_chain1) with the workgroup's barrier in front of the row requests,
# like k_deepfm_v2_joint1 (scripts/r04/33_*), against the previous commit's library (scripts/r04/libsparrow_hip_head.so).
# RESULT (profiles/r04/experiments/r04_35): k_deepfm_pairs1 gains (config 2 10.90 -> 10.79 us, DeepFM.py literal 9.17 -> 8.57 us) -- kept;
# k_rows_chain1 loses (DeepFM_v2.py literal 7.40 -> 7.55 us, NeuralCF 4.25 -> 4.39 us: three gathers per wave issue fast enough that the
# early barrier only delays them) -- NOT kept.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_35}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_chain.py -m gpu -x -q -k "pairs or deepfm or rows or neuralcf or v2" > $O/pytest.log 2>&1
tail -1 $O/pytest.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%-26s kernel %.2f us   frac %.3f' % (sys.argv[2], l['roofline']['avg_launch_us'], l['roofline']['frac']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in new head new head; do
  if [ $lib = head ]; then cp scripts/r04/libsparrow_hip_head.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; fi
  for w in deepfm_c2 deepfm_ref deepfm_v2_ref neuralcf_ref; do
    timeout 200 python bench.py --workload $w --steps 300 --warmup 30 $STRICT 2>$O/${w}_$lib.err | tail -1 > $O/${w}_$lib.json
    show $O/${w}_$lib.json "$w $lib"
  done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
