#!/bin/bash
# Round 4: k_mlp_rows with the wide part as a template parameter (no run-time branch around its loads: hipcc's waitcnt pass had put two
# vmcnt(0) behind the next task's row gathers) against the previous commit's library.  MLP tests, then configs 5 and EmbeddingMLP.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_30}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py -m gpu -x -q -k "mlp or wide or Wide or embedding" > $O/pytest_mlp.log 2>&1
tail -1 $O/pytest_mlp.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%-30s kernel %.2f us   frac %.3f   value %.4g' % (sys.argv[2], l['roofline']['avg_launch_us'], l['roofline']['frac'], l['value']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
for lib in new head new head; do
  if [ $lib = head ]; then cp scripts/r04/libsparrow_hip_head.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so; fi
  for w in widedeep_c5 embedding_mlp_ref; do
    timeout 200 python bench.py --workload $w --steps 100 --warmup 10 $STRICT 2>$O/${w}_$lib.err | tail -1 > $O/${w}_$lib.json
    show $O/${w}_$lib.json "$w $lib"
  done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
