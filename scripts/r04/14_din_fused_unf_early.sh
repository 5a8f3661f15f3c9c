#!/bin/bash
# Round 4: k_din_fused's tail with the raw rows requested BEFORE the slot loop and the folded rows added last; then several batches
# per launch on the fused kernel (SPRK_DIN_FUSED_MB=1) against the attention + tail pipeline.  BASELINE config 3.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_14
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "din or DIN" > $O/pytest_din.log 2>&1
tail -3 $O/pytest_din.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%s: step %.2f us   dominant kernel %.2f us   value %.4g' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us'], l['value']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
for unf in 1 0 1; do
  SPRK_DIN_FUSED_UNF=$unf timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/unf$unf.err | tail -1 > $O/unf$unf.json
  show $O/unf$unf.json "strict UNF=$unf"
done
MBF="--cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe"
for mb in 0 1 0 1; do
  SPRK_DIN_FUSED_MB=$mb timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/mb$mb.err | tail -1 > $O/mb$mb.json
  show $O/mb$mb.json "16 batches per launch, FUSED_MB=$mb"
done
