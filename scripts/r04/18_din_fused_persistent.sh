#!/bin/bash
# Round 4: k_din_fused<MB> as a PERSISTENT launch (tables and image staged once, every wave walks its tasks, no barrier per task)
# against the attention + tail pipeline, several batches per launch; and the one-batch strict form of the same tree.  BASELINE config 3.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_18}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "din or DIN" > $O/pytest_din.log 2>&1
tail -3 $O/pytest_din.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%s: step %.2f us   dominant kernel %.2f us   value %.4g   (%.2f us per batch)' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us'], l['value'], 32768 / l['value'] * 1e6))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/strict.err | tail -1 > $O/strict.json
show $O/strict.json "strict"
MBF="--cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe"
for mb in 0 1 0 1; do
  SPRK_DIN_FUSED_MB=$mb timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/mb$mb.err | tail -1 > $O/mb$mb.json
  show $O/mb$mb.json "16 batches per launch, FUSED_MB=$mb"
done
SPRK_DIN_FUSED_MB=1 timeout 200 python bench.py --workload din_c3 --steps 256 --warmup 64 --launch-batches 64 $MBF 2>$O/mb1_64.err | tail -1 > $O/mb1_64.json
show $O/mb1_64.json "64 batches per call, FUSED_MB=1"
SPRK_DIN_FUSED_MB=1 SPRK_MANY_STREAMS=0 timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/mb1_s0.err | tail -1 > $O/mb1_s0.json
show $O/mb1_s0.json "16 batches per launch, FUSED_MB=1, one stream"
# with the oracle check on
SPRK_DIN_FUSED_MB=1 timeout 300 python bench.py --workload din_c3 --steps 64 --warmup 16 --cpu-seconds 2 --side-workloads= 2>$O/check.err | tail -1 > $O/check.json
python -c "
import json;l=json.loads(open('$O/check.json').read());print('checked line (FUSED_MB=1): value %.4g' % l['value'])"
tail -2 $O/check.err
