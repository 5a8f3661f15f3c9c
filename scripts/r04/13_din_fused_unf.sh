#!/bin/bash
# Round 4: k_din_fused's tail with the two large-vocabulary columns (userId, candidate movieId) as raw split rows on the matrix pipe
# (128 bytes per id) instead of folded rows (512 bytes per id): DIN parity tests first, then A/B on one box, BASELINE config 3.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_13
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "din or DIN" > $O/pytest_din.log 2>&1
tail -3 $O/pytest_din.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for unf in 1 0 1 0; do
  SPRK_DIN_FUSED_UNF=$unf timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/unf$unf.err | tail -1 > $O/unf$unf.json
  python - $O/unf$unf.json $unf <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('UNF=%s fused step %.2f us   attention-only %.2f us   value %.3g' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us'], l['value']))
except Exception as e:
    print('UNF=%s FAILED %s' % (sys.argv[2], e))
PY
done
# with the oracle check on (the bench's own parity leg)
timeout 300 python bench.py --workload din_c3 --steps 60 --warmup 6 --cpu-seconds 2 --side-workloads= 2>$O/check.err | tail -1 > $O/check.json
python -c "
import json;l=json.loads(open('$O/check.json').read());print('checked line: value %.4g  check %s' % (l['value'], l.get('check')))"
