#!/bin/bash
# Round 3, GPU call 4: pairs kernel with the deep part's own tables + one-task shape; joint/joint1 bit identity; whole suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_04
mkdir -p $O
echo "=== tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -25 | tee $O/pytest_gpu.log
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print(sys.argv[1].split('/')[-1], l['roofline']['kernel'][:40], 'value %.3g' % l['value'], 'strict us %.3f frac %.4f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']),
          'two-streams %.3g' % l.get('value_one_batch_per_launch_two_streams', 0), 'err', l['config']['oracle_check_max_abs_err'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
b pairs_one python bench.py --workload deepfm_c2 --cpu-seconds 0
b pairs_loop SPRK_V1_ONE=0 python bench.py --workload deepfm_c2 --cpu-seconds 0
b pairs_c4 python bench.py --workload deepfm_c4 --cpu-seconds 0
tail -3 $O/pairs_c4.err
