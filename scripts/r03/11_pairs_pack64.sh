#!/bin/bash
# Round 3, GPU call 15: pair-dot DeepFM with {E_fm | E_deep} in one 128-byte line for the two deep fields (first-order weights of
# those fields from a compact array): 3 big lines per sample instead of 5.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_11
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "pairs or deepfm or pair_dot or golden or reference_lines" 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -6
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
l = json.loads(open(sys.argv[1]).read())
print(sys.argv[1].split('/')[-1], l['config']['kernel'], 'value %.4g' % l['value'], 'us/step %.3f' % (l['ms_per_step'] * 1e3),
      'strict us %.3f frac %.4f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']), 'two-streams %.3g' % l.get('value_one_batch_per_launch_two_streams', 0), 'err', l['config']['oracle_check_max_abs_err'])
PY
}
b pairs_c2 python bench.py --workload deepfm_c2 --cpu-seconds 0
b pairs_c2_again python bench.py --workload deepfm_c2 --cpu-seconds 0
b deepfm_ref python bench.py --workload deepfm_ref --cpu-seconds 0
