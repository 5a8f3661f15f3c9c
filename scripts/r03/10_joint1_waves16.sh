#!/bin/bash
# Round 3, GPU call 13: k_deepfm_v2_joint1 with 16-wave workgroups (one per CU: half the image staging traffic).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_10
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "joint1" 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -3
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
l = json.loads(open(sys.argv[1]).read())
h = l.get('roofline_hbm_resident')
print(sys.argv[1].split('/')[-1], 'value %.4g' % l['value'], 'strict us %.3f frac %.4f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']),
      'two-streams %.3g' % l.get('value_one_batch_per_launch_two_streams', 0), ('hbm: us %.3f frac %.4f' % (h['avg_launch_us'], h['frac'])) if h else '')
PY
}
b w16_1 python bench.py --cpu-seconds 0 --side-workloads=
b w16_2 python bench.py --cpu-seconds 0 --side-workloads=
