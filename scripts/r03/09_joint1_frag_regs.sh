#!/bin/bash
# Round 3, GPU call 12: k_deepfm_v2_joint1 with its A fragments in registers (loaded at entry) vs read from LDS at scoring time.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_09
mkdir -p $O
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "joint1 or forward_many or deepfm_v2" 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -6
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
l = json.loads(open(sys.argv[1]).read())
h = l.get('roofline_hbm_resident')
print(sys.argv[1].split('/')[-1], 'value %.4g' % l['value'], 'strict us %.3f frac %.4f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']),
      'two-streams %.3g' % l.get('value_one_batch_per_launch_two_streams', 0), ('hbm: us %.3f frac %.4f' % (h['avg_launch_us'], h['frac'])) if h else '')
PY
}
for i in 1 2; do
b fr_regs_$i python bench.py --cpu-seconds 0 --side-workloads=
b fr_lds_$i SPRK_V2J1_FRAG_REGS=0 python bench.py --cpu-seconds 0 --side-workloads=
done
