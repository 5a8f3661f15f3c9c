#!/bin/bash
# Round 4: the several-batches-per-launch DIN pipeline (attention launch of a group + one tail launch, groups alternating over two
# streams) with its attention on k_din_attn_cols (8-wave workgroups, 122 VGPRs: two per CU, and room beside them for the tail kernel)
# against k_din_fused<TAIL = false> (164 VGPRs: one workgroup per CU).  Same box, twice each.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_07
mkdir -p $O
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    r = l['roofline']
    print('%-28s' % sys.argv[1].split('/')[-1], 'us/step %.2f' % (l['ms_per_step'] * 1e3), ' attention-only us %.2f' % r['avg_launch_us'],
          ' strict step us %.2f' % r.get('step_us_all_kernels', 0), ' err', l['config'].get('oracle_check_max_abs_err'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
MBF="--cpu-seconds 0 --side-workloads= --no-hardware-probe"
for rep in 1 2; do
b c3_mb_cols_$rep python bench.py --workload din_c3 --steps 320 --warmup 32 $MBF
b c3_mb_fusedattn_$rep SPRK_DIN_MB_ATTN_FUSED=1 python bench.py --workload din_c3 --steps 320 --warmup 32 $MBF
b c3_mb_r3path_$rep SPRK_DIN_FUSED=0 python bench.py --workload din_c3 --steps 320 --warmup 32 $MBF
done
