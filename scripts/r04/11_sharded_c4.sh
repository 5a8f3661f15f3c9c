#!/bin/bash
# Round 4: BASELINE config 4 with its 27 M-row item tables as sprk_vtable's (row-sharded; here a world of one and two gloo-launched
# ranks sharing the device) against the replicated tables: same kernel, same time expected at N = 1; functional at N = 2.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_11
mkdir -p $O
b() { out=$1; shift; timeout 400 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read()); r = l['roofline']
    print('%-26s' % sys.argv[1].split('/')[-1], 'n_gpus', l['n_gpus'], 'value %.4g' % l['value'], 'strict us %.2f frac %.3f' % (r['avg_launch_us'], r['frac']), 'err', l['config']['oracle_check_max_abs_err'], '|', l['config'].get('tables_row_sharded'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[1].replace('.json', '.err')).read()[-1500:])
PY
}
COMMON="--steps 100 --warmup 10 --cpu-seconds 0 --side-workloads= --hbm-resident 0 --no-hardware-probe"
b c4_pairs_replicated python bench.py --workload deepfm_c4 $COMMON
b c4_pairs_sharded_n1 python bench.py --workload deepfm_c4 --shard-tables $COMMON
b c4_v2_sharded_n1 python bench.py --workload deepfm_v2_c4 --shard-tables $COMMON
b c4_pairs_sharded_gloo2 python bench.py --gpus 2 --backend gloo --workload deepfm_c4 --shard-tables $COMMON
