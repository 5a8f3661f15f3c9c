#!/bin/bash
# run S: k_deepfm_pairs on the derived {E | w1} row table
set -u
mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
timeout 900 python -m pytest tests -m gpu -q -x -k "pairs or deepfm_pair or stated or deepfm" 2>&1 | tail -4 | tee $O/pytest_pairs.log
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-100; tail -2 $O/$out.err; }
b pairs_tab python bench.py --workload deepfm_c2 --cpu-seconds 0
b pairs_notab env SPRK_V1_ROWTAB=0 python bench.py --workload deepfm_c2 --cpu-seconds 0
b pairs_tab_strict python bench.py --workload deepfm_c2 --cpu-seconds 0 --launch-batches 1 --overlap-streams 0
b pairs_notab_strict env SPRK_V1_ROWTAB=0 python bench.py --workload deepfm_c2 --cpu-seconds 0 --launch-batches 1 --overlap-streams 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02s/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step']*1e3, j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'), j['config'].get('oracle_check_max_abs_err'))
    except Exception as e: print(f, 'ERR', e)
PY
