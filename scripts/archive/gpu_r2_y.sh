#!/bin/bash
# run Y: k_deepfm_pairs with a static split scale for deep0's embedding block
set -u
mkdir -p gpurun_out/r02y
O=gpurun_out/r02y
timeout 900 python -m pytest tests -m gpu -q -x -k "pairs or deepfm_pair or stated or deepfm or sweep" 2>&1 | tail -3 | tee $O/pytest_pairs.log
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-100; tail -2 $O/$out.err; }
b pairs_static python bench.py --workload deepfm_c2 --cpu-seconds 0
b pairs_dynamic env SPRK_V1_STATIC_SCALE=0 python bench.py --workload deepfm_c2 --cpu-seconds 0
b c4pairs_static python bench.py --workload deepfm_c4 --steps 200 --warmup 20 --cpu-seconds 0
b c4pairs_dynamic env SPRK_V1_STATIC_SCALE=0 python bench.py --workload deepfm_c4 --steps 200 --warmup 20 --cpu-seconds 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02y/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step']*1e3, j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'), j['config'].get('oracle_check_max_abs_err'))
    except Exception as e: print(f, 'ERR', e)
PY
