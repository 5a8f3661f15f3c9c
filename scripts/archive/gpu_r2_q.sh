#!/bin/bash
# run Q: k_din_attn with several batches per launch
set -u
mkdir -p gpurun_out/r02q
O=gpurun_out/r02q
timeout 900 python -m pytest tests -m gpu -q -x -k "din or dien" 2>&1 | tail -4 | tee $O/pytest_din.log
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-120; tail -2 $O/$out.err; }
b c3_mb python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20
b c3_nomb env SPRK_DIN_ATTN_MB=0 python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20
b c3_mb_1stream python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20 --overlap-streams 0
b c3_nomb_1stream env SPRK_DIN_ATTN_MB=0 python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20 --overlap-streams 0
b c3_mb_b python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02q/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step']*1e3, j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'), j['config'].get('oracle_check_max_abs_err'))
    except Exception as e: print(f, 'ERR', e)
PY
