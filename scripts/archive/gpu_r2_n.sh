#!/bin/bash
# run N: k_din_attn with the swizzled LDS tile; 3 vs 4 waves per SIMD
set -u
mkdir -p gpurun_out/r02n
O=gpurun_out/r02n
timeout 600 python -m pytest tests -m gpu -q -x -k "din" 2>&1 | tail -4 | tee $O/pytest_din.log
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-120; tail -2 $O/$out.err; }
b c3_w12 python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20
b c3_w16 env SPRK_DIN_WPB=16 python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20
b c3_w4 env SPRK_DIN_WPB=4 python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20
b c3_w12_strict python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20 --launch-batches 1 --overlap-streams 0
b c3_w16_strict env SPRK_DIN_WPB=16 python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20 --launch-batches 1 --overlap-streams 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02n/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step']*1e3, j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'), j['config'].get('oracle_check_max_abs_err'))
    except Exception as e: print(f, 'ERR', e)
PY
