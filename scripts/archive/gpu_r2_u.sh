#!/bin/bash
# run U: k_din_attn with 16 waves per workgroup (4 per SIMD), no row prefetch
set -u
mkdir -p gpurun_out/r02u
O=gpurun_out/r02u
SPRK_DIN_WPB=16 timeout 900 python -m pytest tests -m gpu -q -x -k "din or dien" 2>&1 | tail -4 | tee $O/pytest_din16.log
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-100; tail -2 $O/$out.err; }
b c3_w16 env SPRK_DIN_WPB=16 python bench.py --workload din_c3 --cpu-seconds 0 --steps 320 --warmup 32
b c3_w12 python bench.py --workload din_c3 --cpu-seconds 0 --steps 320 --warmup 32
b c3_w16_hot env SPRK_DIN_WPB=16 python bench.py --workload din_c3 --cpu-seconds 0 --steps 320 --warmup 32 --dist hot
b c3_w16_1stream env SPRK_DIN_WPB=16 python bench.py --workload din_c3 --cpu-seconds 0 --steps 320 --warmup 32 --overlap-streams 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02u/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step']*1e3, j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'), j['config'].get('oracle_check_max_abs_err'))
    except Exception as e: print(f, 'ERR', e)
PY
