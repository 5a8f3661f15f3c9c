#!/bin/bash
# run M: is k_din_attn sensitive to where its rows come from?  (uniform ids over the 131 k-row table vs a 1 k-row window)
set -u
mkdir -p gpurun_out/r02m
O=gpurun_out/r02m
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-200; tail -2 $O/$out.err; }
b c3_uniform python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20
b c3_hot python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20 --dist hot
b c3_uniform_strict python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20 --launch-batches 1 --overlap-streams 0
b c3_hot_strict python bench.py --workload din_c3 --cpu-seconds 0 --steps 200 --warmup 20 --dist hot --launch-batches 1 --overlap-streams 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02m/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step']*1e3, j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
