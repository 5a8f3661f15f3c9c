#!/bin/bash
# run W: the plan interpreter with gather-like segments in groups of four
set -u
mkdir -p gpurun_out/r02w
O=gpurun_out/r02w
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^NCCL\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" | tail -3 | tee $O/pytest.log
SPRK_FORCE_INTERPRETER=1 timeout 900 python -m pytest tests -m gpu -q -x -k "golden or config or sweep or ragged or missing" 2>&1 | grep -v "^NCCL\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" | tail -3 | tee $O/pytest_interp.log
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-100; tail -2 $O/$out.err; }
b c2_interp_group env SPRK_FORCE_INTERPRETER=1 python bench.py --cpu-seconds 0 --hbm-resident 0 --steps 200 --warmup 20
b c2_interp_nogroup env SPRK_FORCE_INTERPRETER=1 SPRK_TILE_GROUP=0 python bench.py --cpu-seconds 0 --hbm-resident 0 --steps 200 --warmup 20
b v2ref_interp_group env SPRK_FORCE_INTERPRETER=1 python bench.py --workload deepfm_v2_ref --steps 100 --warmup 10 --cpu-seconds 0
b v2ref_interp_nogroup env SPRK_FORCE_INTERPRETER=1 SPRK_TILE_GROUP=0 python bench.py --workload deepfm_v2_ref --steps 100 --warmup 10 --cpu-seconds 0
b c4pairs_interp_group env SPRK_V1_CHAIN=0 python bench.py --workload deepfm_c4 --steps 100 --warmup 10 --cpu-seconds 0
b c4pairs_interp_nogroup env SPRK_V1_CHAIN=0 SPRK_TILE_GROUP=0 python bench.py --workload deepfm_c4 --steps 100 --warmup 10 --cpu-seconds 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02w/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step']*1e3, j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'), j['config'].get('oracle_check_max_abs_err'), j['roofline'].get('kernel','')[:20])
    except Exception as e: print(f, 'ERR', e)
PY
