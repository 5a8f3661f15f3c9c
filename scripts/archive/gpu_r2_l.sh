#!/bin/bash
# round 2, run L: the whole GPU suite (log for profiles/r02), peer-write all-gather on one box, v2j numerics fold A/B
set -u
mkdir -p gpurun_out/r02l
O=gpurun_out/r02l
timeout 1200 python -m pytest tests -m gpu -q -rs --durations=8 2>&1 | grep -v "^NCCL\|^$" | tail -40 | tee $O/pytest_gpu.log
b() { out=$1; shift; timeout 600 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-220; tail -3 $O/$out.err; }
b bench_c2_driver python bench.py --gpus 1 --steps 20 --warmup 5
b bench_c2_strict python bench.py --cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0
b bench_c2_lb1 python bench.py --cpu-seconds 0 --launch-batches 1 --hbm-resident 0
b bench_c2_forced_collective_peer env SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0 --hbm-resident 0 --collective peer
b bench_c2_gloo2_peer python bench.py --gpus 2 --backend gloo --collective peer --steps 20 --warmup 5 --cpu-seconds 0 --hbm-resident 0 --min-region-ms 1 --regions 1 --settle-ms 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02l/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step'], j.get('value_one_batch_per_launch'), j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'), j['config'].get('collective'))
    except Exception as e: print(f, 'ERR', e)
PY
