#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_tail
for rep in 1 2; do
for v in f16 f32; do
  if [ $v = f32 ]; then export SPRK_TAIL_POOLED_F16=0; else unset SPRK_TAIL_POOLED_F16; fi
  for wl in din_ref din_c3; do
    python bench.py --workload $wl --steps 200 --warmup 20 --cpu-seconds 0 --no-hardware-probe --no-check > gpurun_out/r03_tail/${wl}_$v.json 2>/dev/null
    python - <<PY
import json
l=json.loads([x for x in open('gpurun_out/r03_tail/${wl}_$v.json').read().splitlines() if x.startswith('{"metric"')][-1])
r=l['roofline']
print('$rep $v $wl', 'us/step %.2f' % (l['ms_per_step']*1e3), 'stage %.2f strict step %.2f tail(strict) %.2f' % (r['avg_launch_us'], r.get('step_us_all_kernels', 0), r.get('step_us_all_kernels', 0)-r['avg_launch_us']))
PY
  done
done
done
