#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_lb
for lb in 16 32 64; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --launch-batches $lb --cpu-seconds 0 --side-workloads "" --hbm-resident 0 --no-hardware-probe --input-batches 64 > gpurun_out/r03_lb/lb$lb.json 2> gpurun_out/r03_lb/lb$lb.err
  python - <<PY
import json
l=json.loads([x for x in open('gpurun_out/r03_lb/lb$lb.json').read().splitlines() if x.startswith('{"metric"')][-1])
print($lb, 'value %.4g us/step %.3f' % (l['value'], l['ms_per_step']*1e3), l['roofline_timed_region']['frac'] if 'roofline_timed_region' in l else None)
PY
done
