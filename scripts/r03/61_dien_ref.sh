#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_dien
python bench.py --workload dien_ref --steps 200 --warmup 20 --cpu-seconds 0 --no-hardware-probe > gpurun_out/r03_dien/dien_ref.json 2> gpurun_out/r03_dien/dien_ref.err
tail -3 gpurun_out/r03_dien/dien_ref.err
python - <<PY
import json
l=json.loads([x for x in open('gpurun_out/r03_dien/dien_ref.json').read().splitlines() if x.startswith('{"metric"')][-1])
print('value %.4g us/step %.2f' % (l['value'], l['ms_per_step']*1e3), json.dumps(l['roofline'])[:500], l['config'].get('oracle_check_max_abs_err'))
PY
