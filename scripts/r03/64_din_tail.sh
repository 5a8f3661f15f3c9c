#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_tail
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_blocks.py tests/test_gpu_shape_sweep.py tests/test_gpu_stated_sizes.py -x -q -m gpu -k "din or dien or DIN or tail" 2>&1 | grep -v "NCCL\|RCCL" | tail -5
for wl in din_ref dien_ref din_c3; do
  python bench.py --workload $wl --steps 200 --warmup 20 --cpu-seconds 0 --no-hardware-probe > gpurun_out/r03_tail/$wl.json 2>/dev/null
  python - <<PY
import json
l=json.loads([x for x in open('gpurun_out/r03_tail/$wl.json').read().splitlines() if x.startswith('{"metric"')][-1])
r=l['roofline']
print('$wl', 'value %.4g us/step %.2f' % (l['value'], l['ms_per_step']*1e3), 'stage %.2f us, strict step %.2f us' % (r['avg_launch_us'], r.get('step_us_all_kernels', 0)), l['config'].get('oracle_check_max_abs_err'))
PY
done
