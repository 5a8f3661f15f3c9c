#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_dien
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_blocks.py -x -q -m gpu -k "dien" 2>&1 | grep -v "NCCL\|RCCL" | tail -8
for v in mfma lanes; do
  if [ $v = lanes ]; then export SPRK_DIEN_MFMA=0; else unset SPRK_DIEN_MFMA; fi
  python bench.py --workload dien_ref --steps 200 --warmup 20 --cpu-seconds 0 --no-hardware-probe > gpurun_out/r03_dien/dien_ref_$v.json 2>/dev/null
  python - <<PY
import json
l=json.loads([x for x in open('gpurun_out/r03_dien/dien_ref_$v.json').read().splitlines() if x.startswith('{"metric"')][-1])
print('$v', 'value %.4g us/step %.2f' % (l['value'], l['ms_per_step']*1e3), 'stage us %.2f frac %.4f' % (l['roofline']['avg_launch_us'], l['roofline']['frac']), l['config'].get('oracle_check_max_abs_err'))
PY
done
