#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_unf
timeout 1200 python -m pytest tests/test_gpu_rows_chain.py tests/test_reference_blocks.py tests/test_gpu_shape_sweep.py -x -q -m gpu 2>&1 | grep -v "NCCL\|RCCL" | tail -12
for v in unf folded; do
  if [ $v = folded ]; then export SPRK_ROWS_UNF=0; else unset SPRK_ROWS_UNF; fi
  python bench.py --workload deepfm_v2_ref --steps 400 --warmup 40 --cpu-seconds 0 --no-hardware-probe > gpurun_out/r03_unf/v2_ref_$v.json 2>/dev/null
  python bench.py --workload deepfm_v2_ref --steps 400 --warmup 40 --cpu-seconds 0 --no-hardware-probe --launch-batches 1 --overlap-streams 0 > gpurun_out/r03_unf/v2_ref_${v}_strict.json 2>/dev/null
  python - <<PY
import json
a=json.loads([x for x in open('gpurun_out/r03_unf/v2_ref_$v.json').read().splitlines() if x.startswith('{"metric"')][-1])
b=json.loads([x for x in open('gpurun_out/r03_unf/v2_ref_${v}_strict.json').read().splitlines() if x.startswith('{"metric"')][-1])
print('$v', 'value %.4g (%.2f us/step)' % (a['value'], a['ms_per_step']*1e3), 'strict %.2f us frac %.4f' % (b['roofline']['avg_launch_us'], b['roofline']['frac']), a['config'].get('kernel'))
PY
done
