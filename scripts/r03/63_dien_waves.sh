#!/bin/bash
# DM_WAVES sweep for k_dien_seq_mfma (rebuilds the library on the GPU box: hipcc is there)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_dien
for wv in 4 8 16; do
  sed -i "s/^#define DM_WAVES .*/#define DM_WAVES $wv/" sparrowrecsys_amd/csrc/k_dien_mfma.h
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  python bench.py --workload dien_ref --steps 200 --warmup 20 --cpu-seconds 0 --no-hardware-probe > gpurun_out/r03_dien/dien_ref_w$wv.json 2>/dev/null
  python - <<PY
import json
l=json.loads([x for x in open('gpurun_out/r03_dien/dien_ref_w$wv.json').read().splitlines() if x.startswith('{"metric"')][-1])
print('waves $wv', 'value %.4g us/step %.2f' % (l['value'], l['ms_per_step']*1e3), 'stage us %.2f' % (l['roofline']['avg_launch_us']), l['config'].get('oracle_check_max_abs_err'))
PY
done
