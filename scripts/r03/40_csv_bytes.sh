#!/bin/bash
# byte-parallel tokenizer: parity tests, then 20 M rows with the byte-parallel pass and with round 2's line-at-a-time pass
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_csv
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | grep -v "NCCL\|RCCL" | tail -15
for v in bytes lines; do
  if [ $v = lines ]; then export SPRK_CSV_BYTES=0; else unset SPRK_CSV_BYTES; fi
  timeout 600 python scripts/bench_ingest.py --rows 20000000 --threads 128 --device > gpurun_out/r03_csv/ingest_20m_$v.json 2> gpurun_out/r03_csv/ingest_20m_$v.err
  python - <<PY
import json
l=json.loads(open('gpurun_out/r03_csv/ingest_20m_$v.json').read().strip().splitlines()[-1])
print('$v', l['device'])
PY
done
unset SPRK_CSV_BYTES
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/csvprof -o csv -- python $GRAFT_REPO_ROOT/scripts/bench_ingest.py --rows 20000000 --threads 128 --device > /dev/null 2>&1
f=$(ls /tmp/csvprof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r03_csv/ingest_kernel_stats.csv && head -8 $f | cut -c1-150
