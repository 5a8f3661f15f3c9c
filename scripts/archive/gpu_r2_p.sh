#!/bin/bash
# run P: device CSV tokenizer, per-kernel times at 20 M rows
set -u
R=$(pwd)
mkdir -p gpurun_out/r02p
O=$R/gpurun_out/r02p
timeout 900 python -m pytest tests/test_gpu_ingest.py -q -x 2>&1 | tail -5
timeout 900 python scripts/bench_ingest.py --rows 20000000 --threads 128 --device > $O/bench_ingest_20m.json 2> $O/bench_ingest_20m.err; cat $O/bench_ingest_20m.json; tail -3 $O/bench_ingest_20m.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ingest -- python $R/scripts/bench_ingest.py --rows 20000000 --threads 128 --device > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" | cut -c1-200 | tee $O/ingest_kernel_stats.csv
