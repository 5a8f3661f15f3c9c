#!/bin/bash
# run O: device CSV tokenizer -- parity tests and throughput
set -u
mkdir -p gpurun_out/r02o
O=gpurun_out/r02o
timeout 900 python -m pytest tests/test_gpu_ingest.py -q -x 2>&1 | tail -25 | tee $O/pytest_ingest.log
timeout 900 python scripts/bench_ingest.py --rows 2000000 --threads 1,32,128 --device > $O/bench_ingest.json 2> $O/bench_ingest.err; cat $O/bench_ingest.json; tail -3 $O/bench_ingest.err
