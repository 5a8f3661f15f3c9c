#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "mlp or stated or golden or fold" 2>&1 | tail -8 | tee gpurun_out/r2_pytest_i.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-160; tail -2 gpurun_out/$out.err; }
b r2i_bench_c5 python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
