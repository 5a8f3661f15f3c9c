#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "din or dien or stated or describe" 2>&1 | tail -12 | tee gpurun_out/r2_pytest_h.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-120; tail -2 gpurun_out/$out.err; }
b r2h_bench_c3 python bench.py --workload din_c3 --steps 200 --warmup 20 --cpu-seconds 0
SPRK_DIN_WPB=4 b r2h_bench_c3_wpb4 python bench.py --workload din_c3 --steps 200 --warmup 20 --cpu-seconds 0
