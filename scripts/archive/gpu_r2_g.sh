#!/bin/bash
set -u
mkdir -p gpurun_out
python scripts/dbg_pairs.py 2>&1 | tail -9
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r2_pytest_g.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-260; tail -2 gpurun_out/$out.err; }
b r2g_bench_pairs python bench.py --workload deepfm_c2 --steps 200 --warmup 20 --cpu-seconds 0
b r2g_bench_c3 python bench.py --workload din_c3 --steps 200 --warmup 20 --cpu-seconds 0
b r2g_bench_c5 python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
