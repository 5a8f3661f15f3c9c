#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r2_pytest_d.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-300; tail -2 gpurun_out/$out.err; }
b r2d_bench_neuralcf_ref python bench.py --workload neuralcf_ref --steps 200 --warmup 20 --cpu-seconds 0
