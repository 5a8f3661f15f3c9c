#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pairs or deepfm or golden or first_dense" 2>&1 | tail -25 | tee gpurun_out/r2_pytest_f1.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-260; tail -2 gpurun_out/$out.err; }
b r2f_bench_pairs python bench.py --workload deepfm_c2 --steps 200 --warmup 20 --cpu-seconds 0
SPRK_DYN_F16=0 b r2f_bench_pairs_f32 python bench.py --workload deepfm_c2 --steps 200 --warmup 20 --cpu-seconds 0
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r2_pytest_f.log
