#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mlp" 2>&1 | tail -25 | tee gpurun_out/r2_pytest_e1.log
timeout 1200 python -m pytest tests/test_gpu_stated_sizes.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r2_pytest_e2.log
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r2_pytest_e.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-300; tail -2 gpurun_out/$out.err; }
b r2e_bench_c5 python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
SPRK_MLP_ROWS=0 b r2e_bench_c5_chain python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
b r2e_bench_v2ref python bench.py --workload deepfm_v2_ref --steps 200 --warmup 20 --cpu-seconds 0
