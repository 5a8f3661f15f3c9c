#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_host_api.py -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r2_pytest_j.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-160; tail -3 gpurun_out/$out.err; }
SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 b r2j_force_sprk python bench.py --cpu-seconds 0 --hbm-resident 0 --collective sprk
SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 b r2j_force_torch python bench.py --cpu-seconds 0 --hbm-resident 0 --collective torch
