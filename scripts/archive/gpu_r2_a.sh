#!/bin/bash
# round-2 first look: sanity, launch floor, v2j trace, baseline bench at the driver's flags
set -u
mkdir -p gpurun_out
echo "=== launch floor"
timeout 120 scripts/ubench/launch_floor 2>&1 | tee gpurun_out/r2_launch_floor.log
echo "=== trace"
timeout 300 python scripts/exp_v2.py trace 2>&1 | tail -40 | tee gpurun_out/r2_trace.log
echo "=== bench driver flags"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/r2_bench_driver.json | cut -c1-400
timeout 300 python bench.py --cpu-seconds 0 --launch-batches 1 --overlap-streams 0 2>&1 | tail -1 | tee gpurun_out/r2_bench_strict.json | cut -c1-400
echo "=== pytest"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r2_pytest_a.log
