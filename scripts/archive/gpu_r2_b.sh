#!/bin/bash
# round-2 second look: dispatch skew / device-side boundary ubench, the re-worked bench line, new workloads, new host-API tests
set -u
mkdir -p gpurun_out
echo "=== launch floor / skew"
timeout 200 scripts/ubench/launch_floor 2>&1 | tail -22 | tee gpurun_out/r2_launch_floor2.log
echo "=== pytest host api"
timeout 600 python -m pytest tests/test_gpu_host_api.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r2_pytest_hostapi.log
echo "=== bench driver flags"
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-600; tail -3 gpurun_out/$out.err; }
b r2_bench_driver2 python bench.py --gpus 1 --steps 20 --warmup 5
b r2_bench_default python bench.py --cpu-seconds 0 --hbm-resident 0
for w in deepfm_c2 din_c3 widedeep_c5 deepfm_v2_ref neuralcf_ref deepfm_v2_c4 deepfm_c4; do
  echo "--- $w"
  b r2_bench_$w python bench.py --workload $w --steps 200 --warmup 20 --cpu-seconds 0
done
echo "=== gloo 2 ranks on one GPU, self-spawned"
b r2_bench_gloo2 python bench.py --gpus 2 --backend gloo --steps 40 --warmup 5 --cpu-seconds 0
echo "=== pytest all"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r2_pytest_b.log
