#!/bin/bash
set -u
mkdir -p gpurun_out
echo "=== pytest rows chain + savedmodel pins"
timeout 900 python -m pytest tests/test_gpu_rows_chain.py tests/test_savedmodel_pins.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r2_pytest_rows.log
echo "=== pytest all"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r2_pytest_c.log
b() { out=$1; shift; timeout 900 "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err; tail -1 gpurun_out/$out.json | cut -c1-300; tail -2 gpurun_out/$out.err; }
b r2c_bench_driver python bench.py --gpus 1 --steps 20 --warmup 5
for w in deepfm_v2_ref neuralcf_ref deepfm_v2_c4; do
  echo "--- $w"
  b r2c_bench_$w python bench.py --workload $w --steps 200 --warmup 20 --cpu-seconds 0
done
SPRK_V2_ROWS=1 b r2c_bench_c2_rows python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --hbm-resident 0
