#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
for e in 1 0; do
echo "v1_chain=$e: $(SPRK_V1_CHAIN=$e python bench.py --workload deepfm_c2 --cpu-seconds 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|max_abs_err": [0-9.e-]*' | tr '\n' ' ')"
done
echo "v1 zipf: $(python bench.py --workload deepfm_c2 --cpu-seconds 0 --dist zipf 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ')"
echo "v1 b1m: $(python bench.py --workload deepfm_c2 --cpu-seconds 0 --batch 1048576 --steps 300 --warmup 30 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ')"
