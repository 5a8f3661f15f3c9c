#!/bin/bash
set -u
mkdir -p gpurun_out
echo "=== pytest all gpu"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
echo "=== bench"
for e in 1 0; do
echo "fold=$e din: $(SPRK_TILE_FOLD=$e python bench.py --steps 300 --warmup 30 --workload din_c3 --cpu-seconds 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"step_us_all_kernels": [0-9.]*\|max_abs_err": [0-9.e-]*' | tr '\n' ' ')"
echo "fold=$e deepfm pairs: $(SPRK_TILE_FOLD=$e python bench.py --workload deepfm_c2 --cpu-seconds 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|max_abs_err": [0-9.e-]*' | tr '\n' ' ')"
done
