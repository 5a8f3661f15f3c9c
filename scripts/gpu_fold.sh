#!/bin/bash
set -u
mkdir -p gpurun_out
echo "=== pytest all gpu"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
echo "=== bench"
for e in 1 0; do
echo "din_tail=$e din: $(SPRK_DIN_TAIL=$e python bench.py --steps 300 --warmup 30 --workload din_c3 --cpu-seconds 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"step_us_all_kernels": [0-9.]*\|max_abs_err": [0-9.e-]*' | tr '\n' ' ')"
done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --steps 50 --warmup 5 --workload din_c3 --cpu-seconds 0 --no-check > $R/gpurun_out/prof_c3.log 2>&1
head -4 $R/gpurun_out/prof_c3/c3_kernel_stats.csv | cut -c1-200
