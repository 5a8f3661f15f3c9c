#!/bin/bash
# DIN round: parity tests of the attention kernel, then bench A/B (k_din_attn vs legacy k_din_pool) + kernel trace.
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "=== pytest din"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "din" 2>&1 | tail -15 | tee gpurun_out/pytest_din.log
echo "=== bench din"
timeout 300 python bench.py --steps 300 --warmup 30 --workload din_c3 --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/bench_c3_attn.json
SPRK_DIN_LEGACY=1 timeout 300 python bench.py --steps 100 --warmup 10 --workload din_c3 --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/bench_c3_legacy.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --steps 50 --warmup 5 --workload din_c3 --cpu-seconds 0 --no-check > $R/gpurun_out/prof_c3.log 2>&1
cd $R
for f in $(find gpurun_out/prof_c3 -name "*kernel_stats.csv"); do echo "--- $f"; head -5 $f | cut -c1-220; done
